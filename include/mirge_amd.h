/*
 * mirge_amd.h -- C-ABI of the MI355X-native short-read annotation engine.
 *
 * This is the drop-in boundary for ONE path of miRge2.0's annotate mode: the
 * sequential bowtie cascade over collapsed unique reads plus the per-read
 * count tally.  The reference has no function ABI at this boundary -- it
 * shells out to the external `bowtie` / `bowtie-inspect` binaries and parses
 * their text output.  Each entry point below names the reference call site
 * (file:line under src/mirge/) whose work it replaces:
 *
 *   RAP = utils/runAnnotationPipeline.py   SUM = utils/summarize.py
 *   MAIN = __main__.py
 *
 * Conventions
 *   - plain C, no torch / C++ types in any signature;
 *   - every function returns 0 on success, <0 on error; the message is kept
 *     per-thread and read with mrg_last_error() (the Python host turns a
 *     non-zero status into the reference's "Alignment to library %s exited
 *     with none-zero status." + exit(1), RAP:661-663);
 *   - the caller allocates and owns every bulk buffer; the library owns only
 *     the opaque handles (mrg_index, mrg_ctx) and frees them in *_free/_destroy;
 *   - pointers named d_* are DEVICE (HBM) pointers, everything else is host
 *     memory; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - there is no CPU fallback: every compute entry point fails with
 *     MRG_ERR_NO_DEVICE when no gfx950 device is usable.
 *
 * Read encoding: 2 bits per base, A=0 C=1 G=2 T=3, base i of a read in bits
 * [2i,2i+1] of word (i/32), words stored structure-of-arrays:
 * d_reads[w * n + r] is word w of read r (coalesced when a wave takes 64
 * consecutive reads).  Reads containing N carry an optional mask of the same
 * shape (bit 2i of the d_nmask word set = base i is N, its 2-bit code is then
 * 0; N always counts as a mismatch, as in bowtie 1).
 */
#ifndef MIRGE_AMD_H
#define MIRGE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRG_OK 0
#define MRG_ERR_ARG (-1)
#define MRG_ERR_IO (-2)
#define MRG_ERR_NO_DEVICE (-3)
#define MRG_ERR_HIP (-4)
#define MRG_ERR_NOMEM (-5)
#define MRG_ERR_FORMAT (-6)

#define MRG_MAX_PASSES 16
#define MRG_MAX_WORDS 8 /* reads up to 255 nt (the length is one byte) */

typedef struct mrg_index mrg_index; /* host-side FM index of one library */
typedef struct mrg_ctx mrg_ctx;     /* one per GPU: HBM copies + workspaces */

int mrg_version(void);
const char *mrg_last_error(void);

/* ------------------------------------------------------------------ *
 * Index: replaces `bowtie-build` (offline) and `bowtie-inspect`
 * (SUM:6, RAP:610-611, RAP:630).  Host-only C++; no GPU needed.
 * ------------------------------------------------------------------ */

/* Build from in-memory entries (names[i], seqs[i] are NUL-terminated; seqs may
 * contain N/n, which splits an entry into un-alignable gaps as bowtie does). */
int mrg_index_build(const char *const *names, const char *const *seqs,
                    uint32_t n_ref, mrg_index **out);
/* Build from a FASTA file (multi-line records allowed; name = header up to
 * first whitespace, as bowtie-build records it). */
int mrg_index_build_fasta(const char *fasta_path, mrg_index **out);
/* Build from the reference's own library file: `<prefix>.1.ebwt` (bowtie 1), the only form in which
 * miRge.Libs ships its libraries (MAIN:262-281).  Entry names and sequences are recovered from
 * the BWT as `bowtie-inspect` does (SUM:6, RAP:610-611,630, W2C:649) and indexed like a FASTA
 * file.  EXPERIMENTAL: the format is restated from bowtie 1.1.x without a bowtie-built sample at
 * hand and validated by round trip with an independent test writer only (tests/helpers/); the
 * reader therefore re-derives every redundant field of the file (side occurrence counts, fchr,
 * ftab order, fragment table) from the BWT it decoded and fails with MRG_ERR_IO on any mismatch. */
int mrg_index_build_ebwt(const char *prefix, mrg_index **out);
int mrg_index_save(const mrg_index *ix, const char *path);
int mrg_index_load(const char *path, mrg_index **out);
void mrg_index_free(mrg_index *ix);

typedef struct mrg_index_info {
  uint32_t n_ref;      /* library entries (histogram bins for the miRNA lib) */
  uint32_t n_seg;      /* N-free segments */
  uint32_t n_bases;    /* concatenated text length (without sentinel) */
  uint32_t n_blocks;   /* 16-byte occ blocks (32 BWT symbols each) */
  uint32_t n_super;    /* superblocks (65536 BWT symbols each), 4 words each */
  uint32_t primary;    /* BWT row holding the sentinel */
  uint32_t text_words; /* 2-bit packed text, 32-bit words incl. padding */
  uint8_t ftab_ks[4];  /* k of each k-mer jump table, largest first, 0 = absent: {big 12..14 or 0,
                        * main 8..11, 6, 4} */
  uint32_t C[4];       /* first BWT row of each symbol */
  uint64_t bytes_fm;   /* n_blocks * 16 + n_super * 16 */
  uint64_t bytes_sa;   /* (n_bases + 1) * 8 */
} mrg_index_info;

int mrg_index_get_info(const mrg_index *ix, mrg_index_info *info);
/* `bowtie-inspect -n`: name of entry i, valid for the index lifetime (SUM:6-9). */
int mrg_index_name(const mrg_index *ix, uint32_t i, const char **name);
/* `bowtie-inspect`: sequence of entry i (N restored) into buf (cap bytes incl. NUL). */
int mrg_index_seq(const mrg_index *ix, uint32_t i, char *buf, uint32_t cap,
                  uint32_t *len);

/* Raw views for tests and the oracle's CPU port (read-only, index lifetime). */
typedef struct mrg_index_view {
  const uint32_t *blocks;    /* n_blocks * 4 words: cnt[4] as uint16, lo, hi */
  const uint32_t *super;     /* n_super * 4 words: C[c] + count before the superblock */
  const uint32_t *text;      /* text_words */
  const uint64_t *sa;        /* n_bases + 1 rows: pos | before<<32 | after<<40 | seg<<48 */
  const uint32_t *ftab;      /* jump tables of ftab_ks back to back, 4^k + 1 words each: the rows
                              * starting with k-mer c (numbered lexicographically, first base
                              * most significant) are [T[c], T[c+1]) */
  const uint32_t *seg_start; /* n_seg + 1 */
  const uint32_t *seg_ref;   /* n_seg */
  const uint32_t *seg_off;   /* n_seg */
  const uint32_t *chunk_seg; /* (n_bases >> 5) + 2 */
  const uint32_t *ctx;       /* NULL, or for libraries of >= 2^20 bases one word per suffix-array
                              * row: bits 0-15 the 8 bases left of the row's position (the nearest
                              * in the top two), bits 16-31 the bases 8..15 after it */
  const uint32_t *kbits;     /* NULL, or for libraries of at most 190 000 bases the presence bitmap of
                              * their 9-mers: 4^9 bits, bit c = 9-mer with code c (first base in
                              * the low two bits) occurs */
} mrg_index_view;
int mrg_index_get_view(const mrg_index *ix, mrg_index_view *view);

/* Exact-match dictionary of a library of at most 2^22 bases (read-only view, index lifetime; built on
 * the first request for a key length and uploaded by mrg_ctx_add_library): what answers the `-n 0` /
 * `-v 0` runs (RAP:577, :598, :688) for one-word reads in ONE 16-byte load.  Open addressing, 2^log2_slots
 * slots of two uint64:
 *   slots[2 i]      the 32 text bases from the slot's text position (2 bits per base, first base lowest)
 *   slots[2 i + 1]  low 32 bits: library entry; high 32 bits: bits 0-5 bases to the end of the
 *                   position's N-free segment (clamped to 63), bits 6-9 chain = how far behind its
 *                   home slot a key homed at slot i may sit (15 = overflowed: ask the FM index),
 *                   bit 10 occupied, bits 11-31 offset of the position in its entry
 * home slot of a key = ((uint32) of its first key_bases bases * 0x9E3779B1) >> (32 - log2_slots);
 * keys are inserted in text order, a position that an earlier one makes unreachable is left out, so
 * the first match along home, home + 1, ... home + chain is the lowest (entry, offset). */
typedef struct mrg_dict_view {
  const uint64_t *slots;
  uint32_t log2_slots;
  uint32_t key_bases;
  uint64_t n_keys;     /* positions stored */
  uint64_t n_overflow; /* HOME SLOTS whose chain overflowed (a count of homes: every further position of such a home is left to the FM index) */
} mrg_dict_view;
int mrg_index_get_dict(const mrg_index *ix, uint32_t key_bases, mrg_dict_view *view);

/* ------------------------------------------------------------------ *
 * Context: one per GPU.  Replaces the per-pass process spawn + .ebwt
 * load of RAP:643 / RAP:689 with libraries resident in HBM.
 * ------------------------------------------------------------------ */
int mrg_ctx_create(int device, mrg_ctx **out);
void mrg_ctx_destroy(mrg_ctx *ctx);
/* Upload one library; *lib_id is what mrg_pass_cfg.lib refers to. */
int mrg_ctx_add_library(mrg_ctx *ctx, const mrg_index *ix, int32_t *lib_id);
/* Tunables (all have defaults): "lds_budget" = most bytes of LDS a match
 * workgroup may spend on a staged library (occ blocks + packed text; 0 serves
 * every library from HBM/L2); "wstop" = interval width at which a seed search
 * stops narrowing and hands the occurrences to verification (0 = narrow to the
 * end of the piece; default 8); "ftab" = 1/0 use the k-mer jump table for the first k steps
 * of a seed search; "wide_rows" = seed intervals wider than this many rows are
 * verified cooperatively by the whole wave (default 64); "hint_min_len" / "hint_max_len" = the
 * caller's promise that every read of the NEXT mrg_cascade_run has a length in that range (defaults
 * 0 / 255 = unknown; reset to the defaults by every run): a pass whose length window excludes the
 * whole range is not launched, and a one-word batch that may hold reads under "split_min_len" is split
 * (below); the hints decide no result: every kernel reads d_lens; "kmer_filter" = 1/0 stage a small
 * library's 9-mer presence bitmap in LDS and skip the jump-table load of a seed piece whose
 * last 9 bases do not occur in the library (default 1); "ctx_wide_rows" = the same
 * threshold for libraries of >= 2^20 bases, whose cooperative path drops most rows by their
 * stored text context (default 32); "fuse" = 0 one launch per pass, 1 (default)
 * consecutive passes with at most one seed mismatch after the first launched pass share one
 * fused launch per run of small (bitmap-filtered) / large libraries, 2 = one group regardless of
 * library size, 3 = only the small-library runs are fused; "pair_seeds" = 1 (default) / 0: a pass
 * with two seed mismatches on a library of at most 4 Mbp searches reads of at least 15 nt (after
 * -5 / -3) through the six pairs of four anchors instead of three pigeonhole pieces (the pair tables
 * are built by mrg_ctx_add_library: set 0 BEFORE adding libraries to save their memory, or at any
 * time to run the piece search); "split_strata" = 1 (default) / 0: such a pass, when it does run the
 * piece search, is launched as strata 1-2 and stratum 3 separately; "stratum_rows" = 0 (default) /
 * 1 / 2: that piece search in stratum_kernel (rows compacted over the wave) for the last stratum /
 * every strata launch; "pair_big" = 5 (default) / 0 / 4..7: anchor length A of the pair tables of
 * libraries of >= 2^20 bases -- a one-mismatch pass inside a fused launch searches reads whose seed
 * region is 3A .. 4A - 1 bases through three anchor pairs instead of two short pigeonhole pieces
 * (tables built on the device the first time such reads are met; mrg_pass_stats.pair_anchor reports
 * A); "dict" = 1 (default) / 0: batches of one-word reads without an N mask run the dictionary
 * kernels for the passes that can (a pass without seed mismatches on a library of at most 4 Mbp: one
 * load of its exact-match dictionary per read, built by mrg_ctx_add_library while the option is 1);
 * "dict_key" = 16 (default) / 8..16: key length of those dictionaries (set BEFORE adding libraries;
 * reads shorter than the key take the FM index); "seed_units" = 1 (default) / 0: in such batches the runs
 * of passes with at most one seed mismatch after the first launch go through seed_kernel (libraries of at
 * most 4 Mbp searched with one policy as ONE index of their concatenation, built when a cascade first
 * plans it); "split_mixed" = 1 (default) / 0 and "split_min_len" = 16 (default: the reference's own length floor, trim_file.py:33; round 3: 20) / 0..32: a batch with more
 * than one word per read, an N mask, or (by the length hint) reads under split_min_len nt -- unless the hint
 * puts every read on one side: all longer than 32 nt, or all under split_min_len -- is split on the
 * device into the reads of split_min_len .. 32 nt without N, which run the cascade through the dictionary
 * kernels as the one-word batch they are, and the rest, which runs it through the FM kernels first -- two
 * cascades over disjoint lists adding to the same counters (mrg_pass_stats then names the kernels and times
 * of the second one, ms_rest the times of the first, n_launches counts both);
 * (round 4) "dict_max_bases" = 1 << 22 (default): libraries up to this size get an exact-match dictionary;
 * raise it (e.g. 1 << 30) BEFORE adding a large library that a pass searches without seed mismatch (mRNA
 * `-n 0`, runAnnotationPipeline.py:584/598) -- 16 B x 2..4 slots per base of HBM, skipped when less than that
 * + 8 GB is free -- and that pass becomes one 16-byte gather per read; "seed_impl" = -1 (default: seed_kernel
 * for a launch on small libraries, wave_seed_kernel with 96 registers for one on large libraries) / 0
 * seed_kernel / 1, 2 wave_seed_kernel (64-80 / 80-96 registers); "pair_impl" = 1 (default) / 0: the anchor-pair
 * search of one-word batches in pair_wave_kernel / stratum_kernel; "grid_pct" = 100 (default) / 1..100: every
 * cascade launch with this share of its workgroups (room for another cascade's launches on another stream);
 * (round 5) "fused_step" = 0 (default) / 1: for a caller that follows EVERY mrg_cascade_run* with an mrg_tally_run* on the
 * same stream: the cascade records no per-pass events (mrg_pass_stats.ms = 0: take the times from a run with the option
 * off) and d_pass_counts is written by the tally launch instead of by a launch of its own -- nothing then sits between
 * the launches of a step, which may be captured into a hipGraph;
 * "collapse_fast" = 1 (default) / 0: mrg_collapse_run takes its duplication-aware path for batches that fit it
 * (one-word reads without N, at most 29 nt) / always the general column-by-column sort;
 * "device_tables" = 1 (default) / 0: libraries added afterwards get the derived tables of a large library (>= 2^20
 * bases) filled on the device / built on the host and uploaded (mrg_ctx_library_check_tables);
 * (round 6) "pos_lists" = 1 (default) / 0, before mrg_ctx_add_library: a library with seed buckets also gets, for every
 * k-mer whose bucket overflows (an interspersed element, poly-A, a tandem motif: 10^2..10^5 rows), the k-mer's rows in
 * TEXT ORDER (fm_index.hpp: seed_pos_lists); "pos_scan" = 1 (default) / 0 at run time: a seed launch answers such a
 * seed by walking that list up to the first valid alignment (plus a 64-ary search of the suffix-sorted rows for an
 * exact occurrence) instead of verifying every row of the suffix interval -- same answers (runAnnotationPipeline.py:
 * 581-584 offers every read to every library, whatever the library holds);
 * "walk_cap" = 256 (default; 0..256): records a wave of wave_seed_kernel may leave behind its stream (reads whose seeds
 * meet repeats); beyond it such a seed is verified row by row as before round 6 (tests shrink it);
 * "long_lane" = 0 (default) / 1: in a batch of two words per read, the N-free reads of 33..63 nt take the FM kernels / go
 * with the one-word reads through the dictionary kernels' LONG instantiations (seeds from the first 32 bases, the second
 * word fetched where an alignment is verified; exact_dict_kernel and the FM kernels of that lane cannot see them: a length
 * is taken only when every pass either runs in a seed launch, keeps it out by its window, or -- pair_wave_kernel -- sees
 * at most 32 bases of it behind the trims or could not align it at all).  Same results (tests/test_gpu_split.py,
 * test_gpu_random_worlds.py); measured slower -- 1.10 against 0.67 ms for the 6.4 M reads of 33..40 nt of
 * `bench.py --workload varlen`: a long read's candidate costs two or three dependent text trips where a short one is
 * judged from its 16-byte row -- hence off;
 * "wide_rows_16", "round_large": see DESIGN.md. */
int mrg_ctx_set_option(mrg_ctx *ctx, const char *key, int64_t value);
int mrg_ctx_device_info(const mrg_ctx *ctx, int32_t *n_cu, uint64_t *hbm_bytes,
                        char *arch, uint32_t arch_cap);
/* What a resident library's derived structures hold (round 5; the bench line of an unfriendly library set reports them):
 * out4 = { text positions stored in its exact-match dictionary, HOME SLOTS whose chain overflowed (a count of
 * homes, not of positions: every further position of such a home is left to the FM fallback), log2 of the
 * dictionary's slots (0: no dictionary), k of its seed buckets (0: none) }. */
int mrg_ctx_library_stats(const mrg_ctx *ctx, int32_t lib, uint64_t *out4);
/* Self-check of a resident library's derived tables (round 5).  With "device_tables" = 1 (default) a library of at
 * least 2^20 bases gets its jump tables, row context, wide rows, seed buckets (csrc/libtables.hip) and exact-match
 * dictionary (csrc/dictbuild.hip) filled ON THE DEVICE from the suffix-array rows and the packed text -- what replaces
 * the `.ebwt` load of runAnnotationPipeline.py:643 at the start of a run; the index file stores none of them.  This call
 * rebuilds the first four on the host (fm_index.cpp) and compares them with the device arrays word by word:
 * mismatches4 = { jump tables, row context, wide rows, seed buckets (round 6: with the headers and the position lists
 * of the overflowing k-mers) }, UINT64_MAX for a table the library does not have.  (The dictionary's layout depends on who fills it; what a lookup finds does not: tests/test_gpu_dictbuild.py.)
 * Slow -- seconds of host time for a 137 Mbp library: tests and bench gates call it, a run does not. */
int mrg_ctx_library_check_tables(mrg_ctx *ctx, int32_t lib, const mrg_index *index, uint64_t *mismatches4);
/* A context is used by one host thread at a time (calls on it are serialised by the caller).  It keeps
 * one device scratch arena, grown on demand by mrg_collapse_run (40 B per raw read + 64 MB) and
 * mrg_list_best_count and reused by later calls; this frees it (after synchronising the device), e.g.
 * once the one collapse of a run is done. */
int mrg_ctx_release_scratch(mrg_ctx *ctx);

/* One alignment pass = one bowtie command line of RAP:577-599 / RAP:688. */
typedef struct mrg_pass_cfg {
  int32_t lib;          /* library id from mrg_ctx_add_library */
  int32_t seed_len;     /* -l (28) for -n mode; >= MRG_MAX_WORDS*32 for -v mode */
  int32_t max_mm_seed;  /* -n N, or V for -v mode */
  int32_t max_mm_total; /* 2 for -n mode (-e 70 at Q40), V for -v mode */
  int32_t trim5;        /* -5 */
  int32_t trim3;        /* -3 */
  int32_t min_len;      /* length filter of RAP:543-554: process min_len<=len<=max_len */
  int32_t max_len;      /* (255 = no upper bound: the reference's filters are `< 26`, `> 25` or none) */
  int32_t poly_t;       /* 1 = pass 3 (RAP:664-686): need T{3,}$, strip all 3' T, >=11 nt */
  int32_t reserved;
} mrg_pass_cfg;

typedef struct mrg_pass_stats {
  uint64_t processed;  /* "# reads processed" (RAP:9-18) */
  uint64_t aligned;    /* "# reads with at least one reported alignment" */
  uint64_t steps;      /* FM backward-extension (LF) steps executed */
  uint64_t candidates; /* seed occurrences verified against the text */
  uint64_t lookups;    /* k-mer jump-table loads (each replaces k LF steps) */
  float ms;            /* device time of the pass (the reference's cpuTime); a fused launch is
                          charged to its first pass, the other passes of the group report 0 */
  uint32_t lds_bytes;  /* library bytes staged in LDS for this pass (0 = served from HBM/L2) */
  uint32_t lds_mode;   /* 0 nothing, 1 occ blocks, 2 occ blocks + text, 3 text only: names the
                          match_kernel<W, blocks, text> instantiation that ran; 4 = the pass ran
                          inside a fused launch (fused_kernel<W>, only its 9-mer bitmap in LDS);
                          5 / 6 = stratum_kernel<W> with / without the packed text in LDS;
                          7 = exact_dict_kernel (one slot load of the library's exact-match
                          dictionary per read: steps = 0, lookups = slot / table loads, candidates =
                          slots / rows compared); 8 / 9 = seed_kernel<false, 8> / seed_kernel<true, 6>
                          (the pass rode, as a unit or a member of a unit, in the seed launch of its
                          group's first pass -- 9: a launch with a library that has seed buckets; steps = 0,
                          lookups = jump-table / bucket / slot loads and candidates = rows of the unit,
                          reported with the unit's first pass) */
  uint32_t group;      /* index of the first pass of the launch this pass ran in (itself when it
                          had a launch of its own) */
  uint32_t n_launches; /* kernel launches that carried this pass: 1, or 2 for a 2-mismatch pass split
                          into strata 1-2 and stratum 3; 0 for a pass that was not launched or rode
                          in the fused launch of its group's first pass; a split batch ("split_mixed")
                          counts the launches of both of its cascades */
  uint32_t kbits_log2; /* log2 of the bits of the 9-mer presence bitmap the pass filtered seed
                          pieces with (18 = the library's full bitmap, 13..17 = folded for a
                          fused launch, 0 = no filter) */
  uint32_t pair_anchor; /* != 0, pass with ONE seed mismatch (fused launch, large library): reads of
                          3 x pair_anchor .. 4 x pair_anchor - 1 seed bases were searched through the
                          three pairs of three anchors, the others through the pigeonhole pieces.
                          != 0, TWO seed mismatches: reads of at least 4 x pair_anchor seed bases
                          were searched through the six pairs of four anchors of that many bases
                          (lookups = pair lookups, candidates = their rows, no LF steps), reads
                          of at least 4 x (pair_anchor - 1) through anchors one base shorter;
                          still shorter reads went through the stratum-first pigeonhole pieces */
  float ms_rest;       /* a split batch ("split_mixed"): device time of this pass in the FIRST cascade (the
                          long reads, reads with N, very short reads: FM kernels); ms and the kernel fields
                          above describe the second one (the one-word reads); 0 when the batch was not split */
  uint32_t variant;    /* which instantiation behind lds_mode 8 / 9 (round 4): 0 = seed_kernel (tiles of 256 reads,
                          three barriers per tile), 1 = wave_seed_kernel<false, 8> / <true, 6> (every wave on its
                          own: a read with nothing left to look up is finished on the spot, the others are parked
                          and worked off 64 at a time), 2 = wave_seed_kernel<false, 6> / <true, 5> (more registers);
                          + 4 = the launch's input list carried its reads (16-byte entries written by the seed
                          launch in front of it: seed_kernel<.., .., true>);
                          + 8 (round 6) = the LONG instantiation: the batch's reads of 33..63 nt ride this launch too
                          (seeds from their first 32 bases, the second word compared where an alignment is verified).
                          lds_mode 7 (exact_dict_kernel): 16 (round 6) = the streaming launch gave every workgroup one
                          contiguous stretch of the batch instead of every gridDim-th chunk of 4096 reads (taken when
                          more than 2 % of the (workgroup, trip) slots of the chunked run would stay empty).
                          lds_mode 11 (round 4) = pair_wave_kernel: the anchor-pair search of a 2-mismatch pass for
                          one-word reads without N, items and rows compacted over the wave */
  uint32_t reserved;
} mrg_pass_stats;

/* Bytes of device workspace mrg_cascade_run needs for n reads (three survivor lists of n + 2^23 entries of 16 bytes --
 * the seed launches' lists carry their reads --, segment counts, counters, and since round 6 the 67 MB of records of the
 * reads wave_seed_kernel answers behind its stream: 48 B per read + 470 MB).  A workspace belongs to ONE cascade at a
 * time: cascades of one context on several streams each bring their own. */
int mrg_cascade_workspace_bytes(uint64_t n, uint64_t *bytes);

/*
 * The cascade (RAP:636-705): passes run in order; a read is offered to pass i
 * only if no earlier pass claimed it (writeSeqToAnnot RAP:543-554 +
 * updateAnnotDic RAP:341-345).  Outputs, one entry per read:
 *   d_pass_id  -1 = unannotated, else index of the claiming pass
 *   d_ref_id   library entry index (order of mrg_index_name), -1 if unannotated
 *   d_pos      0-based offset in the entry of the first untrimmed base
 *              (SAM POS-1), -1 if unannotated
 *   d_mm       mismatches of the reported alignment (0 if unannotated)
 * Tie rule (bowtie's own choice is RNG-driven): fewest mismatches, then lowest
 * entry index, then lowest offset.
 * d_pass_counts: 2*n_pass uint64 (processed, aligned per pass) written on
 * device so a sharded run can all-reduce them without a host round trip.
 * Asynchronous on `stream`; call mrg_cascade_stats after synchronising.
 */
int mrg_cascade_run(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read,
                    const uint8_t *d_lens, const uint64_t *d_nmask, uint64_t n,
                    const mrg_pass_cfg *passes, uint32_t n_pass, int8_t *d_pass_id,
                    int32_t *d_ref_id, int32_t *d_pos, uint8_t *d_mm,
                    uint64_t *d_pass_counts, void *d_workspace,
                    uint64_t workspace_bytes, void *stream);
/*
 * The four assignment arrays of mrg_cascade_run as ONE 32-bit word per read ("4 B packed
 * assignment out"), for callers that move assignments over PCIe: 4 bytes per read instead of 10.
 *   bits 28-31  claiming pass + 1 (0 = unannotated)
 *   bits 26-27  mismatches, saturating at 3
 *   bits  8-25  library entry, saturating at MRG_PACKED_REF_SAT (2^18 - 1)
 *   bits  0-7   offset of the alignment in the entry, saturating at MRG_PACKED_POS_SAT (255)
 * Every miRNA / hairpin / tRNA alignment fits (what the isomiR, A-to-I, GFF and tRF consumers read);
 * a saturated field says "ask the full arrays" (an mRNA entry past the 262143rd, an offset deep inside
 * a transcript).  n_pass must be at most 15.  Asynchronous on `stream`.
 */
#define MRG_PACKED_REF_SAT 0x3FFFFu
#define MRG_PACKED_POS_SAT 0xFFu
int mrg_pack_assignments(mrg_ctx *ctx, const int8_t *d_pass_id, const int32_t *d_ref_id, const int32_t *d_pos,
                         const uint8_t *d_mm, uint64_t n, uint32_t *d_packed, void *stream);
/* The cascade with the packed word as its ONLY per-read output (every kernel writes d_packed[r]
 * instead of the four arrays: 4 bytes out per read, what SURVEY.md 8d's byte accounting assumes), and
 * the two tallies reading it.  Same semantics otherwise; the saturation rules above apply (the count
 * tally needs entries of the miRNA library only, the edit tally their offsets too: always exact). */
int mrg_cascade_run_packed(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read, const uint8_t *d_lens,
                           const uint64_t *d_nmask, uint64_t n, const mrg_pass_cfg *passes, uint32_t n_pass,
                           uint32_t *d_packed, uint64_t *d_pass_counts, void *d_workspace, uint64_t workspace_bytes,
                           void *stream);
int mrg_tally_run_packed(mrg_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_quant, uint64_t n, uint32_t n_samples,
                         uint32_t n_mirna, uint32_t n_pass, int32_t canon_pass, int32_t isomir_pass, uint64_t *d_counts,
                         void *stream);
int mrg_edit_tally_run_packed(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read, const uint8_t *d_lens,
                              const uint64_t *d_nmask, const uint32_t *d_packed, const uint32_t *d_quant,
                              const uint8_t *d_keep, const uint32_t *d_remap, uint64_t n, uint32_t n_samples,
                              uint32_t n_bins, int32_t lib, int32_t canon_pass, int32_t isomir_pass, int32_t isomir_trim5,
                              uint32_t flank5, uint32_t flank3, uint32_t from_base, uint32_t to_base, uint64_t *d_counts,
                              void *stream);
/*
 * The cascade for reads of ANY length (round 5).  writeSeqToAnnot (RAP:543-554) writes every unannotated read
 * into a pass's FASTA whatever its length -- the length filters of RAP:574 are `< 26`, `> 25` or none -- and bowtie
 * aligns it end to end; the batches of mrg_cascade_run describe a length in one byte.  A host sends the reads
 * beyond 255 nt (an untrimmed long-cycle run, a read-through: a handful per sample) here, in the RAGGED form:
 *   d_words     read r = d_words[d_word_off[r] .. d_word_off[r + 1]), ceil(len / 32) words, 32 bases per word,
 *               base i of the read in bits [2 (i % 32), 2 (i % 32) + 1] of word i / 32 (as in the packed batches)
 *   d_nmask     the same shape, or NULL when no read has an N
 *   d_word_off  n + 1 offsets in words;  d_lens  n lengths in bases (any length < 2^31; short reads are fine too)
 * Same passes, same policy fields and the same outputs as mrg_cascade_run -- fewest mismatches, then lowest entry,
 * then lowest offset; -1 / -1 / -1 / 0 for an unannotated read -- with two differences: a pass's max_len of 255 or
 * more means "no upper bound" (the reference has none), and trim5 may be any non-negative number.  One wave per read
 * over the library's FM index (csrc/long_reads.hip): written for any length, not for speed.
 *   d_pass_counts  NULL, or the 2 * n_pass uint64 (processed, aligned per pass) of the batch's mrg_cascade_run:
 *                  the long reads' counts are ADDED to them (run it after that call, on the same stream)
 *   stats          NULL, or n_pass host entries: processed / aligned / steps / candidates / lookups / ms are ADDED
 *                  to what they hold (hand in mrg_cascade_stats' array, or a zeroed one)
 * Synchronises `stream`.
 */
int mrg_cascade_run_long(mrg_ctx *ctx, const uint64_t *d_words, const uint64_t *d_nmask, const uint64_t *d_word_off,
                         const uint32_t *d_lens, uint64_t n, const mrg_pass_cfg *passes, uint32_t n_pass,
                         int8_t *d_pass_id, int32_t *d_ref_id, int32_t *d_pos, uint8_t *d_mm, uint64_t *d_pass_counts,
                         mrg_pass_stats *stats, void *stream);
/* Number of cascade runs this context has launched (a caller that reads statistics later can tell
 * whether they are still those of its own run). */
int mrg_cascade_run_id(const mrg_ctx *ctx, uint64_t *run_id);
/* Synchronises `stream` of the LAST run and returns its per-pass statistics. */
int mrg_cascade_stats(mrg_ctx *ctx, mrg_pass_stats *stats, uint32_t n_pass);

/*
 * The tally (SUM:34-66): per sample s and read r with quant[r][s] != 0:
 *   trimmedUniq[s]++ ; cat[pass][s] += quant ; cat[n_pass][s] (remReads) for -1
 *   pass == canon_pass : mir_quant[ref][s] += q ; mir_iscan[ref][s] += q
 *   pass == isomir_pass: mir_quant[ref][s] += q
 * d_counts layout (uint64, caller zeroes it):
 *   [0, M*S) mir_quant | [M*S, 2*M*S) mir_iscan |
 *   [2*M*S, 2*M*S + (n_pass+1)*S) category totals | then S trimmedUniq
 */
int mrg_tally_counts_len(uint32_t n_mirna, uint32_t n_samples, uint32_t n_pass,
                         uint64_t *len);
int mrg_tally_run(mrg_ctx *ctx, const int8_t *d_pass_id, const int32_t *d_ref_id,
                  const uint32_t *d_quant, uint64_t n, uint32_t n_samples,
                  uint32_t n_mirna, uint32_t n_pass, int32_t canon_pass,
                  int32_t isomir_pass, uint64_t *d_counts, void *stream);

/* ------------------------------------------------------------------ *
 * Multi-GPU: one process per GPU, the globally collapsed read set cut into contiguous shards,
 * every library replicated, and ONE collective per batch: the sum over ranks of the fused count
 * vector [mir_quant | mir_iscan | category totals | trimmedUniq | per-pass processed, aligned]
 * (mrg_tally_run's d_counts followed by mrg_cascade_run's d_pass_counts, all uint64).  The
 * reference has no counterpart (it is a single process); `filter` (utils/filter.py:7-13) is
 * non-linear and must run on the reduced vector.  RCCL over xGMI, bound at run time (dlopen): a
 * host that already has an RCCL mapped (PyTorch) shares it.
 *   mrg_comm_unique_id  rank 0 fills 128 bytes and hands them to the other ranks out of band
 *                       (a file, a pipe, an environment variable)
 *   mrg_comm_init       collective over the `world` processes, each with its own context / GPU
 *   mrg_allreduce       in place, uint64 sum, asynchronous on `stream`; a no-op for world == 1
 * ------------------------------------------------------------------ */
#define MRG_COMM_ID_BYTES 128
int mrg_comm_unique_id(void *id128);
int mrg_comm_init(mrg_ctx *ctx, const void *id128, int32_t rank, int32_t world);
int mrg_allreduce(mrg_ctx *ctx, uint64_t *d_buf, uint64_t n, void *stream);
int mrg_comm_destroy(mrg_ctx *ctx);

/*
 * A-to-I position tally: the per-read part of A2IEditing (utils/writeDataToCSV.py:145-229) and
 * judgeAllign (:35-69) on the alignments the cascade already produced, for the reads claimed by
 * the exact-miRNA pass (canon_pass) or the isomiR pass (isomir_pass, whose reported position is
 * that of the first base after its `-5` trim: isomir_trim5).  A library entry is flank5 + mature
 * + flank3 bases (runAnnotationPipeline.py:413: 2 and 6); a read is kept when judgeAllign keeps
 * it -- starts at most 1 nt after the mature sequence, at most 1 mismatch and enough matches over
 * the window that function compares -- and, if d_keep is given, d_keep[r] != 0 (the host's
 * genome-uniqueness / RPM selection, :1251-1287, :1293-1300).  Per output bin b (= entry, or
 * d_remap[entry] for merged miRNA names; n_bins of them) and sample s, uint64, caller zeroes:
 *   d_counts[(b * S + s) * 3 + 0]  count_true: summed counts of the kept reads
 *   d_counts[(b * S + s) * 3 + 1]  seq_true:   number of kept reads with a non-zero count
 *   d_counts[(b * S + s) * 3 + 2]  canonical:  counts of kept reads that are substrings of the mature sequence
 *   d_counts[n_bins * S * 3 + (b * 32 + i) * S + s]  counts of kept reads showing `to_base` where
 *       the mature sequence has `from_base` at (0-based) position i < mature length - 5 (:147,:168)
 * Bases: A=0 C=1 G=2 T=3 (A-to-I reads as A -> G: from_base 0, to_base 2).
 */
int mrg_edit_counts_len(uint32_t n_bins, uint32_t n_samples, uint64_t *len);
int mrg_edit_tally_run(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read,
                       const uint8_t *d_lens, const uint64_t *d_nmask, const int8_t *d_pass_id,
                       const int32_t *d_ref_id, const int32_t *d_pos, const uint32_t *d_quant,
                       const uint8_t *d_keep, const uint32_t *d_remap, uint64_t n, uint32_t n_samples,
                       uint32_t n_bins, int32_t lib, int32_t canon_pass, int32_t isomir_pass,
                       int32_t isomir_trim5, uint32_t flank5, uint32_t flank3, uint32_t from_base,
                       uint32_t to_base, uint64_t *d_counts, void *stream);

/*
 * Best stratum of every read against ONE library, forward strand only: d_best_mm[r] = fewest
 * mismatches of a valid alignment (255 = the read does not align), d_count[r] = number of
 * alignments reaching it (saturates at 255; also 255 when a seed is too repetitive to be
 * walked).  Replaces the genome bowtie runs of the -ai path (utils/writeDataToCSV.py:1263
 * `-n 1 -f -a -3 2` and :1488 `-n 0 -f -a -3 2`), which only ask "is the best hit unique"
 * (:1277-1287) and "does it align" (:1491-1496); the host trims the 3' 2 nt, submits each
 * read and its reverse complement, and sums over the chromosome libraries.
 */
int mrg_count_best(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read,
                   const uint8_t *d_lens, const uint64_t *d_nmask, uint64_t n, int32_t lib,
                   int32_t seed_len, int32_t max_mm_seed, int32_t max_mm_total,
                   uint8_t *d_best_mm, uint8_t *d_count, void *stream);

/*
 * Every alignment of each read's best stratum against one library: what `bowtie -a --best
 * --strata` prints (flags of every cascade command line, runAnnotationPipeline.py:577-599) and
 * parseAlignment3 (RAP:41-52) collects for the tRF tables (RAP:657-660, :698-701).  Two sweeps so
 * the caller owns the output buffers:
 *   mrg_list_best_count  d_best_mm[n] (255 = unaligned), d_offsets[n+1] = exclusive prefix of
 *                        the stratum sizes, *total = d_offsets[n]; synchronises the stream;
 *   mrg_list_best_fill   alignment k of read r at d_ref/d_pos[d_offsets[r] + k] (entry index,
 *                        0-based offset in the entry; order within a read unspecified); slots
 *                        >= cap are not written.
 */
int mrg_list_best_count(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read,
                        const uint8_t *d_lens, const uint64_t *d_nmask, uint64_t n, int32_t lib,
                        int32_t seed_len, int32_t max_mm_seed, int32_t max_mm_total,
                        uint8_t *d_best_mm, uint64_t *d_offsets, uint64_t *total, void *stream);
int mrg_list_best_fill(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read,
                       const uint8_t *d_lens, const uint64_t *d_nmask, uint64_t n, int32_t lib,
                       int32_t seed_len, int32_t max_mm_seed, int32_t max_mm_total,
                       const uint8_t *d_best_mm, const uint64_t *d_offsets, uint64_t cap,
                       int32_t *d_ref, int32_t *d_pos, void *stream);

/*
 * Host-buffer convenience for a caller without its own device allocator (the
 * ctypes stub of INTEGRATION.md): H2D, cascade, tally, D2H in one call.
 * counts may be NULL (then quant/n_samples are ignored).
 */
int mrg_annotate_host(mrg_ctx *ctx, const uint64_t *reads, uint32_t words_per_read,
                      const uint8_t *lens, const uint64_t *nmask, uint64_t n,
                      const mrg_pass_cfg *passes, uint32_t n_pass, int8_t *pass_id,
                      int32_t *ref_id, int32_t *pos, uint8_t *mm,
                      mrg_pass_stats *stats, const uint32_t *quant,
                      uint32_t n_samples, uint32_t n_mirna, int32_t canon_pass,
                      int32_t isomir_pass, uint64_t *counts);

/* mrg_cascade_run_long with HOST buffers (same ragged form; stats may be NULL, else it is added to). */
int mrg_annotate_long_host(mrg_ctx *ctx, const uint64_t *words, const uint64_t *nmask, const uint64_t *word_off,
                           const uint32_t *lens, uint64_t n, const mrg_pass_cfg *passes, uint32_t n_pass,
                           int8_t *pass_id, int32_t *ref_id, int32_t *pos, uint8_t *mm, mrg_pass_stats *stats);

/* ------------------------------------------------------------------ *
 * Ingest: FASTQ -> packed reads (host), raw reads -> unique reads with
 * per-sample counts (device).  Replaces trim_file with `-ad none`
 * (utils/trim_file.py:89-134: 3' quality trimming at Q10, 16-nt minimum)
 * and quantReads (utils/quantReads.py:3-24).
 * ------------------------------------------------------------------ */
typedef struct mrg_fastq mrg_fastq;
typedef struct mrg_fastq_info {
  uint64_t n_total;        /* records read ("totalReads") */
  uint64_t n_kept;         /* records kept after trimming ("trimmedReads") */
  int32_t phred;           /* 33 or 64, as trim_file.py:104-106 reports it */
  uint32_t words_per_read; /* 1, 2, 4 or 8 */
  uint32_t max_len;
  int32_t has_n;
  uint64_t n_long;         /* kept reads longer than 255 nt (MRG_MAX_WORDS words, one length byte): not among n_kept, not
                              in the packed batch; read them with mrg_fastq_long_read / mrg_fastq_copy_long and annotate
                              them with mrg_cascade_run_long (the reference accepts any length) */
} mrg_fastq_info;
/* Plain or gzip FASTQ.  qual_cutoff 10 and min_len 16 are the reference's values
 * (trim_file.py:30,33).  adapter = the `-ad` value after __main__.py:123-127: NULL or "none",
 * "+N" (UnconditionalCutter, trim_file.py:34-35) or one or more comma-separated 3' adapter
 * sequences (AdapterCutter at error rate 0.12, trim_file.py:38-41).  threads = trimming worker
 * threads (the reference's `-cpu` worker processes, trim_file.py:93-98); <= 0 picks one per
 * hardware thread, at most 32.  One thread inflates and splits records, the workers trim. */
int mrg_fastq_load(const char *path, int32_t qual_cutoff, int32_t min_len, const char *adapter,
                   int32_t threads, mrg_fastq **out);
/* One sample read by SEVERAL readers (round 5: `--gpus N` on a single FASTQ file; the reference ingests a sample in one
 * process, __main__.py:289-314): reader `part` of `n_parts`.  A plain file is cut into n_parts byte ranges at record
 * starts (a line that starts with '@' whose line after next starts with '+': every reader finds the same cuts) and
 * each reader reads its own; a gzip file cannot be entered in the middle: every reader inflates all of it and takes
 * every n_parts-th block of records (trimming, adapter search and packing are shared out, the inflate is not).  The
 * handle's n_total / n_kept / long reads are those of the share (their sums over the parts are the file's); the
 * quality base is the one the FILE's first record decides, whatever the part; info.phred is 0 unless part == 0. */
int mrg_fastq_load_part(const char *path, int32_t qual_cutoff, int32_t min_len, const char *adapter,
                        int32_t threads, int32_t part, int32_t n_parts, mrg_fastq **out);
/* cutadapt's 3' adapter search on one upper-case read (the AdapterCutter step above, exposed
 * for callers that trim outside a FASTQ file and for the tests): out6 = found, read_start
 * (where the read is cut), read_stop, adapter_stop, matches, errors. */
int mrg_adapter_locate(const char *adapter, const char *read, double max_error_rate,
                       int32_t min_overlap, int32_t *out6);
int mrg_fastq_get_info(const mrg_fastq *fq, mrg_fastq_info *info);
/* i-th over-long read (upper-case ASCII, valid until mrg_fastq_free). */
int mrg_fastq_long_read(const mrg_fastq *fq, uint64_t i, const char **seq);
/* The info.n_long over-long reads packed into the ragged form of mrg_cascade_run_long (same two-call protocol as
 * mrg_pack_reads_ragged: words == NULL fills word_off[n_long + 1] and lens only).  In file order, duplicates included:
 * the host collapses them (by their text, mrg_fastq_long_read) before it annotates them. */
int mrg_fastq_copy_long(const mrg_fastq *fq, uint64_t *words, uint64_t *nmask, uint64_t *word_off, uint32_t *lens,
                        int *has_n);
/* Copy the packed reads out, widened to words_per_read words (>= the file's own);
 * nmask may be NULL when has_n is 0. */
int mrg_fastq_copy(const mrg_fastq *fq, uint32_t words_per_read, uint64_t *words, uint8_t *lens,
                   uint64_t *nmask);
void mrg_fastq_free(mrg_fastq *fq);

/*
 * The same ingest ON THE DEVICE for raw text (round 3): the host reads the file, cuts it into blocks
 * of whole records (mrg_fastq_block_cut gives the offset of the last record boundary of a buffer:
 * buf[0, cut) is a block, the rest is carried over; at_eof != 0 takes everything) and uploads them;
 * mrg_fastq_parse_device splits the records, applies the 3' quality rule (phred = 33 or 64, as the
 * first record's qualities decide: trim_file.py:104-110), the `-ad +N` cutter (cut > 0 drops the
 * first `cut` bases, cut < 0 the last -cut; 0 = `-ad none`), the minimum length and the 2-bit packing
 * -- the packed reads never exist on the host.  Kept reads come out in file order:
 * d_words[w * cap + i], d_lens[i], d_nmask[w * cap + i] (may be NULL; info.has_n then says whether
 * one was needed).  Strict four-line records only ('\n' or "\r\n", final newline optional):
 * info.status != 0 (1-3: record info.bad_record, 1-based, is ill-formed -- not a '@' header or a
 * blank line, no '+' line, sequence and quality of different length; 4: the line count is not a
 * multiple of four; 5: more kept reads than `cap`) means nothing was packed: use mrg_fastq_load,
 * which also accepts blank lines between records and words the errors as the reference's loader
 * does.  info.n_long = kept reads longer than min(32 * words_per_read, 255) bases (not packed: call again
 * with more words, up to MRG_MAX_WORDS).  Adapter SEQUENCES (`-ad illumina`): mrg_fastq_parse_device_ad below.
 * Synchronises `stream`; at most 2^31 - 2 bytes per call.
 */
typedef struct mrg_fastq_device_info {
  uint64_t n_records;  /* records in the block ("totalReads") */
  uint64_t n_kept;     /* reads packed */
  uint64_t n_long;     /* kept by the rules but longer than the words offered */
  uint64_t bad_record;
  uint32_t max_len;    /* longest packed read */
  int32_t has_n;
  int32_t status;
  int32_t reserved;
} mrg_fastq_device_info;
int mrg_fastq_block_cut(const char *buf, uint64_t len, int32_t at_eof, uint64_t *cut);
/* (round 4) mrg_fastq_parse_device_ad: the same with adapter SEQUENCES -- `adapters` = the `-ad` value after
 * __main__.py:123-127 resolved its aliases, "SEQ[,SEQ...]" (at most 4 of at most 64 bases; NULL or "" = none):
 * cutadapt's 3' search (`-a`, error rate 0.12, minimum overlap 3: trim_file.py:30-41), one thread per read,
 * after the quality trim and the cutter; the adapter with the most matched bases cuts the read. */
int mrg_fastq_parse_device_ad(mrg_ctx *ctx, const char *d_text, uint64_t n_bytes, int32_t phred, int32_t qual_cutoff,
                              int32_t min_len, int32_t cut, const char *adapters, uint32_t words_per_read, uint64_t cap,
                              uint64_t *d_words, uint8_t *d_lens, uint64_t *d_nmask, mrg_fastq_device_info *info, void *stream);
int mrg_fastq_parse_device(mrg_ctx *ctx, const char *d_text, uint64_t n_bytes, int32_t phred, int32_t qual_cutoff,
                           int32_t min_len, int32_t cut, uint32_t words_per_read, uint64_t cap, uint64_t *d_words,
                           uint8_t *d_lens, uint64_t *d_nmask, mrg_fastq_device_info *info, void *stream);

/*
 * Parallel inflate of a `.fastq.gz` sample (round 4): the reference reads gzip samples through ONE
 * inflate stream (parseArgument.py:32, __main__.py:289-314, trim_file.py:89-134).  mrg_gz_open maps the
 * file and inflates it with `threads` workers (block starts found by trial, the 32 KB window in front of
 * a chunk kept symbolic until the chunk in front is done: csrc/pgzip.cpp); mrg_gz_read returns the next
 * bytes of the text in order, exactly what gzread returns (*got == 0: end of file), checking every
 * member's CRC-32 and length.  threads <= 1, plain files and files too small to cut take zlib's own
 * reader.  mrg_fastq_load uses the same reader.  One reader per handle; not thread-safe.
 */
typedef struct mrg_gz mrg_gz;
int mrg_gz_open(const char *path, int32_t threads, mrg_gz **out);
int mrg_gz_read(mrg_gz *gz, void *buf, uint64_t len, uint64_t *got);
/* parallel != NULL: 1 when the parallel reader serves the file; merged != NULL: chunk starts dropped so far */
int mrg_gz_info(const mrg_gz *gz, int32_t *parallel, uint64_t *merged);
void mrg_gz_close(mrg_gz *gz);

/*
 * The compact wire form of a collapsed read set, for callers whose unique reads live on the HOST (the
 * reference's seqDic after quantReads.py:3-24): 6.5 bytes per 22-nt read over PCIe instead of the 13 of
 * the arrays (8-byte word + length + 32-bit count) -- what bounds a host-resident pipeline is the upload.
 *   runs       HOST array of n_runs (length, count) uint32 pairs: the reads come grouped by length
 *              (groups in any order, at most 64 of them; N-free reads of at most 32 nt) -- this replaces
 *              the length array
 *   d_bits     the reads as a bit stream of 64-bit words: read j of a run of L-base reads is the 2 L bits
 *              from bit 2 L j of the run's words (little-endian bit order: bit b of the run = bit b % 64 of
 *              word b / 64; base k of a read in its bits 2k, 2k+1 as in the packed word); every run starts
 *              a new word.  n_words = the words given: the runs' words + at least one of padding
 *   d_quant8   n x n_samples bytes, the per-sample counts; 255 = the count is in the escape list
 *              (NULL: no counts are expanded, d_quant is not written)
 *   d_esc      n_esc pairs of uint32: (flat index into the n x n_samples counts, count)
 * widened on the device into the arrays mrg_cascade_run / mrg_tally_run take: d_reads [1][n], d_lens [n],
 * d_quant [n][n_samples] (16-byte aligned).  Asynchronous on `stream`; the runs are read before the
 * call returns.
 */
int mrg_expand_compact(mrg_ctx *ctx, const uint64_t *d_bits, uint64_t n_words, const uint32_t *runs, uint32_t n_runs,
                       const uint8_t *d_quant8, const uint32_t *d_esc, uint64_t n_esc, uint64_t n,
                       uint32_t n_samples, uint64_t *d_reads, uint8_t *d_lens, uint32_t *d_quant, void *stream);

/*
 * Collapse n raw reads (device arrays, layout as for mrg_cascade_run; d_sample gives the
 * sample of each read or is NULL for one sample) into unique reads:
 *   d_u_reads [words_per_read][cap], d_u_lens [cap], d_u_nmask ([..][cap] or NULL),
 *   d_quant [n_unique][n_samples] (uint32), d_len_hist [256][n_samples] (uint64, the
 *   reference's readLengthDic), *n_unique on the host.  cap >= n is always enough.
 * max_len is a hint nobody needs any more (the call reads the lengths: its histogram is an output anyway); a batch
 * of one-word reads without N of at most 29 nt (2 max_len + sample bits <= 58, at most 16 distinct lengths) takes
 * the duplication-aware path (csrc/collapse.hip: copies of a sequence are merged in LDS before anything is
 * partitioned), everything else a stable radix sort of read ids column by column; a sample id >= n_samples is
 * MRG_ERR_ARG.  Uniques come out ordered by (length, bases); the call synchronises `stream`.
 */
int mrg_collapse_run(mrg_ctx *ctx, const uint64_t *d_reads, uint32_t words_per_read,
                     const uint8_t *d_lens, const uint64_t *d_nmask, const uint16_t *d_sample,
                     uint64_t n, uint32_t n_samples, uint32_t max_len, uint64_t cap,
                     uint64_t *d_u_reads, uint8_t *d_u_lens, uint64_t *d_u_nmask, uint32_t *d_quant,
                     uint64_t *d_len_hist, uint64_t *n_unique, void *stream);

/*
 * mapped.csv / unmapped.csv (utils/writeDataToCSV.py:582-619, :1172-1188) streamed from the
 * columnar HOST arrays the cascade produced instead of the reference's dict of dicts:
 *   row = uniqueSequence,annotFlag,<slot 1>,...,<slot n_slots>,<count of sample 1>,...
 * mapped != 0: the reads with pass_id >= 0, annotFlag 1, slot pass_id + 1 = the entry's name
 * (names[names_off[pass] + ref_id], names_off has n_slots + 1 entries); mapped == 0: the reads
 * with pass_id < 0, annotFlag 0, all slots empty.  reads is SoA with `stride` elements between
 * the words of a read (a slice of a larger array is fine); header (may be NULL) is written first
 * unless append != 0.  Rows come out in array order.  *rows = rows written.
 */
int mrg_write_read_table(const char *path, int32_t mapped, const char *header, int32_t append,
                         const uint64_t *reads, uint32_t words_per_read, uint64_t stride,
                         const uint8_t *lens, const uint64_t *nmask, uint64_t n, const int8_t *pass_id,
                         const int32_t *ref_id, const uint32_t *quant, uint32_t n_samples,
                         uint32_t n_slots, const char *const *names, const uint64_t *names_off,
                         uint64_t *rows);

/*
 * isomirs.csv and isomirs.samples.csv (utils/writeDataToCSV.py:1090-1170, over the grouping of :588-606) from the same
 * arrays (round 6): the reads claimed by canon_pass ("exact miRNA", slot 1) and isomir_pass ("isomiR miRNA", slot 9) are
 * grouped by group_of_entry[ref_id] -- the caller's map from a miRNA library entry to its name with the ".SNP..." suffix
 * stripped (:599-600), n_groups names in group_names --, groups in the order their first read appears, a group's isomiR
 * reads in array order.  filtered[s] = the sample's mirnaReadsFiltered (the RPM divisor).  Values are formatted as Python 2's
 * str(float) and the entropies are added in the reference's order: the files equal report.write_isomir_tables' byte for
 * byte (tests/test_report_tables.py).  header1 / header2: the two header lines.  *rows = rows of isomirs.csv.
 */
int mrg_write_isomir_tables(const char *isomirs_path, const char *samples_path, const char *header1, const char *header2,
                            const uint64_t *reads, uint32_t words_per_read, uint64_t stride, const uint8_t *lens,
                            const uint64_t *nmask, uint64_t n, const int8_t *pass_id, const int32_t *ref_id,
                            const uint32_t *quant, uint32_t n_samples, int32_t canon_pass, int32_t isomir_pass,
                            const int32_t *group_of_entry, uint64_t n_entries, const char *const *group_names,
                            uint32_t n_groups, const double *filtered, uint64_t *rows);

/* Packing helper used by hosts without numpy: ASCII reads -> SoA words. */
int mrg_pack_reads(const char *const *seqs, uint64_t n, uint32_t words_per_read,
                   uint64_t *reads, uint8_t *lens, uint64_t *nmask, int *has_n);
/* ... and -> the ragged form of mrg_cascade_run_long.  Two calls: with words == NULL only word_off[n + 1] (and lens,
 * if given) are filled -- word_off[n] is the number of words to allocate --, then with the buffers. */
int mrg_pack_reads_ragged(const char *const *seqs, uint64_t n, uint64_t *words, uint64_t *nmask, uint64_t *word_off,
                          uint32_t *lens, int *has_n);

#ifdef __cplusplus
}
#endif
#endif /* MIRGE_AMD_H */
