// Kernel parameter blocks and launch entry points shared by kernels.hip and
// capi.hip (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrg {

// Match workgroups are 1024 threads: 16 waves share one staged copy of the
// library (or, for HBM-served libraries, of its superblock table) and one
// survivor ring; two workgroups per CU give the full 32 waves.
template <bool LDSI>
struct MatchBlock {
  static constexpr uint32_t kThreads = 1024u;
};
constexpr uint32_t kTallyThreads = 1024u;
constexpr uint32_t kTallyCatReplicas = 32u;  // LDS copies of each category bin (one per lane & 31)
// 9-mer presence bitmap of a small library (FmIndex::kbits): 4^9 bits
constexpr uint32_t kKmerBitsK = 9u;
constexpr uint32_t kKmerBitsWords = (1u << (2u * kKmerBitsK)) / 32u;
// Survivor lists are segmented: workgroup b of the producing pass owns segment b.
constexpr uint32_t kMaxSegments = 2048u;
// entries a survivor list can hold beyond one per read: every launch rounds its workgroups' segments up to its chunk size
constexpr uint64_t kListSlack = 1ull << 23;
constexpr uint32_t kMatchCtlBytes = (4u + 4u * 16u + 10u) * 4u;  // control words, 16 B per wave, 5 counters

// Packed assignment (include/mirge_amd.h: mrg_pack_assignments / mrg_cascade_run_packed): one word per read.
// Kernels that are handed a `packed` array write it INSTEAD of pass_id / ref_id / pos / mm.
__host__ __device__ inline uint32_t pack_assignment(int32_t pass, uint32_t ref, uint32_t pos, uint32_t mm) {
  return ((uint32_t)(pass + 1) << 28) | ((mm < 3u ? mm : 3u) << 26) | ((ref < 0x3FFFFu ? ref : 0x3FFFFu) << 8) | (pos < 0xFFu ? pos : 0xFFu);
}

// The jump tables of one library in ascending k (k[0] = 0: tables not used; a missing big table
// repeats the main one), with the word offset of each inside `ftab`.
struct JumpTables {
  uint32_t k[4];
  uint32_t off[4];
};

struct MatchParams {
  // library (device pointers)
  const uint32_t* blocks;  // 16 B per 32 BWT symbols
  const uint32_t* super;   // 16 B per 65536 BWT symbols
  const uint32_t* text;
  const uint64_t* sa;      // 8 B rows: pos | before<<32 | after<<40 | seg<<48
  const uint32_t* ctx;     // per-row text context of a large library (null otherwise)
  const uint32_t* kbits;   // 9-mer presence bitmap of a small library, staged in LDS (null = not used)
  const uint32_t* ftab;    // k-mer jump tables: lo, hi per k-mer
  JumpTables tabs;
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  uint32_t n, nblk, nsup, primary, text_words;
  uint32_t simple_segs;  // every entry is one N-free segment: seg == entry, offset 0
  // reads
  const uint64_t* reads;
  const uint8_t* lens;
  const uint64_t* nmask;  // may be null
  uint32_t n_total;          // SoA stride and identity list length
  const uint32_t* idx_in;    // null = identity list of n_total reads
  const uint32_t* in_count;  // entries in each of the producer's segments
  uint32_t in_nseg, in_seg_cap;
  uint32_t* idx_out;         // null on the last pass
  uint32_t* out_count;       // [gridDim.x]
  uint32_t out_seg_cap;
  // pair_wave_kernel (dict.hip) only: words per entry of the input list (1 = indices, 4 = index, length, read:
  // SeedParams); it writes 16-byte entries; out_init: the cascade's first launch wrote every output already
  uint32_t in_stride, out_init;
  // pair_wave_kernel, round 6: the second word of every read when the batch's reads of 33..63 nt ride the dictionary kernels
  // (a read of 33..35 nt is 32 bases or fewer behind `-5 1 -3 2`), null = none
  const uint64_t* reads_hi;
  // outputs
  int8_t* pass_id;
  int32_t* ref_id;
  int32_t* pos;
  uint8_t* mm;
  uint32_t* packed;    // non-null: the one output array (the four above are not touched)
  uint64_t* counters;  // processed, aligned, steps, candidates, lookups of this pass
  // policy
  int32_t seed_len, max_mm_seed, max_mm_total, trim5, trim3, min_len, max_len, poly_t;
  int32_t pass_index;
  uint32_t wstop;
  uint32_t wide_rows;  // seed intervals wider than this are verified by the whole wave
  // STRATA launches only: the strata [k_first, k_last] (piece counts) this launch searches.  A
  // 2-mismatch pass is split into two launches: strata 1..2 over the incoming reads (a best hit
  // with fewer than 2 mismatches is final, everything else survives) and stratum 3 over the
  // COMPACTED survivors -- the three-piece search finds every alignment with <= 2 mismatches on
  // its own, so nothing has to be carried over, and its ~45 candidates per read are verified with
  // every lane holding a read instead of the 45 % that reach that stratum.
  int32_t k_first, k_last;
  uint32_t count_processed;  // 0: the launch continues a pass whose reads were counted already
  uint32_t uniform_len;      // stratum_kernel: != 0 = every read has this length (lens is not read)
  // stratum_kernel: pair tables of the library (fm_index.hpp: PairTables); pair_anchor = 0: none.
  // Reads whose seed region holds the four anchors (>= 4 * pair_anchor bases) are searched through
  // the six anchor pairs, shorter ones through the pigeonhole pieces of strata [k_first, k_last].
  const uint32_t* pair_jump;
  const uint64_t* pair_rows;
  uint32_t pair_row_off[3];
  uint32_t pair_anchor;
  // a second set with anchors one base shorter, for the reads between 4 (pair_anchor - 1) and
  // 4 pair_anchor seed bases (16..18-nt reads under -5 1 -3 2: their three 4..5-base pigeonhole
  // pieces match ~350 rows each in 88 kbp); null = none
  const uint32_t* pair_jump_s;
  uint32_t pair_row_off_s[3];  // into pair_rows as well (the second set's lists follow the first's)
};

// ---------------------------------------------------------------------------
// Fused launch: several consecutive cascade passes (RAP:636-705 iterations) over ONE walk of
// the survivor list.  A read that a sub-pass does not claim is offered to the next one while it
// is still in registers, so the list round trip and the lens[r] / reads[r] gathers -- what the
// small-library passes spend most of their time on -- happen once per group instead of once
// per pass.  Libraries are served from L2/HBM (their texts are only touched by the few
// candidates that survive the filters); LDS holds what answers most reads without any memory
// request: one folded 9-mer presence bitmap per small library.
// ---------------------------------------------------------------------------
constexpr uint32_t kMaxFused = 8u;
constexpr uint32_t kFusedCntReplicas = 16u;  // LDS slots per 64-bit counter (lane & 15)

struct SubPass {
  const uint32_t* blocks;
  const uint32_t* super;  // read from L2 in the rare LF step (not staged)
  const uint32_t* text;
  const uint64_t* sa;
  const uint32_t* ctx;    // row context of a large library (null otherwise)
  const uint4* sa16;      // wide rows of a large library (null otherwise): row, 16 bases left, 16 bases from +8
  // pair tables of a large library (pairs.hip), pair_anchor = A = 0: none.  A one-mismatch sub-pass
  // alone in its round searches the reads whose seed region is 3A .. 4A - 1 bases (two pigeonhole
  // pieces of less than 2A bases) through the three anchor pairs (0,1) (1,2) (0,2) instead;
  // fused_kernel<W, true> only.
  const uint32_t* pair_jump;
  const uint64_t* pair_rows;
  uint32_t pair_row_off[2];
  uint32_t pair_anchor;
  const uint32_t* ftab;
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  const uint64_t* nmask;  // = FusedParams::nmask (so that verify_row reads one parameter block)
  uint64_t* counters;     // processed, aligned, steps, candidates, lookups of this pass
  JumpTables tabs;
  uint32_t n, primary, simple_segs;
  uint32_t text_lds_off;    // LDS word offset of the staged packed text (when text_lds_words != 0)
  uint32_t text_lds_words;  // 0 = the text is read from L2/HBM
  uint32_t kb_bit;   // this sub-pass's bit in the entries of its round's 9-mer table; 0xFF = no filter
  int32_t seed_len, max_mm_seed, max_mm_total, trim5, trim3, min_len, max_len, poly_t;
  int32_t pass_index;
};

struct FusedParams {
  SubPass sub[kMaxFused];
  uint32_t n_sub;
  uint32_t kb_words;   // LDS words of all staged bitmaps
  uint32_t txt_words;  // LDS words of all staged texts (right after the bitmaps)
  const uint64_t* reads;
  const uint8_t* lens;
  const uint64_t* nmask;
  uint32_t n_total;
  uint32_t uniform_len;      // != 0: every read has this length (lens is not read)
  const uint32_t* idx_in;
  const uint32_t* in_count;
  uint32_t in_nseg, in_seg_cap;
  uint32_t* idx_out;         // null: the group ends the cascade
  uint32_t* out_count;
  uint32_t out_seg_cap;
  int8_t* pass_id;
  int32_t* ref_id;
  int32_t* pos;
  uint8_t* mm;
  uint32_t* packed;  // non-null: the one output array
  uint32_t wstop;
  // rounds: runs of sub-passes whose seed pieces are looked up together (speculatively across the
  // sub-passes of the run; the owner replays the results in cascade order)
  uint32_t n_rounds;
  uint8_t round_first[kMaxFused], round_count[kMaxFused];
  // 9-mer table of a round of bitmap-filtered sub-passes: entry c (code & (2^log2 - 1)) holds one bit
  // per sub-pass of the round ("this library has a 9-mer with that code"), `bits` (4 or 8) bits per
  // entry, so ONE LDS read answers a 9-mer for every library of the round.  log2 == 0: no table.
  const uint32_t* round_kb_src[kMaxFused];
  uint32_t round_kb_off[kMaxFused];   // LDS word offset
  uint8_t round_kb_log2[kMaxFused], round_kb_bits[kMaxFused];
};

// LDS bytes of a fused launch besides the bitmaps: sub-pass table + one 16-byte result slot per
// thread + counters + control words + wave slots
constexpr uint32_t fused_fixed_lds_bytes() {
  return kMaxFused * 40u * 4u + 1024u * 16u + kMaxFused * 3u * kFusedCntReplicas * 8u + 16u + 16u * 128u;  // + owner maps
}
// a sub-pass needs two bits of the per-lane item mask: at most this many sub-passes per round
constexpr uint32_t kMaxRoundSubs = 8u;

hipError_t launch_fused(const FusedParams& p, uint32_t words_per_read, uint32_t grid, uint32_t lds_bytes,
                        hipStream_t stream);  // (the anchor-pair instantiation when a sub-pass has pair tables)
// pairs.hip: pair tables of a large library from its device-resident suffix array and text
hipError_t build_pair_tables_device(const uint64_t* sa, const uint32_t* text, uint32_t n_rows, uint32_t anchor,
                                    uint32_t n_gaps, uint32_t* jump, uint64_t* rows, uint32_t* row_off,
                                    hipStream_t stream);

// ---------------------------------------------------------------------------
// dict.hip: the dictionary kernels for batches of one-word reads without N.  Same survivor-list
// format, outputs and counter slots as match_kernel / fused_kernel, so every launch of the plan can
// be either.
// ---------------------------------------------------------------------------
// exact_dict_kernel: a pass with no seed mismatch on a library that has an exact-match dictionary
// (dict_index.hpp): one 16-byte slot load per read; reads shorter than the dictionary's key, and
// keys whose chain overflowed, take the FM index (prefix interval of the largest jump table that
// fits the read, every row verified against the text).
struct ExactParams {
  const uint4* slots;
  uint32_t log2_slots, key_bases;
  const uint32_t* kbits;  // the library's 9-mer presence bitmap (null = none): staged in LDS as a reject filter
  // FM index of the same library, for the fallback
  const uint32_t* blocks;
  const uint32_t* super;
  uint32_t primary;
  const uint32_t* ftab;
  JumpTables tabs;
  const uint64_t* sa;
  const uint32_t* text;
  uint32_t n;
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  uint32_t simple_segs;
  // reads (one word each, no N mask)
  const uint64_t* reads;
  const uint8_t* lens;
  uint32_t n_total;
  const uint32_t* idx_in;    // null = identity list of n_total reads
  const uint32_t* in_count;
  uint32_t in_nseg, in_seg_cap;
  uint32_t* idx_out;         // null on the last pass
  uint32_t* out_count;
  uint32_t out_seg_cap;
  int8_t* pass_id;
  int32_t* ref_id;
  int32_t* pos;
  uint8_t* mm;
  uint32_t* packed;    // non-null: the one output array
  uint64_t* counters;  // processed, aligned, steps (0), candidates (slots / rows compared), lookups (slot + table loads)
  int32_t seed_len, max_mm_total, trim5, trim3, min_len, max_len, poly_t, pass_index;
};
constexpr uint32_t kExactChunk = 4096u;  // reads a workgroup takes per trip (four per lane)
hipError_t launch_exact_dict(const ExactParams& p, uint32_t grid, hipStream_t stream);
bool exact_dict_streams(const ExactParams& p);  // the launch takes the streaming instantiation: it writes every output of the batch
bool exact_dict_stretches(const ExactParams& p, uint32_t grid);  // ... its <.., .., true> form: one contiguous stretch per workgroup (mrg_pass_stats.variant bit 4)

// seed_kernel: one launch = a run of consecutive passes with at most one seed mismatch, as UNITS:
//   kind 0  V + 1 disjoint seeds of k = floor(R / (V + 1)) bases at the front of the seed region
//           (R = min(read, shortest -l of the members)): at most V mismatches there leave one seed
//           clean.  A seed's first k' bases (k' = the largest jump table <= k) name an interval of
//           suffix-array rows = text positions; every row is a candidate start, verified from its
//           16-byte wide row (position, segment room and 32 bases of context) or, when the context
//           does not cover the read, from the text.  Several libraries searched with the same policy
//           (tRNA `-v 1`, snoRNA and rRNA `-n 1`: RAP:579,581,582) are ONE unit over the index of
//           their concatenation: members[] tells which pass an entry belongs to.  Small units filter
//           seeds through presence bitmaps of the library's 8..11-mers (L2-resident) first.
//   kind 1  a pass without seed mismatches on a library with an exact-match dictionary.
// A workgroup (four waves; six of them per CU hide each other's memory trips) works through tiles
// of 256 reads in three phases with dense lanes in each: per
// read, which (unit, seed) ITEMS need the index -> per item, the jump-table load and its ROWS ->
// per row, verification and a 64-bit atomic min into the read's LDS slot -> per read, the claim in
// cascade order (the key starts with the pass index), outputs, survivors and counters.
constexpr uint32_t kSeedMaxUnits = 3u, kSeedMaxMembers = 4u, kSeedThreads = 256u;
constexpr uint32_t kSeedRowsPerItem = 24u;  // wider intervals are verified by the whole workgroup
constexpr uint32_t kSeedWideCap = 64u;
// Word offset of the k-mer presence bitmap (k = 8..11, 4^k bits each, back to back) inside a seed
// library's `kbits` array: arithmetic, so that a lane with its own k needs no table load.
__host__ __device__ constexpr uint32_t seed_kbits_word_off(uint32_t k) {
  return k <= 8u ? 0u : k == 9u ? 2048u : k == 10u ? 2048u + 8192u : 2048u + 8192u + 32768u;
}
constexpr uint32_t kSeedKbitsWords = 2048u + 8192u + 32768u + 131072u;

struct SeedMember {
  int32_t pass_index, seed_len, max_mm_total;
  uint32_t entry_lo;  // first entry of this member in the unit's (union) library
};
struct SeedUnit {
  uint32_t kind;
  const uint32_t* ftab;
  JumpTables tabs;
  const uint4* sa16;
  const uint4* buckets;  // seed buckets of the library (fm_index.hpp), null = none: a seed of exactly bucket_k
  uint32_t bucket_k;     // bases finds its rows in ONE 128-byte line instead of jump table + rows
  const uint4* pos_rows;  // position lists of the k-mers whose bucket overflows (fm_index.hpp: seed_pos_lists), null = none
  const uint64_t* sa;
  const uint32_t* text;
  const uint32_t* blocks;  // occ blocks + superblocks + sentinel row: the FM fallback of a dictionary unit (kind 1)
  const uint32_t* super;
  uint32_t primary;
  uint32_t n;
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  uint32_t simple_segs;
  // wave_seed_kernel, one large library searched with ONE seed mismatch: pair tables of three anchors of bpair_anchor
  // bases (pairs.hip; gaps A and 2 A) for the reads whose seed region is 3 A .. 4 A - 1 bases (16..19-nt reads:
  // their two seeds of 8..9 bases would name 40..170 rows each in 11 Mbp); 0 = none
  const uint32_t* bpair_jump;
  const uint64_t* bpair_rows;
  uint32_t bpair_row_off[2];
  uint32_t bpair_anchor;
  const uint4* slots;  // kind 1
  uint32_t log2_slots, key_bases;
  const uint32_t* kbits;  // presence bitmaps of the k-mers, k = 8..11, at word offsets seed_kbits_word_off(k); null = none
  int32_t max_mm_seed, trim5, trim3, min_len, max_len, poly_t;
  int32_t min_seed_len, max_total;  // over the members
  uint32_t n_members;
  SeedMember m[kSeedMaxMembers];
};
struct SeedParams {
  SeedUnit unit[kSeedMaxUnits];
  uint32_t n_units;
  uint32_t impl;            // 0 = seed_kernel (tiles of 256 reads, three barriers per tile), 1 = wave_seed_kernel (every wave on its own)
  uint32_t wave_regs;       // wave_seed_kernel: 1 = the instantiation with more registers, fewer resident workgroups
  uint32_t reads_per_lane;  // seed_kernel: 1
  uint32_t item_cap;        // seed_kernel, per unit: 256 x seeds
  uint32_t row_cap;         // rows a tile (seed_kernel) / a wave (wave_seed_kernel) can queue
  uint64_t* stats;          // counter slots [pass][5]: processed, aligned, steps, candidates, lookups
  const uint64_t* reads;
  const uint8_t* lens;
  uint32_t n_total;
  // the input list: entries of in_stride words, word 0 = the read's index (1: an index list as exact_dict_kernel and
  // the FM kernels write them; 4: a list that CARRIES its reads -- index, length, the packed read: 16 bytes).  The
  // seed launches WRITE the second form: the launch behind them streams its reads instead of gathering them through
  // the indices (pair_wave_kernel walks 22 % of the batch: 4.5 lines touched per line's worth of reads)
  const uint32_t* idx_in;
  uint32_t in_stride;
  const uint32_t* in_count;
  uint32_t in_nseg, in_seg_cap;
  uint32_t* idx_out;  // null: the launch ends the cascade; else 16-byte entries
  uint32_t* out_count;
  uint32_t out_seg_cap;
  uint32_t out_init;  // the first launch of the cascade wrote every output ("unannotated" included): the last one need not
  int8_t* pass_id;
  int32_t* ref_id;
  int32_t* pos;
  uint8_t* mm;
  uint32_t* packed;  // non-null: the one output array
  // wave_seed_kernel, units with position lists: walk_cap records of 32 bytes per wave (grid x 4 waves), null = none
  uint4* walk_buf;
  uint32_t walk_cap;
  uint32_t walk_diag;  // experiments: the number of such reads goes into the `steps` counter of the launch's first pass
  // round 6: the second word of every read (a two-word batch whose reads of 33..63 nt ride the seed kernels), null = none.
  // (Behind everything else: the walk reads list pointers and counts from the kernel-argument segment every trip, and a
  // field in front of them moved their offsets -- the headline's large launch lost 4 % to that.)
  const uint64_t* reads_hi;
};
uint32_t seed_lds_bytes(const SeedParams& p);
// workgroups per CU the instantiation a launch gets can keep resident (registers; LDS permitting)
uint32_t seed_wgs_per_cu(const SeedParams& p);
hipError_t launch_seed(const SeedParams& p, uint32_t grid, hipStream_t stream);

// ingest.hip: FASTQ text (whole four-line records) on the device -> packed reads.  h_info[7]: records, kept
// (packed), kept but longer than 32 W bases, longest packed read, has N, status (0 ok), first bad record.
// ads: the adapter sequences of `-ad <sequence>[,<sequence>...]` (upper case; n = 0: none), searched after the quality trim
// and the cutter as cutadapt does (error rate 0.12, k = floor(0.12 x length) errors in the band, minimum overlap 3)
constexpr uint32_t kAdapterMaxLen = 64u, kAdapterMax = 4u;
struct AdapterSet {
  uint32_t n;
  uint8_t len[kAdapterMax], k[kAdapterMax];
  char seq[kAdapterMax][kAdapterMaxLen];
};
hipError_t fastq_parse_device(const char* d_text, uint64_t n_bytes, int32_t phred, int32_t cutoff, int32_t min_len, int32_t cut,
                              const AdapterSet& ads, uint32_t W, uint64_t cap, uint64_t* d_words, uint8_t* d_lens, uint64_t* d_nmask,
                              uint64_t* h_info, hipStream_t stream);
// ingest.hip: the compact wire form of a host-resident collapsed read set (mrg_expand_compact) -> the
// arrays of the cascade and the tally.  Run r: reads [end[r-1], end[r]) have len[r] bases, 2 len[r] bits each
// from word base[r] of the bit stream.
constexpr uint32_t kCompactMaxRuns = 64;
struct CompactRuns {
  uint32_t n;
  uint32_t end[kCompactMaxRuns];
  uint32_t base[kCompactMaxRuns];
  uint8_t len[kCompactMaxRuns];
};
hipError_t expand_compact(const uint64_t* d_bits, const CompactRuns& runs, const uint8_t* d_quant8, const uint32_t* d_esc, uint64_t n_esc,
                          uint64_t n, uint32_t n_samples, uint64_t* d_reads, uint8_t* d_lens, uint32_t* d_quant, hipStream_t stream);
// libtables.hip: a large library's derived tables filled on the device from its rows and packed text (each restates the
// fm_index.cpp function named beside it).  ks: the jump tables' k in storage order (0 = absent); tmp of
// jump_tables_device_temp_bytes(n + 1) bytes.  Asynchronous on `stream`.
size_t jump_tables_device_temp_bytes(uint32_t n_rows);
hipError_t build_jump_tables_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, const uint8_t ks[4], uint32_t* ftab,
                                    void* tmp, hipStream_t stream);                                                              // build_jump_tables
hipError_t build_row_context_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, uint32_t* ctx, hipStream_t stream);  // build_row_context
hipError_t build_wide_rows_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, uint32_t* sa16, hipStream_t stream);   // fill_wide_rows
hipError_t build_seed_buckets_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, const uint32_t* tab, uint32_t k,
                                     uint32_t* buckets, hipStream_t stream);                                                     // fill_seed_buckets
// position lists of the k-mers whose bucket overflows + the headers in their buckets (fm_index.hpp: seed_pos_lists);
// *out_rows is hipMalloc'ed by the call (null: no k-mer overflows); synchronises the stream
hipError_t build_seed_pos_lists_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, const uint32_t* tab, uint32_t k,
                                       uint32_t* buckets, uint32_t** out_rows, uint64_t* out_n, hipStream_t stream);

// dictbuild.hip: the exact-match dictionary of a large library filled on the device (same slot format and rules as
// dict_index.cpp).  slots: 2^log2_slots x 16 bytes, zeroed by the call; tmp: exact_dict_device_temp_bytes(n) bytes;
// counts[0] = positions stored, counts[1] = home slots whose chain overflowed.  Synchronises the stream.
bool exact_dict_device_ok(uint32_t n, uint32_t key_bases, uint32_t log2_slots);
size_t exact_dict_device_temp_bytes(uint32_t n);
hipError_t build_exact_dict_device(const uint32_t* text, uint32_t n, const uint32_t* seg_start, const uint32_t* seg_ref, const uint32_t* seg_off,
                                   const uint32_t* chunk_seg, uint32_t key_bases, uint32_t log2_slots, void* slots, void* tmp, uint64_t counts[2],
                                   hipStream_t stream);

hipError_t launch_pack_assignments(const int8_t* pass_id, const int32_t* ref_id, const int32_t* pos, const uint8_t* mm, uint64_t n,
                                   uint32_t* packed, hipStream_t stream);

// long_reads.hip: the cascade for reads of ANY length (runAnnotationPipeline.py:543-554 writes every unannotated
// read into a pass's FASTA whatever its length; the packed batches of mrg_cascade_run stop at 255 nt, one length
// byte).  Such reads are rare (an untrimmed long-cycle run, a read-through): one WAVE per read, the read left in
// global memory in the ragged form (word_off[r] .. word_off[r + 1]), every pass of the cascade in turn over the
// library's own FM index -- pigeonhole pieces of the seed region by backward search, every row of a piece's
// interval located through the suffix array and verified against the text, 32 bases per step, by the wave's lanes.
struct LongPass {
  const uint32_t* blocks;
  const uint32_t* super;
  const uint32_t* text;
  const uint64_t* sa;
  const uint32_t* ftab;
  JumpTables tabs;
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  uint32_t n, primary, simple_segs;
  int32_t seed_len, max_mm_seed, max_mm_total, trim5, trim3, min_len, max_len, poly_t;
};
struct LongParams {
  const LongPass* pass;  // device array, n_pass entries
  uint32_t n_pass;
  const uint64_t* words;     // ragged: read r = words[word_off[r] .. word_off[r + 1]), 32 bases per word
  const uint64_t* nmask;     // same shape or null
  const uint64_t* word_off;  // n + 1
  const uint32_t* lens;
  uint32_t n;
  uint32_t wstop;
  int8_t* pass_id;
  int32_t* ref_id;
  int32_t* pos;
  uint8_t* mm;
  uint64_t* counters;     // [n_pass][5]: processed, aligned, steps, candidates, lookups (added to)
  uint64_t* pass_counts;  // null, or [n_pass][2]: processed, aligned (added to: the count vector a sharded run reduces)
};
hipError_t launch_long_reads(const LongParams& p, uint32_t grid, hipStream_t stream);

constexpr uint32_t kCountThreads = 256u;

struct CountParams {
  const uint32_t* blocks;
  const uint32_t* super;
  const uint32_t* text;
  const uint64_t* sa;
  const uint32_t* ctx;
  const uint32_t* ftab;
  JumpTables tabs;
  uint32_t n, nsup, primary;
  const uint64_t* reads;
  const uint8_t* lens;
  const uint64_t* nmask;
  uint64_t n_reads;
  int32_t seed_len, max_mm_seed, max_mm_total;
  uint32_t wstop, max_rows;
  uint8_t* best_mm;   // written by the count sweep, read by the list sweep
  uint8_t* count;     // saturating, or
  uint32_t* count32;  // exact (when non-null)
  // list sweep (out_ref != null)
  const uint32_t *seg_start, *seg_ref, *seg_off, *chunk_seg;
  const uint64_t* offsets;
  int32_t* out_ref;
  int32_t* out_pos;
  uint64_t out_cap;
  uint32_t only_todo;  // round 6: count_variants_kernel ran first; count_kernel takes the reads it marked (best_mm 254) only
};

struct TallyParams {
  const uint32_t* packed;  // non-null: pass and entry come from the packed words (pass_id / ref_id unused)
  const int8_t* pass_id;
  const int32_t* ref_id;
  const uint32_t* quant;
  uint64_t n;
  uint32_t n_samples, n_mirna, n_pass;
  int32_t canon_pass, isomir_pass;
  uint64_t* counts;
  uint32_t vec4;  // one sample and 16-byte aligned arrays: four reads per lane and trip
  // "fused_step": the per-pass processed / aligned of the cascade in front go out with this launch (its first
  // workgroup copies them) instead of with a launch of their own; null = nothing to export
  const uint64_t* export_stats;  // [n_pass][5]
  uint64_t* export_out;          // [n_pass][2]
  uint32_t export_n_pass;
};

// A-to-I position tally (writeDataToCSV.py:145-229 on the cascade's own alignments)
constexpr uint32_t kEditPositions = 32u;  // mature-miRNA positions tallied per entry
constexpr uint32_t kEditThreads = 1024u;
constexpr uint32_t kEditHashLog2 = 11u, kEditHashSlots = 1u << kEditHashLog2;  // LDS hash of position bins
struct EditParams {
  const uint64_t* reads;
  const uint8_t* lens;
  const uint64_t* nmask;  // may be null
  uint32_t words_per_read;
  const uint32_t* packed;  // non-null: pass, entry and offset come from the packed words
  const int8_t* pass_id;
  const int32_t* ref_id;
  const int32_t* pos;
  const uint32_t* quant;
  const uint8_t* keep;    // may be null: per-read 0/1 (the genome-uniqueness / RPM selection of the host)
  const uint32_t* remap;  // may be null: entry -> output bin (merged miRNA names)
  uint64_t n;
  uint32_t n_samples, n_bins;  // n_bins = output miRNA bins (entries when remap is null)
  int32_t canon_pass, isomir_pass, isomir_trim5;
  uint32_t flank5, flank3;     // library entry = flank5 + mature + flank3 (RAP:413: 2 and 6)
  uint32_t from_base, to_base; // the substitution tallied per position (A -> G: 0, 2)
  // library text
  const uint32_t* text;
  const uint32_t* seg_start;   // entry e = text [seg_start[e], seg_start[e + 1]) (every entry one N-free segment)
  uint32_t text_words, n_entries;
  uint64_t* counts;            // [n_bins][S][3] then [n_bins][kEditPositions][S]
  uint32_t vec4;               // one sample, one-word reads, 16-byte aligned arrays: four reads per lane and trip
};
// LDS bytes of the privatised totals (lds_hist) and of the staged library (lds_lib)
constexpr uint64_t edit_hist_lds_bytes(uint64_t n_bins, uint64_t S) { return n_bins * S * 16u + ((n_bins * S + 3u) & ~3ull) * 4u; }
constexpr uint64_t kEditHashLdsBytes = (uint64_t)kEditHashSlots * 12u;
hipError_t launch_edit_tally(const EditParams& p, bool lds_hist, bool lds_lib, uint32_t grid, uint32_t lds_bytes,
                             hipStream_t stream);

// lds_mode: 0 = index in HBM/L2, 1 = occ blocks in LDS, 2 = occ blocks + text in LDS,
//           3 = text only in LDS (occ blocks from L2)
hipError_t launch_match(const MatchParams& p, uint32_t words_per_read, int lds_mode,
                        uint32_t grid, uint32_t lds_bytes, hipStream_t stream);
// stratum_kernel: strata [k_first, k_last] of a stratum-first pass, rows compacted over the wave.
// LDS = superblocks + (lds_text ? packed text : 0) + (p.kbits ? bitmap : 0) + kStratumCtlBytes
constexpr uint32_t kStratumCtlBytes = 16u + 1024u * 8u + 5u * 8u;
hipError_t launch_stratum(const MatchParams& p, uint32_t words_per_read, bool lds_text, uint32_t grid,
                          uint32_t lds_bytes, hipStream_t stream);
// dict.hip: pair_wave_kernel, the anchor-pair search of a 2-mismatch pass for one-word reads without N, every wave
// on its own with items and rows compacted over the wave (workgroups of kSeedThreads; chunks of 1024 list entries)
uint32_t pair_wave_lds_total();
hipError_t launch_pair_wave(const MatchParams& p, uint32_t grid, hipStream_t stream);
hipError_t launch_tally(const TallyParams& p, bool lds_hist, uint32_t grid,
                        uint32_t lds_bytes, hipStream_t stream);
// collapse.hip: raw reads -> unique reads + per-sample counts + length histogram
hipError_t collapse_reads(const uint64_t* d_reads, uint32_t W, const uint8_t* d_lens,
                          const uint64_t* d_nmask, const uint16_t* d_sample, uint32_t n,
                          uint32_t n_samples, uint32_t max_len, uint64_t cap, uint64_t* d_u_words,
                          uint8_t* d_u_lens, uint64_t* d_u_nmask, uint32_t* d_quant,
                          uint64_t* d_len_hist, uint32_t* h_n_unique, hipStream_t stream, void* arena_base = nullptr,
                          uint64_t arena_bytes = 0, int n_cu = 0, bool allow_fast = true);  // arena: device scratch the temporaries are carved from
hipError_t launch_count_variants(const CountParams& p, uint32_t grid, hipStream_t stream);
hipError_t launch_count(const CountParams& p, uint32_t words_per_read, uint32_t grid, uint32_t lds_bytes,
                        hipStream_t stream);
hipError_t exclusive_sum_u32_u64(const uint32_t* in, uint64_t* out, uint64_t n_plus_1, hipStream_t stream);
hipError_t launch_export_pass_counts(const uint64_t* stats, uint32_t n_pass, uint64_t* out,
                                     hipStream_t stream);
// A batch split into two segmented lists: the reads the dictionary kernels take (min_len <= length
// <= 32, no N: they read word 0 only) and the rest.  Workgroup b owns segment b of both lists (seg_cap entries each, chunks of 1024
// reads).
struct SplitParams {
  const uint8_t* lens;
  const uint64_t* nmask;  // null = no read has an N
  uint32_t n_total;
  uint32_t min_len;
  uint64_t long_ok;         // round 6: bit b = reads of 33 + b nt go with the one-word reads too (the seed kernels' LONG instantiations)
  const uint64_t* nmask_hi;  // the second mask word of every read (null: none has an N)
  uint32_t seg_cap;
  uint32_t* idx_short;
  uint32_t* cnt_short;
  uint32_t* idx_rest;
  uint32_t* cnt_rest;
};
hipError_t launch_split(const SplitParams& p, uint32_t grid, hipStream_t stream);
// A segmented survivor list that carries its reads (16-byte entries: SeedParams) -> the index list the FM kernels
// and exact_dict_kernel read, segment for segment (non-default plans and the spike-in pass only)
hipError_t launch_list_thin(const uint4* fat_in, uint32_t* thin_out, const uint32_t* in_count, uint32_t* out_count, uint32_t n_seg,
                            uint32_t seg_cap, hipStream_t stream);

}  // namespace mrg
