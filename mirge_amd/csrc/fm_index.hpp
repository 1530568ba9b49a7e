// Host-side FM index of one reference library (internal header).
//
// Role in the reference: the prebuilt bowtie 1 `.ebwt` files under
// miRge.Libs/<species>/index.Libs (MAIN:262-281) that every cascade pass loads
// (RAP:643, RAP:689).  This is our own format, laid out for the gfx950 match
// kernel, not a reader of bowtie's.
//
// Layout (all little-endian, what is uploaded to HBM verbatim):
//   blocks   16 B per 32 BWT symbols, so ONE 16-byte load answers a rank query:
//            uint16 cnt[4] (symbols before the block, relative to its 65536-symbol
//            superblock, sentinel excluded), uint32 lo (bit0 plane), uint32 hi
//            (bit1 plane).  Same 0.5 B/bp as the canonical 32 B / 64 bp block.
//   super    uint32[4] per 65536 BWT symbols: C[c] + symbols c before the superblock
//   text     2 bits/base, 16 bases per uint32, base p in bits [2(p&15), +1]
//   sa       full suffix array of text+'$' (HBM is 288 GB, so locate is one load,
//            not a sampled-SA walk), 8 B per row: bits 0-31 text position,
//            32-39 bases back to the start of its N-free segment (clamped 255),
//            40-47 bases to the end of the segment (clamped 255), 48-63 segment
//            id (0xFFFF when the library has more segments than that)
//   ftab     jump tables: T[c] = first BWT row whose suffix starts with k-mer c or a later one
//            (k-mers numbered lexicographically, first base most significant), so the interval
//            a backward search of the k-mer ends in is [T[c], T[c+1]) and the first k steps of
//            a seed search are one 8-byte load; 4^k + 1 words per table.  Up to four tables,
//            largest k first (ftab_ks): an optional bigger one
//            (k = 12..14 = ceil(log4 n) for whole-read seeds on large libraries, main + 1 on
//            small ones), the main one
//            (k = 8..11), k = 6 and k = 4 for short seed pieces
//   seg_*    N-free segments of the entries; an alignment must sit in one
//   chunk_seg[p>>5] = segment holding text position (p & ~31)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mrg {

// Stage timing of the library-load path, to stderr, when MIRGE_AMD_TIMING is set (scripts/lib_load_timing.py).
struct StageTimer {
  const char* what;
  bool on;
  double t0, t_last;
  explicit StageTimer(const char* w);
  void lap(const char* stage);
  ~StageTimer();
};

struct OccBlock {
  uint16_t cnt[4];
  uint32_t lo;
  uint32_t hi;
};
static_assert(sizeof(OccBlock) == 16, "occ block must be 16 bytes");
constexpr uint32_t kSuperShift = 16;  // superblock = 65536 BWT symbols

struct FmIndex {
  std::vector<std::string> names;
  std::vector<uint32_t> ref_len;                 // full entry length incl. N
  std::vector<std::vector<uint32_t>> ref_n_runs; // per entry: (start,len) pairs of N runs
  uint32_t n = 0;                                // text bases
  uint32_t primary = 0;
  uint32_t C[4] = {0, 0, 0, 0};
  std::vector<OccBlock> blocks;
  std::vector<uint32_t> super;  // 4 per superblock
  std::vector<uint32_t> text;
  std::vector<uint64_t> sa;
  // Libraries of >= 2^20 bases (their text cannot be staged in LDS) also get one 32-bit word of
  // text context per suffix-array row: bits 0-15 the 8 bases left of the row's position
  // (text[p-1] in the top two), bits 16-31 the 8 bases text[p+8 .. p+16) (text[p+8] in the low
  // two).  A short seed piece matches hundreds of rows of such a library; with the context a
  // candidate whose visible bases already show more mismatches than the pass allows is dropped
  // without touching its suffix-array row or the text.  Derived data, rebuilt on load.
  std::vector<uint32_t> ctx;
  // Libraries of at most kKmerBitsMaxBases bases: presence bitmap of their 9-mers (4^9 bits = 32 KB; bit c =
  // some text position starts the 9-mer with code c, first base in the low two bits).  Staged in
  // LDS, it answers "this seed piece cannot occur" for most pieces of reads that do not come
  // from the library -- without the jump-table load, a random L2 request.  Derived data.
  std::vector<uint32_t> kbits;
  uint8_t ftab_ks[4] = {0, 0, 0, 0};  // k of each table, descending; 0 = table absent
  std::vector<uint32_t> ftab;         // 4^k + 1 row boundaries per table, the tables of ftab_ks back to back
  // false: ftab / ctx are not built yet (a library of >= kLazyDeriveBases bases: a context with `device_tables` fills
  // them on the GPU from sa + text and never needs them here; whoever does calls derive_tables first)
  bool derived = true;
  std::vector<uint32_t> seg_start, seg_ref, seg_off, chunk_seg;
};

// Throws std::runtime_error on malformed input.
void build_index(const std::vector<std::string>& names,
                 const std::vector<std::string>& seqs, FmIndex& out);
void read_fasta(const std::string& path, std::vector<std::string>& names,
                std::vector<std::string>& seqs);
void plan_jump_tables(FmIndex& ix);   // ftab_ks only
void derive_tables(FmIndex& ix);      // jump tables, row context, 9-mer bitmap, if not built yet (not thread safe)
constexpr uint32_t kLazyDeriveBases = 1u << 20;
void build_jump_tables(FmIndex& ix);  // from sa + text
void build_row_context(FmIndex& ix);  // from sa + text
void build_kmer_bits(FmIndex& ix);    // from text
// Wide suffix-array rows of a large library (>= kWideRowMinBases; derived, built at upload time
// and never kept on the host): 16 bytes per row = the 8-byte row of `sa`, the 16 text bases left
// of its position (text[p-1] in the top two bits) and the 16 bases text[p+8 .. p+24).  One
// 16-byte load then locates a candidate AND shows 32 of the bases around the seed: for reads of
// up to 24 nt that is every base the seed did not cover, so a false candidate (2.6 per 11-base
// piece in an 11 Mbp library) is dropped without its text-window request.  Rows [row_lo, row_hi)
// into out[4 * (row_hi - row_lo)].
void fill_wide_rows(const FmIndex& ix, size_t row_lo, size_t row_hi, uint32_t* out);
// Seed buckets of a large library (seed_kernel, kernels.hpp): for every k-mer, k = kSeedBucketK = 11 --
// a seed of a 22..23-nt read under a one-mismatch policy -- the wide rows of the text positions that
// start with it, side by side in ONE 128-byte line addressed by the k-mer itself (first base in the
// low two bits): a seed lookup is one memory trip instead of a jump-table load and then its rows.
// Bucket = kSeedBucketRows rows of 16 bytes like fill_wide_rows', except for word 1: bits 0-5
// bases back to the segment start (clamped to 63), 6-11 bases to its end (clamped to 63), 12-15 in
// row 0 only the number of rows of the k-mer (kSeedBucketOverflow = more than a bucket holds: the
// jump table serves that k-mer), 16-31 segment id; unused rows have position 0xFFFFFFFF.
// seed_bucket_k: the k a library gets buckets for (0 = none: only libraries where a k-mer has
// 0.25..4 rows on average).  Buckets [code_lo, code_hi) into out[32 * (code_hi - code_lo)].
// Round 6: an OVERFLOWING k-mer (poly-A, a tandem repeat, an interspersed element: 10^3..10^5 rows) also gets its
// rows as a POSITION LIST -- the wide rows of its text positions in ascending position order, all lists back to back
// in one array (seed_pos_lists); words 2 and 3 of its bucket's row 0 hold the list's first row and its length.  A
// seed launch that must verify such an interval walks it in text order and stops at the first valid alignment
// (the lowest position IS the answer within a stratum) instead of verifying every row of the suffix interval.
constexpr uint32_t kSeedBucketK = 11, kSeedBucketRows = 8, kSeedBucketOverflow = 15;
uint32_t seed_bucket_k(const FmIndex& ix);
// over_start: null, or seed_pos_lists' start_by_lex (the headers of the overflowing buckets are filled from it)
void fill_seed_buckets(const FmIndex& ix, uint32_t k, uint64_t code_lo, uint64_t code_hi, uint32_t* out, const uint32_t* over_start = nullptr);
// Position lists of the k-mers with more than kSeedBucketRows rows: start_by_lex[c] = rows of overflowing k-mers
// in front of k-mer c (jump-table numbering, 4^k entries), positions = the text positions of every list, ascending
// inside a list, lists in k-mer order.  (Host restatement of libtables.hip: build_seed_pos_lists_device.)
void seed_pos_lists(const FmIndex& ix, uint32_t k, std::vector<uint32_t>& start_by_lex, std::vector<uint32_t>& positions);
// the wide row (fill_wide_rows' format) of one 8-byte suffix-array row
void wide_row_of_row(const FmIndex& ix, uint64_t row, uint32_t* out4);
// Pair tables of a small library, for policies with two seed mismatches (kernels.hip,
// stratum_kernel).  Four disjoint anchors of `anchor` bases at read offsets 0, A, 2A, 3A: two
// mismatches touch at most two of them, so every alignment with <= 2 seed mismatches matches
// exactly on (at least) one of the six anchor PAIRS.  A pair (i, j) is a gapped 2A-base key with
// gap d = (j - i) A between the starts of its halves; table t (d = A, 2A, 3A) lists, for every
// key, the suffix-array rows (same 8-byte format as `sa`) of the text positions p whose bases
// [p, p + A) and [p + d, p + d + A) spell it (first base in the low two bits, first half in the
// low 2A bits) and lie inside p's N-free segment.  jump: 4^(2A) + 1 boundaries per table,
// tables back to back; rows: the three lists back to back, row_off[t] = first row of list t.
// Against a 2 x 6-base pigeonhole piece (84 K bases: ~20 rows per piece, 3 pieces) an 8-base pair
// key leaves ~1.3 rows per lookup, 6 lookups.  Derived data, built at upload time.
struct PairTables {
  uint32_t anchor = 0;
  std::vector<uint32_t> jump;
  std::vector<uint64_t> rows;
  uint32_t row_off[4] = {0, 0, 0, 0};
};
void build_pair_tables(const FmIndex& ix, uint32_t anchor, PairTables& out);
constexpr uint32_t kPairMaxBases = 1u << 22;  // libraries up to this size get pair tables (3 x 8 B per base)
constexpr uint32_t kPairAnchor = 4;
constexpr uint32_t kWideRowMinBases = 1u << 20;
constexpr uint32_t kWideRowRightSkip = 8;
// largest library that gets the bitmap: its packed text (n / 4 bytes) plus the 32 KB bitmap must
// leave room for two match workgroups per CU (80 KB each incl. < 1 KB of control data)
constexpr uint32_t kKmerBitsMaxBases = 190000;
constexpr uint32_t kIndexKmerBitsK = 9;  // = mrg::kKmerBitsK in kernels.hpp
constexpr uint32_t kIndexKmerBitsWords = (1u << (2 * kIndexKmerBitsK)) / 32u;
void save_index(const FmIndex& ix, const std::string& path);
void load_index(const std::string& path, FmIndex& ix);
std::string entry_sequence(const FmIndex& ix, uint32_t i);

// bowtie 1 `.1.ebwt` -> entry names + sequences (what bowtie-inspect prints); ebwt.cpp
void read_ebwt(const std::string& prefix, std::vector<std::string>& names, std::vector<std::string>& seqs);

// Suffix array of s[0..n) over alphabet [0,K); s[n-1] must be the unique
// smallest symbol.  Induced sorting (SA-IS).
void suffix_array(const int32_t* s, int32_t* sa, int32_t n, int32_t K);

}  // namespace mrg
