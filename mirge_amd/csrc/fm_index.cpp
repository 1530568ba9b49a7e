// Host-side construction of the FM index (see fm_index.hpp for the layout and
// the reference role).  Pure C++17, no GPU.
#include "fm_index.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <stdexcept>
#include <thread>

namespace mrg {

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace
StageTimer::StageTimer(const char* w) : what(w), on(std::getenv("MIRGE_AMD_TIMING") != nullptr), t0(now_s()), t_last(t0) {}
void StageTimer::lap(const char* stage) {
  if (!on) return;
  const double t = now_s();
  std::fprintf(stderr, "[timing] %s: %s %.3f s\n", what, stage, t - t_last);
  t_last = t;
}
StageTimer::~StageTimer() {
  if (on) std::fprintf(stderr, "[timing] %s: total %.3f s\n", what, now_s() - t0);
}

// ---------------------------------------------------------------------------
// Suffix array by induced sorting.  `s[n-1]` is the unique smallest symbol.
// Works in place in `sa`; the reduced problem reuses the tail of `sa`.
// ---------------------------------------------------------------------------
namespace {

template <class Sym>
struct Sais {
  const Sym* s;
  int32_t* sa;
  int32_t n;
  int32_t K;
  std::vector<uint8_t> stype;  // 1 = S-type suffix
  std::vector<int32_t> bkt;

  bool lms(int32_t i) const { return i > 0 && stype[i] && !stype[i - 1]; }

  void bucket_ends() {
    std::fill(bkt.begin(), bkt.end(), 0);
    for (int32_t i = 0; i < n; ++i) ++bkt[(int32_t)s[i]];
    int32_t sum = 0;
    for (int32_t c = 0; c < K; ++c) {
      sum += bkt[c];
      bkt[c] = sum;
    }
  }
  void bucket_starts() {
    std::fill(bkt.begin(), bkt.end(), 0);
    for (int32_t i = 0; i < n; ++i) ++bkt[(int32_t)s[i]];
    int32_t sum = 0;
    for (int32_t c = 0; c < K; ++c) {
      int32_t cnt = bkt[c];
      bkt[c] = sum;
      sum += cnt;
    }
  }
  void induce() {
    bucket_starts();
    for (int32_t i = 0; i < n; ++i) {
      int32_t p = sa[i];
      if (p > 0 && !stype[p - 1]) sa[bkt[(int32_t)s[p - 1]]++] = p - 1;
    }
    bucket_ends();
    for (int32_t i = n - 1; i >= 0; --i) {
      int32_t p = sa[i];
      if (p > 0 && stype[p - 1]) sa[--bkt[(int32_t)s[p - 1]]] = p - 1;
    }
  }

  void run() {
    if (n == 1) {
      sa[0] = 0;
      return;
    }
    stype.assign(n, 0);
    bkt.assign(K, 0);
    stype[n - 1] = 1;
    for (int32_t i = n - 2; i >= 0; --i)
      stype[i] = (s[i] < s[i + 1] || (s[i] == s[i + 1] && stype[i + 1])) ? 1 : 0;

    // 1. sort the LMS substrings
    std::fill(sa, sa + n, -1);
    bucket_ends();
    for (int32_t i = 1; i < n; ++i)
      if (lms(i)) sa[--bkt[(int32_t)s[i]]] = i;
    induce();

    // 2. name them
    int32_t n1 = 0;
    for (int32_t i = 0; i < n; ++i)
      if (lms(sa[i])) sa[n1++] = sa[i];
    std::fill(sa + n1, sa + n, -1);
    int32_t names = 0, prev = -1;
    for (int32_t i = 0; i < n1; ++i) {
      int32_t pos = sa[i];
      bool diff = (prev < 0);
      for (int32_t d = 0; !diff; ++d) {
        if (s[pos + d] != s[prev + d] || stype[pos + d] != stype[prev + d]) {
          diff = true;
        } else if (d > 0 && (lms(pos + d) || lms(prev + d))) {
          break;
        }
      }
      if (diff) {
        ++names;
        prev = pos;
      }
      sa[n1 + (pos >> 1)] = names - 1;
    }
    for (int32_t i = n - 1, j = n - 1; i >= n1; --i)
      if (sa[i] >= 0) sa[j--] = sa[i];

    // 3. order the LMS suffixes (recursively when names collide)
    int32_t* sa1 = sa;
    int32_t* s1 = sa + (n - n1);
    if (names < n1) {
      Sais<int32_t> sub{s1, sa1, n1, names, {}, {}};
      sub.run();
    } else {
      for (int32_t i = 0; i < n1; ++i) sa1[s1[i]] = i;
    }

    // 4. induce the full order from the sorted LMS suffixes
    for (int32_t i = 1, j = 0; i < n; ++i)
      if (lms(i)) s1[j++] = i;
    for (int32_t i = 0; i < n1; ++i) sa1[i] = s1[sa1[i]];
    std::fill(sa + n1, sa + n, -1);
    bucket_ends();
    for (int32_t i = n1 - 1; i >= 0; --i) {
      int32_t p = sa[i];
      sa[i] = -1;
      sa[--bkt[(int32_t)s[p]]] = p;
    }
    induce();
  }
};

}  // namespace

void suffix_array(const int32_t* s, int32_t* sa, int32_t n, int32_t K) {
  Sais<int32_t> top{s, sa, n, K, {}, {}};
  top.run();
}

// ---------------------------------------------------------------------------
// FASTA
// ---------------------------------------------------------------------------
void read_fasta(const std::string& path, std::vector<std::string>& names,
                std::vector<std::string>& seqs) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open FASTA " + path);
  std::string line;
  bool have = false;
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == ' '))
      line.pop_back();
    if (line.empty()) continue;
    if (line[0] == '>') {
      size_t e = 1;
      while (e < line.size() && line[e] != ' ' && line[e] != '\t') ++e;
      names.emplace_back(line.substr(1, e - 1));
      seqs.emplace_back();
      have = true;
    } else {
      if (!have) throw std::runtime_error("FASTA " + path + ": sequence before first header");
      seqs.back() += line;
    }
  }
}

// ---------------------------------------------------------------------------
// Index construction
// ---------------------------------------------------------------------------
static inline int base_code(char ch) {
  switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

static void finish_from_codes(const std::vector<uint8_t>& codes, FmIndex& ix,
                              std::vector<uint32_t>& sa32) {
  // codes: text bases 0..3, length n.
  const uint32_t n = (uint32_t)codes.size();
  ix.n = n;

  // suffix array of codes+1 followed by sentinel 0
  {
    std::vector<uint8_t> t(n + 1);
    for (uint32_t i = 0; i < n; ++i) t[i] = codes[i] + 1;
    t[n] = 0;
    std::vector<int32_t> sa(n + 1);
    Sais<uint8_t> top{t.data(), sa.data(), (int32_t)(n + 1), 5, {}, {}};
    top.run();
    sa32.resize(n + 1);
    for (uint32_t i = 0; i <= n; ++i) sa32[i] = (uint32_t)sa[i];
  }

  // symbol totals first: the superblock entries carry C[c]
  uint32_t tot[4] = {0, 0, 0, 0};
  for (uint32_t i = 0; i < n; ++i) ++tot[codes[i]];
  uint32_t sum = 1;  // row 0 is the sentinel suffix
  for (int c = 0; c < 4; ++c) {
    ix.C[c] = sum;
    sum += tot[c];
  }

  // BWT -> occ blocks + superblocks
  const uint32_t m = n + 1;
  const uint32_t nblk = (m >> 5) + 1;
  const uint32_t nsup = (m >> kSuperShift) + 1;
  ix.blocks.assign(nblk, OccBlock{{0, 0, 0, 0}, 0, 0});
  ix.super.assign((size_t)nsup * 4, 0);
  uint32_t run[4] = {0, 0, 0, 0}, sup_base[4] = {0, 0, 0, 0};
  ix.primary = 0;
  for (uint32_t i = 0; i <= m; ++i) {
    if ((i & ((1u << kSuperShift) - 1)) == 0 && (i >> kSuperShift) < nsup) {
      for (int c = 0; c < 4; ++c) {
        sup_base[c] = run[c];
        ix.super[(size_t)(i >> kSuperShift) * 4 + c] = ix.C[c] + run[c];
      }
    }
    if ((i & 31) == 0 && (i >> 5) < nblk) {
      OccBlock& b = ix.blocks[i >> 5];
      for (int c = 0; c < 4; ++c) b.cnt[c] = (uint16_t)(run[c] - sup_base[c]);
    }
    if (i >= m) break;
    uint32_t p = sa32[i];
    if (p == 0) {
      ix.primary = i;  // sentinel row: stored as symbol 0, never counted
      continue;
    }
    uint32_t c = codes[p - 1];
    OccBlock& b = ix.blocks[i >> 5];
    b.lo |= (uint32_t)(c & 1) << (i & 31);
    b.hi |= (uint32_t)((c >> 1) & 1) << (i & 31);
    ++run[c];
  }

  // packed text, padded so a 64-bit window can be read at any base
  const uint32_t words = (n + 15) / 16 + 4;
  ix.text.assign(words, 0);
  for (uint32_t p = 0; p < n; ++p)
    ix.text[p >> 4] |= (uint32_t)codes[p] << ((p & 15) * 2);
}

// k-mer jump tables.  Rows are sorted by suffix, so with the k-mers numbered in the same
// (lexicographic) order -- first base most significant -- the rows starting with k-mer c are
// [T[c], T[c+1]): ONE 4-byte boundary per k-mer.  A suffix shorter than k (at most k-1 rows, at
// the very end of the text) sorts just before the k-mers it prefixes and is taken as padded with
// A, so it lands in the interval of the first of them: a candidate that verification discards.
// A seed piece uses the largest table it is long enough for:
//   ks[0]  "big"  k = ceil(log4 n) when that exceeds 11 (12..14): for whole-read seeds
//          (`-n 0`) on large libraries the interval is then about one row and no LF step --
//          two random 16-byte block loads each -- is left;  4^k + 1 words (1.07 GB at k = 14).
//          Small libraries use the slot for k = main + 1 (<= 11): the 11-base pieces of the
//          1-mismatch passes then meet ~4x fewer false rows (passes 2 and 4: -0.06 ms each),
//          while the 9-10-base pieces of the 2-mismatch pass still have the main table
//   ks[1]  "main" k = ceil(log4 n) clamped to 8..11 (11 = a seed piece of a 22-nt read)
//   ks[2], ks[3]  k = 6 and k = 4 for the short pieces of the 2-mismatch pass (6-7 of 19 nt)
// Derived data: rebuilt on load, not stored in the index file.
namespace {

// 32 bases starting at text position q (first base in the low two bits); positions past the end
// read as A (the text array is padded with zero words).
inline uint64_t window64(const FmIndex& ix, uint64_t q) {
  const size_t w = q >> 4;
  const uint32_t sh = (uint32_t)(q & 15) * 2;
  const size_t nw = ix.text.size();
  const uint64_t t0 = w < nw ? ix.text[w] : 0, t1 = w + 1 < nw ? ix.text[w + 1] : 0, t2 = w + 2 < nw ? ix.text[w + 2] : 0;
  const uint64_t lo = t0 | (t1 << 32);
  return sh ? (lo >> sh) | (t2 << (64 - sh)) : lo;
}

// reverse the order of the low k 2-bit groups of x (first base becomes most significant)
inline uint64_t reverse_pairs(uint64_t x, uint32_t k) {
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
  x = __builtin_bswap64(x);
  return x >> (64 - 2 * k);
}

void fill_jump_table(const FmIndex& ix, uint32_t k, uint32_t* tab) {
  const uint64_t n_codes = 1ull << (2 * k), kmask = n_codes - 1;
  const uint32_t n_rows = (uint32_t)ix.sa.size();
  uint64_t next_c = 0;
  for (uint32_t i = 0; i < n_rows; ++i) {
    const uint32_t p = (uint32_t)ix.sa[i];
    // first base most significant; a suffix shorter than k is padded with A (bases past the end
    // of the text read as A, but the last text word may hold real padding only past n)
    uint64_t win = window64(ix, p) & kmask;
    if ((uint64_t)p + k > ix.n) {
      const uint32_t have = p < ix.n ? ix.n - p : 0;
      win &= have ? ((1ull << (2 * have)) - 1) : 0ull;
    }
    const uint64_t code = reverse_pairs(win, k);
    while (next_c <= code) tab[next_c++] = i;
  }
  while (next_c <= n_codes) tab[next_c++] = n_rows;
}

}  // namespace

void plan_jump_tables(FmIndex& ix) {
  uint32_t k_log = 1;
  while (k_log < 14 && (1ull << (2 * k_log)) < ix.n) ++k_log;
  ix.ftab_ks[1] = (uint8_t)std::min(11u, std::max(8u, k_log));
  ix.ftab_ks[0] = k_log > 11 ? (uint8_t)k_log : (ix.ftab_ks[1] < 11 ? (uint8_t)(ix.ftab_ks[1] + 1) : 0);
  ix.ftab_ks[2] = 6;
  ix.ftab_ks[3] = 4;
}

void derive_tables(FmIndex& ix) {
  if (ix.derived) return;
  StageTimer tm("derive_tables (host)");
  build_jump_tables(ix);
  tm.lap("jump tables");
  build_row_context(ix);
  tm.lap("row context");
  ix.derived = true;
}

void build_jump_tables(FmIndex& ix) {
  plan_jump_tables(ix);
  size_t total = 0, base[4] = {0, 0, 0, 0};
  for (int t = 0; t < 4; ++t) {
    base[t] = total;
    if (ix.ftab_ks[t]) total += ((size_t)1 << (2 * ix.ftab_ks[t])) + 1;
  }
  ix.ftab.assign(total, 0);
  // the tables are independent linear passes over the suffix array: one thread each
  std::vector<std::thread> pool;
  for (int t = 0; t < 4; ++t)
    if (ix.ftab_ks[t])
      pool.emplace_back(fill_jump_table, std::cref(ix), (uint32_t)ix.ftab_ks[t], ix.ftab.data() + base[t]);
  for (auto& th : pool) th.join();
}

void build_kmer_bits(FmIndex& ix) {
  ix.kbits.clear();
  if (ix.n > kKmerBitsMaxBases || ix.n < kIndexKmerBitsK) return;
  ix.kbits.assign(kIndexKmerBitsWords, 0);
  const uint32_t mask = (1u << (2 * kIndexKmerBitsK)) - 1u;
  for (uint32_t p = 0; p + kIndexKmerBitsK <= ix.n; ++p) {
    const uint32_t c = (uint32_t)window64(ix, p) & mask;
    ix.kbits[c >> 5] |= 1u << (c & 31);
  }
}

void build_row_context(FmIndex& ix) {
  ix.ctx.clear();
  if (ix.n < (1u << 20)) return;
  ix.ctx.resize(ix.sa.size());
  const size_t n_rows = ix.sa.size();
  const unsigned n_threads = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < n_threads; ++t)
    pool.emplace_back([&ix, n_rows, n_threads, t] {
      for (size_t i = n_rows * t / n_threads; i < n_rows * (t + 1) / n_threads; ++i) {
        const uint32_t p = (uint32_t)ix.sa[i];
        // left: text[p-8 .. p) with text[p-1] in the top two bits = the 16-bit window at p - 8
        // (bases before the start of the text read as A)
        uint32_t left;
        if (p >= 8) {
          left = (uint32_t)window64(ix, (uint64_t)p - 8) & 0xFFFFu;
        } else {
          left = ((uint32_t)window64(ix, 0) & ((1u << (2 * p)) - 1u)) << (16 - 2 * p);
        }
        uint32_t right = (uint32_t)window64(ix, (uint64_t)p + 8) & 0xFFFFu;
        const uint64_t end = (uint64_t)p + 8;
        if (end >= ix.n) right = 0;
        else if (end + 8 > ix.n) right &= (1u << (2 * (uint32_t)(ix.n - end))) - 1u;
        ix.ctx[i] = left | right << 16;
      }
    });
  for (auto& th : pool) th.join();
}

namespace {
// one wide row (fm_index.hpp: fill_wide_rows): the 8-byte row, the 16 bases left of its position, the 16 bases from + 8
inline void wide_row_of(const FmIndex& ix, uint64_t row, uint32_t* o) {
  const uint32_t p = (uint32_t)row;
  // left: text[p-16 .. p), text[p-1] in the top two bits (bases before the text read as A)
  uint32_t left;
  if (p >= 16) {
    left = (uint32_t)window64(ix, (uint64_t)p - 16);
  } else {
    left = p ? ((uint32_t)window64(ix, 0) & (uint32_t)((1ull << (2 * p)) - 1ull)) << (32 - 2 * p) : 0u;
  }
  // right: text[p+8 .. p+24), text[p+8] in the low two bits (bases past the end read as A)
  uint32_t right = (uint32_t)window64(ix, (uint64_t)p + kWideRowRightSkip);
  const uint64_t start = (uint64_t)p + kWideRowRightSkip;
  if (start >= ix.n) right = 0;
  else if (start + 16 > ix.n) right &= (uint32_t)((1ull << (2 * (ix.n - start))) - 1ull);
  o[0] = (uint32_t)row;
  o[1] = (uint32_t)(row >> 32);
  o[2] = left;
  o[3] = right;
}
}  // namespace

void fill_wide_rows(const FmIndex& ix, size_t row_lo, size_t row_hi, uint32_t* out) {
  const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> pool;
  const size_t n = row_hi - row_lo;
  for (unsigned t = 0; t < n_threads; ++t)
    pool.emplace_back([&ix, row_lo, n, n_threads, t, out] {
      for (size_t k = n * t / n_threads; k < n * (t + 1) / n_threads; ++k) wide_row_of(ix, ix.sa[row_lo + k], out + 4 * k);
    });
  for (auto& th : pool) th.join();
}

uint32_t seed_bucket_k(const FmIndex& ix) {
  // a bucket holds kSeedBucketRows rows: worth it when a k-mer has a few rows on average, and the
  // index must have the jump table of that k (overflowing buckets fall back to it)
  const uint32_t k = kSeedBucketK;
  const double fill = (double)ix.n / (double)(1ull << (2 * k));
  bool has = false;
  for (int t = 0; t < 4; ++t) has |= ix.ftab_ks[t] == k;
  return (has && fill >= 0.25 && fill <= 4.0) ? k : 0u;
}

namespace {
const uint32_t* jump_table_of(const FmIndex& ix, uint32_t k) {
  size_t tab_base = 0;
  bool found = false;
  for (int t = 0; t < 4 && !found; ++t) {
    if (ix.ftab_ks[t] == k) found = true;
    else if (ix.ftab_ks[t]) tab_base += ((size_t)1 << (2 * ix.ftab_ks[t])) + 1;
  }
  if (!found) throw std::runtime_error("seed buckets: the index has no jump table of that k");
  return ix.ftab.data() + tab_base;
}
}  // namespace

void wide_row_of_row(const FmIndex& ix, uint64_t row, uint32_t* out4) { wide_row_of(ix, row, out4); }

void seed_pos_lists(const FmIndex& ix, uint32_t k, std::vector<uint32_t>& start_by_lex, std::vector<uint32_t>& positions) {
  const uint32_t* tab = jump_table_of(ix, k);
  const uint64_t n_codes = 1ull << (2 * k);
  start_by_lex.assign(n_codes, 0u);
  positions.clear();
  for (uint64_t c = 0; c < n_codes; ++c) {
    start_by_lex[c] = (uint32_t)positions.size();
    const uint32_t lo = tab[c], hi = tab[c + 1];
    if (hi - lo <= kSeedBucketRows) continue;
    const size_t at = positions.size();
    for (uint32_t i = lo; i < hi; ++i) positions.push_back((uint32_t)ix.sa[i]);
    std::sort(positions.begin() + at, positions.end());
  }
}

void fill_seed_buckets(const FmIndex& ix, uint32_t k, uint64_t code_lo, uint64_t code_hi, uint32_t* out, const uint32_t* over_start) {
  const uint32_t* tab = jump_table_of(ix, k);
  const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> pool;
  const uint64_t n = code_hi - code_lo;
  for (unsigned t = 0; t < n_threads; ++t)
    pool.emplace_back([&ix, tab, k, code_lo, n, n_threads, t, out, over_start] {
      for (uint64_t q = n * t / n_threads; q < n * (t + 1) / n_threads; ++q) {
        const uint64_t code = code_lo + q;                  // first base in the low two bits
        const uint64_t lex = reverse_pairs(code, k);        // the jump table's numbering
        const uint32_t lo = tab[lex], hi = tab[lex + 1];
        uint32_t* b = out + q * (4u * kSeedBucketRows);
        const uint32_t cnt = hi - lo;
        for (uint32_t i = 0; i < kSeedBucketRows; ++i) {
          uint32_t* o = b + 4 * i;
          if (cnt <= kSeedBucketRows && i < cnt) {
            wide_row_of(ix, ix.sa[lo + i], o);
            const uint32_t before = std::min<uint32_t>(63u, o[1] & 255u), after = std::min<uint32_t>(63u, (o[1] >> 8) & 255u);
            o[1] = before | (after << 6) | (o[1] & 0xFFFF0000u);
          } else {
            o[0] = 0xFFFFFFFFu;
            o[1] = o[2] = o[3] = 0u;
          }
        }
        b[1] |= (cnt <= kSeedBucketRows ? cnt : kSeedBucketOverflow) << 12;
        if (cnt > kSeedBucketRows && over_start) {  // the k-mer's position list (seed_pos_lists)
          b[2] = over_start[lex];
          b[3] = cnt;
        }
      }
    });
  for (auto& th : pool) th.join();
}

void build_pair_tables(const FmIndex& ix, uint32_t anchor, PairTables& out) {
  out = PairTables();
  if (anchor < 2 || anchor > 6) throw std::runtime_error("pair tables: anchor length must be 2..6");
  out.anchor = anchor;
  const uint32_t kb = 2 * anchor;
  const uint64_t amask = (1ull << kb) - 1ull;
  const size_t n_codes = (size_t)1 << (2 * kb);
  out.jump.assign(3 * (n_codes + 1), 0u);
  const size_t n_rows = ix.sa.size();
  // a row can only matter when both anchors lie inside its N-free segment
  auto key_of = [&](uint64_t row, uint32_t d, uint32_t& key) -> bool {
    const uint32_t p = (uint32_t)row, after = (uint32_t)(row >> 40) & 255u;
    if (after < d + anchor) return false;
    key = (uint32_t)((window64(ix, p) & amask) | ((window64(ix, (uint64_t)p + d) & amask) << kb));
    return true;
  };
  std::vector<uint32_t> fill(n_codes);
  for (uint32_t t = 0; t < 3; ++t) {
    const uint32_t d = (t + 1) * anchor;
    uint32_t* jump = out.jump.data() + (size_t)t * (n_codes + 1);
    uint32_t key;
    for (size_t i = 0; i < n_rows; ++i)
      if (key_of(ix.sa[i], d, key)) ++jump[key + 1];
    for (size_t c = 0; c < n_codes; ++c) jump[c + 1] += jump[c];
    out.row_off[t] = (uint32_t)out.rows.size();
    out.rows.resize(out.rows.size() + jump[n_codes]);
    uint64_t* rows = out.rows.data() + out.row_off[t];
    std::copy(jump, jump + n_codes, fill.begin());
    for (size_t i = 0; i < n_rows; ++i)
      if (key_of(ix.sa[i], d, key)) rows[fill[key]++] = ix.sa[i];
  }
  out.row_off[3] = (uint32_t)out.rows.size();
}

void build_index(const std::vector<std::string>& names,
                 const std::vector<std::string>& seqs, FmIndex& ix) {
  if (names.size() != seqs.size()) throw std::runtime_error("names/seqs size mismatch");
  ix = FmIndex();
  ix.names = names;
  uint64_t total = 0;
  for (auto& s : seqs) total += s.size();
  if (total >= 0x7ffffff0ull) throw std::runtime_error("library too large for 32-bit index");

  std::vector<uint8_t> codes;
  codes.reserve(total);
  ix.ref_len.resize(seqs.size());
  ix.ref_n_runs.resize(seqs.size());
  for (size_t r = 0; r < seqs.size(); ++r) {
    const std::string& s = seqs[r];
    ix.ref_len[r] = (uint32_t)s.size();
    uint32_t i = 0;
    const uint32_t L = (uint32_t)s.size();
    while (i < L) {
      if (base_code(s[i]) < 0) {
        uint32_t j = i;
        while (j < L && base_code(s[j]) < 0) ++j;
        ix.ref_n_runs[r].push_back(i);
        ix.ref_n_runs[r].push_back(j - i);
        i = j;
      } else {
        uint32_t j = i;
        ix.seg_start.push_back((uint32_t)codes.size());
        ix.seg_ref.push_back((uint32_t)r);
        ix.seg_off.push_back(i);
        while (j < L) {
          int c = base_code(s[j]);
          if (c < 0) break;
          codes.push_back((uint8_t)c);
          ++j;
        }
        i = j;
      }
    }
  }
  ix.seg_start.push_back((uint32_t)codes.size());

  std::vector<uint32_t> sa32;
  finish_from_codes(codes, ix, sa32);

  // chunk -> segment map for O(1) locate
  const uint32_t nchunk = (ix.n >> 5) + 2;
  ix.chunk_seg.assign(nchunk, 0);
  const uint32_t nseg = (uint32_t)ix.seg_ref.size();
  uint32_t sgi = 0;
  for (uint32_t ch = 0; ch < nchunk; ++ch) {
    uint64_t p = (uint64_t)ch << 5;
    while (sgi + 1 < nseg && ix.seg_start[sgi + 1] <= p) ++sgi;
    ix.chunk_seg[ch] = sgi;
  }

  // 8-byte suffix-array rows: position + distance to both ends of its segment
  std::vector<uint32_t> seg_of(ix.n);
  for (uint32_t sg = 0; sg < nseg; ++sg)
    for (uint32_t p = ix.seg_start[sg]; p < ix.seg_start[sg + 1]; ++p) seg_of[p] = sg;
  ix.sa.resize(sa32.size());
  for (size_t i = 0; i < sa32.size(); ++i) {
    const uint32_t p = sa32[i];
    uint64_t row = p;
    if (p < ix.n) {
      const uint32_t sg = seg_of[p];
      const uint32_t before = std::min<uint32_t>(255u, p - ix.seg_start[sg]);
      const uint32_t after = std::min<uint32_t>(255u, ix.seg_start[sg + 1] - p);
      const uint32_t sid = nseg <= 0xFFFFu ? sg : 0xFFFFu;
      row |= (uint64_t)before << 32 | (uint64_t)after << 40 | (uint64_t)sid << 48;
    } else {
      row |= (uint64_t)0xFFFFu << 48;
    }
    ix.sa[i] = row;
  }
  build_kmer_bits(ix);
  if (ix.n >= kLazyDeriveBases) {
    plan_jump_tables(ix);
    ix.derived = false;
  } else {
    build_jump_tables(ix);
    build_row_context(ix);
  }
}

std::string entry_sequence(const FmIndex& ix, uint32_t r) {
  std::string out(ix.ref_len[r], 'N');
  static const char L[4] = {'A', 'C', 'G', 'T'};
  // segments of this entry are contiguous in seg_ref; find the first by scan
  auto it = std::lower_bound(ix.seg_ref.begin(), ix.seg_ref.end(), r);
  for (size_t sg = it - ix.seg_ref.begin(); sg < ix.seg_ref.size() && ix.seg_ref[sg] == r; ++sg) {
    uint32_t a = ix.seg_start[sg], b = ix.seg_start[sg + 1], off = ix.seg_off[sg];
    for (uint32_t p = a; p < b; ++p)
      out[off + (p - a)] = L[(ix.text[p >> 4] >> ((p & 15) * 2)) & 3];
  }
  return out;
}

// ---------------------------------------------------------------------------
// Serialisation ("MRGFM5\0\0" + counts + raw arrays; the jump tables are rebuilt on load)
// ---------------------------------------------------------------------------
namespace {
const char kMagic[8] = {'M', 'R', 'G', 'F', 'M', '5', 0, 0};

template <class T>
void put_vec(std::ofstream& o, const std::vector<T>& v) {
  uint64_t n = v.size();
  o.write((const char*)&n, 8);
  if (n) o.write((const char*)v.data(), (std::streamsize)(n * sizeof(T)));
}
template <class T>
void get_vec(std::ifstream& in, std::vector<T>& v) {
  uint64_t n = 0;
  in.read((char*)&n, 8);
  if (!in || n > (1ull << 34)) throw std::runtime_error("index file truncated or corrupt");
  v.resize(n);
  if (n) in.read((char*)v.data(), (std::streamsize)(n * sizeof(T)));
  if (!in) throw std::runtime_error("index file truncated");
}
}  // namespace

void save_index(const FmIndex& ix, const std::string& path) {
  std::ofstream o(path, std::ios::binary);
  if (!o) throw std::runtime_error("cannot write " + path);
  o.write(kMagic, 8);
  uint32_t hdr[8] = {ix.n, ix.primary, ix.C[0], ix.C[1], ix.C[2], ix.C[3],
                     (uint32_t)ix.names.size(), 0u};
  o.write((const char*)hdr, sizeof(hdr));
  for (size_t r = 0; r < ix.names.size(); ++r) {
    uint32_t l = (uint32_t)ix.names[r].size();
    o.write((const char*)&l, 4);
    o.write(ix.names[r].data(), l);
    put_vec(o, ix.ref_n_runs[r]);
  }
  put_vec(o, ix.ref_len);
  put_vec(o, ix.blocks);
  put_vec(o, ix.super);
  put_vec(o, ix.text);
  put_vec(o, ix.sa);
  put_vec(o, ix.seg_start);
  put_vec(o, ix.seg_ref);
  put_vec(o, ix.seg_off);
  put_vec(o, ix.chunk_seg);
  if (!o) throw std::runtime_error("short write to " + path);
}

void load_index(const std::string& path, FmIndex& ix) {
  StageTimer tm("load_index");
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + path);
  char magic[8];
  in.read(magic, 8);
  if (!in || std::memcmp(magic, kMagic, 8) != 0)
    throw std::runtime_error(path + " is not a mirge_amd index");
  uint32_t hdr[8];
  in.read((char*)hdr, sizeof(hdr));
  ix = FmIndex();
  ix.n = hdr[0];
  ix.primary = hdr[1];
  for (int c = 0; c < 4; ++c) ix.C[c] = hdr[2 + c];
  uint32_t nref = hdr[6];
  ix.names.resize(nref);
  ix.ref_n_runs.resize(nref);
  for (uint32_t r = 0; r < nref; ++r) {
    uint32_t l = 0;
    in.read((char*)&l, 4);
    if (!in || l > (1u << 20)) throw std::runtime_error("index file corrupt (name)");
    ix.names[r].resize(l);
    in.read(&ix.names[r][0], l);
    get_vec(in, ix.ref_n_runs[r]);
  }
  get_vec(in, ix.ref_len);
  get_vec(in, ix.blocks);
  get_vec(in, ix.super);
  get_vec(in, ix.text);
  get_vec(in, ix.sa);
  get_vec(in, ix.seg_start);
  get_vec(in, ix.seg_ref);
  get_vec(in, ix.seg_off);
  get_vec(in, ix.chunk_seg);
  tm.lap("read file");
  if (ix.sa.size() != (size_t)ix.n + 1 || ix.blocks.size() != (size_t)((ix.n + 1) >> 5) + 1 ||
      ix.super.size() != ((size_t)((ix.n + 1) >> kSuperShift) + 1) * 4 ||
      ix.text.size() < (size_t)(ix.n >> 4) + 3)
    throw std::runtime_error("index file inconsistent");
  // the segment tables and `primary` are device indices later: a corrupt file must not become
  // out-of-bounds reads on the GPU
  const size_t n_seg = ix.seg_ref.size();
  bool ok = ix.primary <= ix.n && ix.ref_len.size() == nref && ix.seg_start.size() == n_seg + 1 &&
            ix.seg_off.size() == n_seg && ix.chunk_seg.size() == (size_t)(ix.n >> 5) + 2 &&
            (n_seg == 0 ? ix.n == 0 : (ix.seg_start.front() == 0 && ix.seg_start.back() == ix.n));
  for (size_t sg = 0; ok && sg < n_seg; ++sg)
    ok = ix.seg_start[sg] <= ix.seg_start[sg + 1] && ix.seg_ref[sg] < nref &&
         (uint64_t)ix.seg_off[sg] + (ix.seg_start[sg + 1] - ix.seg_start[sg]) <= ix.ref_len[ix.seg_ref[sg]];
  for (size_t c = 0; ok && c < ix.chunk_seg.size(); ++c) ok = n_seg == 0 || ix.chunk_seg[c] < n_seg;
  for (size_t i = 0; ok && i < ix.sa.size(); ++i) ok = (uint32_t)ix.sa[i] <= ix.n;
  if (!ok) throw std::runtime_error("index file inconsistent (segment tables)");
  tm.lap("validate");
  build_kmer_bits(ix);
  if (ix.n >= kLazyDeriveBases) {
    plan_jump_tables(ix);
    ix.derived = false;
  } else {
    build_jump_tables(ix);
    build_row_context(ix);
  }
  tm.lap("derived tables of a small library");
}

}  // namespace mrg
