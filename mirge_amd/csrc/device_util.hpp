// Device helpers shared by the dictionary kernels (dict.hip).  gfx950, wave64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mrg {
namespace dev {

constexpr uint64_t kOddBits = 0x5555555555555555ull;

__device__ __forceinline__ uint64_t low_bits(uint32_t nbits) {  // nbits in [0,64]
  return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
}

__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {  // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// k bases packed first-base-lowest -> their lexicographic number (first base most significant)
__device__ __forceinline__ uint32_t lex_code(uint64_t code, uint32_t k) {
  const uint64_t r = __brevll(code) >> (64u - 2u * k);  // groups reversed, bits inside a group swapped
  return (uint32_t)(((r >> 1) & kOddBits) | ((r & kOddBits) << 1));
}

// The kernel's (single, by-value) parameter block where it sits, in the kernel-argument segment, through
// a pointer laundered by an empty asm: fields read through it are scalar loads issued HERE, every time --
// for values a loop needs once per trip but that, kept in registers across it, are spilled (to VGPR
// lanes, or worse to scratch, whose reloads wait with vmcnt(0) and drain the loads in flight).
template <typename Params>
__device__ __forceinline__ const __attribute__((address_space(4))) Params* kernel_args_here() {
  typedef const __attribute__((address_space(4))) Params* kernarg_ptr_t;
  kernarg_ptr_t k = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return k;
}

// The largest jump table with k <= plen (tables in ascending k; k[0] == 0: none usable): its k
// (0 = none) and its word offset inside `ftab`.
template <class Tabs>  // (JumpTables in any address space)
__device__ __forceinline__ uint32_t pick_table(const Tabs& t, int32_t plen, uint32_t& word_off) {
  uint32_t k = plen >= (int32_t)t.k[0] ? t.k[0] : 0u;
  word_off = t.off[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    const bool take = plen >= (int32_t)t.k[i];
    k = take ? t.k[i] : k;
    word_off = take ? t.off[i] : word_off;
  }
  return k;
}

// The jump table a seed of k bases is looked up in without LF steps: the smallest table with K >= k --
// the seed's rows are those of the 4^(K - k) K-mers it is a prefix of, the interval between two
// entries of that table 4^(K - k) apart (`shift` = 2 (K - k); all k bases are then known to match) --
// else the largest table with K <= k (shift = 0, the first K bases match).  Returns K (0 = none).
template <class Tabs>
__device__ __forceinline__ uint32_t pick_seed_table(const Tabs& t, int32_t k, uint32_t& word_off, uint32_t& shift) {
  uint32_t K = 0;
  word_off = 0;
  shift = 0;
#pragma unroll
  for (int i = 3; i >= 0; --i) {  // descending: the last table taken is the smallest one that holds the seed
    const bool take = t.k[i] != 0u && (int32_t)t.k[i] >= k;
    K = take ? t.k[i] : K;
    word_off = take ? t.off[i] : word_off;
  }
  if (K) {
    shift = 2u * (K - (uint32_t)k);
    return K;
  }
  return pick_table(t, k, word_off);
}

// 32 text bases from position p (2-bit packed text in global memory, 3 dword loads)
// (ONE 12-byte request: the three words are consecutive and 4-byte aligned, which is all a multi-dword global load needs;
// as three dword loads every candidate verification cost three L1 / L2 requests)
struct __attribute__((packed, aligned(4))) TextWords3 {
  uint32_t w0, w1, w2;
};
struct __attribute__((packed, aligned(4))) TableEntry2 {  // two neighbouring boundaries of a jump table: one 8-byte request
  uint32_t lo, hi;
};
__device__ __forceinline__ uint64_t text_window(const uint32_t* __restrict__ text, uint32_t p) {
  const uint32_t i = p >> 4, sh = (p & 15) * 2;
  const TextWords3 t3 = *reinterpret_cast<const TextWords3*>(text + i);
  const uint32_t w0 = t3.w0, w1 = t3.w1, w2 = t3.w2;
  const uint64_t lo64 = (uint64_t)w0 | ((uint64_t)w1 << 32);
  return (lo64 >> sh) | ((((uint64_t)w2) << 1) << (63 - sh));
}

// mismatch mask (one bit per differing base, in the even bit positions) of two packed sequences
__device__ __forceinline__ uint64_t mismatch_bits(uint64_t a, uint64_t b) {
  const uint64_t x = a ^ b;
  return (x | (x >> 1)) & kOddBits;
}

// number of trailing T (code 3) of a one-word read of L bases (runAnnotationPipeline.py:664-676)
__device__ __forceinline__ int32_t trailing_t(uint64_t rd, int32_t L) {
  const uint64_t x = ~rd & low_bits(2 * (uint32_t)L);  // non-zero 2-bit group = not T
  const int32_t hb = x ? ((63 - __clzll((long long)x)) >> 1) : -1;
  return L - 1 - hb;
}

// inclusive prefix sum over the wave (seven DPP adds)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int x = (int)v;
  int s = x + __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  s += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);         // row_shr:2
  s += __builtin_amdgcn_update_dpp(0, x, 0x113, 0xf, 0xf, false);         // row_shr:3
  s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xe, false);         // row_shr:4, lanes 4-15 of a row
  s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xc, false);         // row_shr:8, lanes 8-15
  s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);         // row_bcast:15 into rows 1 and 3
  s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);         // row_bcast:31 into rows 2 and 3
  return (uint32_t)s;
}

// first BWT row of the c-suffixes + rank of c before row i, from the 16-byte occ block of row i and the superblock
// table in global memory (kernels.hip: Lib::lf with both tables in LDS)
__device__ __forceinline__ uint32_t lf_step(const uint32_t* __restrict__ blocks, const uint32_t* __restrict__ super, uint32_t primary,
                                            uint32_t c, uint32_t i) {
  const uint4 v = *reinterpret_cast<const uint4*>(blocks + (size_t)(i >> 5) * 4u);
  const uint32_t r = i & 31u;
  const uint32_t pair = (c & 2u) ? v.y : v.x;
  const uint32_t cnt = (c & 1u) ? (pair >> 16) : (pair & 0xffffu);
  uint32_t e = ((c & 1u) ? v.z : ~v.z) & ((c & 2u) ? v.w : ~v.w);
  e &= (1u << r) - 1u;
  uint32_t o = super[(size_t)(i >> 16) * 4u + c] + cnt + (uint32_t)__popc(e);
  o -= (uint32_t)((c == 0u) & (i > primary) & ((i >> 5) == (primary >> 5)));  // the sentinel row is stored as symbol 0
  return o;
}

// The FM-index answer for a one-word read whose first R bases must match letter for letter (a pass without seed
// mismatch; up to max_total mismatches behind base R): what the dictionary kernels fall back to for a read shorter than
// the dictionary's key or a key whose chain overflowed.  A BACKWARD search of the R bases -- their last k through the
// largest jump table, the others one LF step each -- leaves exactly the rows whose suffix starts with them: one row
// for a read that reaches out of a repeat into unique sequence, however many copies the repeat has.  (Round 4 took the
// rows of the first k bases and compared every one with the text, one lane, one row after the other: 10^5 rows for a
// key inside an element with 10^5 copies -- 77 ms for 2 M reads on `bench.py --workload repeats`.)
// Returns the best (mismatches << 32 | text position) or ~0; rows = rows looked at, steps = LF steps made.
// (LONG: reads of up to 63 bases, q = bases 0..31, qh = bases 32..63)
template <bool LONG = false, class Tabs>
__device__ __forceinline__ uint64_t fm_exact_search(const uint32_t* __restrict__ blocks, const uint32_t* __restrict__ super, uint32_t primary,
                                                    const uint32_t* __restrict__ ftab, const Tabs& tabs, const uint64_t* __restrict__ sa,
                                                    const uint32_t* __restrict__ text, uint32_t n, uint64_t q, int32_t L, int32_t R,
                                                    int32_t max_total, uint32_t& best_seg, uint32_t& best_before, uint32_t& rows,
                                                    uint32_t& steps, uint32_t rows_cap = 0xFFFFFFFFu, uint64_t qh = 0ull) {
  uint32_t tab_off = 0;
  const uint32_t k = tabs.k[0] ? pick_table(tabs, R, tab_off) : 0u;
  uint32_t lo = 0, hi = n + 1u;
  int32_t j = R;
  // the bases from read offset j on (32 of them)
  auto from = [&](int32_t at) -> uint64_t {
    if (!LONG) return q >> (2u * (uint32_t)at);
    if (at >= 32) return qh >> (2u * (uint32_t)(at - 32));
    return at ? ((q >> (2u * (uint32_t)at)) | (qh << (64u - 2u * (uint32_t)at))) : q;
  };
  if (k) {
    j = R - (int32_t)k;
    const uint32_t* tab = ftab + tab_off + lex_code(from(j) & low_bits(2u * k), k);
    lo = tab[0];
    hi = tab[1];
  }
  steps = 0;
  while (j > 0 && hi > lo) {
    --j;
    const uint32_t c = (uint32_t)from(j) & 3u;
    lo = lf_step(blocks, super, primary, c, lo);
    hi = lf_step(blocks, super, primary, c, hi);
    ++steps;
  }
  rows = hi > lo ? hi - lo : 0u;
  if (rows > rows_cap) return ~0ull;  // (the caller leaves a wide interval to the whole wave)
  uint64_t best = ~0ull;
  // mismatches behind the R exact bases, inside the read: the first word's share, the second word's
  const uint64_t tailmask = LONG ? (low_bits(2u * (uint32_t)min(L, 32)) & ~low_bits(2u * (uint32_t)min(R, 32)))
                                 : (low_bits(2u * (uint32_t)L) & ~low_bits(2u * (uint32_t)R));
  const uint64_t tailmask_h = (LONG && L > 32) ? (low_bits(2u * (uint32_t)(L - 32)) & ~low_bits(2u * (uint32_t)max(R - 32, 0))) : 0ull;
  for (uint32_t i = lo; i < hi; ++i) {
    const uint64_t row = sa[i];
    if ((uint32_t)L > ((uint32_t)(row >> 40) & 255u)) continue;  // (the read would leave the N-free segment)
    const uint32_t s = (uint32_t)row;
    uint32_t mmt = 0;
    if (L > R) {
      if (!LONG || tailmask) mmt = (uint32_t)__popcll(mismatch_bits(text_window(text, s), q) & tailmask);
      if ((int32_t)mmt > max_total) continue;
      if (LONG && tailmask_h) {
        mmt += (uint32_t)__popcll(mismatch_bits(text_window(text, s + 32u), qh) & tailmask_h);
        if ((int32_t)mmt > max_total) continue;
      }
    }
    const uint64_t key = ((uint64_t)mmt << 32) | s;
    if (key < best) {
      best = key;
      best_seg = (uint32_t)(row >> 48);
      best_before = (uint32_t)(row >> 32) & 255u;
    }
  }
  return best;
}

// entry (ref, pos) of text position s whose suffix-array row said (seg16, before): the epilogue of
// every match kernel
struct SegTables {
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  uint32_t simple_segs;
};
__device__ __forceinline__ void locate_entry(const SegTables& t, uint32_t s, uint32_t seg16, uint32_t before,
                                             uint32_t& ref, uint32_t& pos) {
  uint32_t sg = seg16;
  if (sg == 0xFFFFu) {  // more than 65535 segments: walk the chunk map
    sg = t.chunk_seg[s >> 5];
    while (t.seg_start[sg + 1] <= s) ++sg;
  }
  ref = sg;
  if (t.simple_segs && before < 255u) {
    pos = before;
  } else {
    uint32_t off = 0;
    if (!t.simple_segs) {
      ref = t.seg_ref[sg];
      off = t.seg_off[sg];
    }
    pos = s - t.seg_start[sg] + off;
  }
}

}  // namespace dev
}  // namespace mrg
