// Derived tables of a large library, filled on the device from its suffix-array rows and packed text -- the other
// half of "what replaces the `.ebwt` load" (runAnnotationPipeline.py:643) beside dictbuild.hip.  They are functions of
// (text, suffix array): the index file stores neither, and the host used to rebuild them at every load (jump tables
// of the 137 Mbp mRNA library: 3.5 s of a single thread per table; row context 0.5 s; wide rows 0.5 s + their upload).
// Each kernel restates the host function it replaces (fm_index.cpp), bit for bit:
//   jump tables   fill_jump_table: tab[c] = first row whose k-base prefix code (first base most significant, a suffix
//                 shorter than k padded with A) is >= c, taken over the RUNNING MAXIMUM of the codes, as the host's
//                 single pass does (text positions inside an N run read as A and can sort out of code order).  The
//                 codes of the largest k are computed once; a shorter table's code is a prefix of it (a shift), and
//                 the running maximum commutes with the shift.
//   row context   build_row_context: 8 bases left / 8 bases from +8 of every row's position
//   wide rows     fill_wide_rows: the 8-byte row, 16 bases left, 16 bases from +8
//   seed buckets  fill_seed_buckets: the wide rows of every 11-mer with at most 8 rows, one 128-byte line per 11-mer
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_util.hpp"
#include "fm_index.hpp"
#include "kernels.hpp"
#include "prims.hpp"

namespace mrg {

namespace {

using namespace dev;

#define TCK(expr)                    \
  do {                               \
    hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return e_; \
  } while (0)

struct TabLib {
  const uint32_t* text;  // packed text, text_words readable words (zero padded)
  const uint64_t* sa;    // n + 1 rows
  uint32_t n, n_rows, text_words;
};

// fm_index.cpp: window64 -- 32 bases from text position q, positions past the array read as A
__device__ __forceinline__ uint64_t window64_dev(const TabLib& b, uint64_t q) {
  const uint64_t w = q >> 4;
  const uint32_t sh = (uint32_t)(q & 15u) * 2u;
  const uint64_t t0 = w < b.text_words ? b.text[w] : 0u, t1 = w + 1u < b.text_words ? b.text[w + 1u] : 0u,
                 t2 = w + 2u < b.text_words ? b.text[w + 2u] : 0u;
  const uint64_t lo = t0 | (t1 << 32);
  return sh ? (lo >> sh) | (t2 << (64u - sh)) : lo;
}

// lexicographic code of the k bases at the position of row i (fill_jump_table)
__device__ __forceinline__ uint32_t row_code(const TabLib& b, uint32_t i, uint32_t k) {
  const uint32_t p = (uint32_t)b.sa[i];
  uint64_t win = window64_dev(b, p) & low_bits(2u * k);
  if ((uint64_t)p + k > b.n) {
    const uint32_t have = p < b.n ? b.n - p : 0u;
    win &= have ? low_bits(2u * have) : 0ull;
  }
  return lex_code(win, k);
}

constexpr uint32_t kScanThreads = 256u, kScanPer = 16u, kScanTile = kScanThreads * kScanPer;

__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, uint32_t lane) {
#pragma unroll
  for (uint32_t off = 1; off < 64u; off <<= 1) {
    const uint32_t o = __shfl_up(v, off, 64);
    if (lane >= off) v = max(v, o);
  }
  return v;
}

// max of the workgroup's values in front of this thread (0 for the first), *total = the workgroup's max
__device__ __forceinline__ uint32_t block_excl_max(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_max(v, lane);
  if (lane == 63u) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < kScanThreads / 64u; ++w) {
    const uint32_t t = wave_tot[w];
    base = w < wave ? max(base, t) : base;
    all = max(all, t);
  }
  __syncthreads();
  *total = all;
  const uint32_t before = __shfl_up(incl, 1, 64);
  return max(base, lane ? before : 0u);
}

// pass 1: a tile's codes -> its maximum; pass 3: codes again, running maximum with the tile's prefix, in place
template <bool APPLY>
__global__ void __launch_bounds__(kScanThreads) code_max_kernel(const TabLib b, uint32_t k, uint32_t* __restrict__ tile_max,
                                                                uint32_t* __restrict__ run_max) {
  __shared__ uint32_t wave_tot[kScanThreads / 64u];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer;
  uint32_t v[kScanPer];
  uint32_t m = 0;
#pragma unroll
  for (uint32_t j = 0; j < kScanPer; ++j) {
    v[j] = base + j < b.n_rows ? row_code(b, (uint32_t)(base + j), k) : 0u;
    m = max(m, v[j]);
  }
  uint32_t total;
  uint32_t run = block_excl_max(m, wave_tot, &total);
  if (!APPLY) {
    if (threadIdx.x == 0) tile_max[blockIdx.x] = total;
    return;
  }
  run = max(run, tile_max[blockIdx.x]);  // (exclusive prefix of the tiles in front)
#pragma unroll
  for (uint32_t j = 0; j < kScanPer; ++j) {
    run = max(run, v[j]);
    if (base + j < b.n_rows) run_max[base + j] = run;
  }
}

// pass 2: exclusive running maximum of the tile maxima, one workgroup (a library has at most 2^20 tiles)
__global__ void __launch_bounds__(kScanThreads) tile_max_scan_kernel(uint32_t* __restrict__ tile_max, uint32_t n_tiles) {
  __shared__ uint32_t wave_tot[kScanThreads / 64u];
  uint32_t carry = 0;
  for (uint32_t lo = 0; lo < n_tiles; lo += kScanThreads) {
    const uint32_t i = lo + threadIdx.x;
    const uint32_t v = i < n_tiles ? tile_max[i] : 0u;
    uint32_t total;
    const uint32_t before = block_excl_max(v, wave_tot, &total);
    if (i < n_tiles) tile_max[i] = max(carry, before);
    carry = max(carry, total);
  }
}

// tab[c] = first row whose running maximum (shifted to k bases) is >= c; the last row also closes the table
__global__ void __launch_bounds__(256) jump_fill_kernel(const uint32_t* __restrict__ run_max, uint32_t n_rows, uint32_t shift, uint32_t n_codes,
                                                        uint32_t* __restrict__ tab) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_rows) return;
  const uint32_t b = run_max[i] >> shift;
  uint32_t c = i ? (run_max[i - 1u] >> shift) + 1u : 0u;
  for (; c <= b; ++c) tab[c] = i;
  if (i == n_rows - 1u)
    for (c = b + 1u; c <= n_codes; ++c) tab[c] = n_rows;
}

__global__ void __launch_bounds__(256) row_context_kernel(const TabLib b, uint32_t* __restrict__ ctx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= b.n_rows) return;
  const uint32_t p = (uint32_t)b.sa[i];
  uint32_t left;
  if (p >= 8u) left = (uint32_t)window64_dev(b, (uint64_t)p - 8u) & 0xFFFFu;
  else left = ((uint32_t)window64_dev(b, 0) & ((1u << (2u * p)) - 1u)) << (16u - 2u * p);
  uint32_t right = (uint32_t)window64_dev(b, (uint64_t)p + 8u) & 0xFFFFu;
  const uint64_t end = (uint64_t)p + 8u;
  if (end >= b.n) right = 0u;
  else if (end + 8u > b.n) right &= (1u << (2u * (uint32_t)(b.n - end))) - 1u;
  ctx[i] = left | (right << 16);
}

// fm_index.cpp: wide_row_of
__device__ __forceinline__ uint4 wide_row_dev(const TabLib& b, uint64_t row) {
  const uint32_t p = (uint32_t)row;
  uint32_t left;
  if (p >= 16u) left = (uint32_t)window64_dev(b, (uint64_t)p - 16u);
  else left = p ? ((uint32_t)window64_dev(b, 0) & (uint32_t)low_bits(2u * p)) << (32u - 2u * p) : 0u;
  uint32_t right = (uint32_t)window64_dev(b, (uint64_t)p + kWideRowRightSkip);
  const uint64_t start = (uint64_t)p + kWideRowRightSkip;
  if (start >= b.n) right = 0u;
  else if (start + 16u > b.n) right &= (uint32_t)low_bits(2u * (uint32_t)(b.n - start));
  return make_uint4((uint32_t)row, (uint32_t)(row >> 32), left, right);
}

__global__ void __launch_bounds__(256) wide_rows_kernel(const TabLib b, uint4* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < b.n_rows) out[i] = wide_row_dev(b, b.sa[i]);
}

// one thread per (k-mer, row of its bucket): fm_index.cpp: fill_seed_buckets
__global__ void __launch_bounds__(256) seed_buckets_kernel(const TabLib b, const uint32_t* __restrict__ tab, uint32_t k, uint4* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint64_t code = t / kSeedBucketRows;  // first base in the low two bits
  const uint32_t r = (uint32_t)(t % kSeedBucketRows);
  if (code >= (1ull << (2u * k))) return;
  const uint32_t lex = lex_code(code, k);
  const uint32_t lo = tab[lex], hi = tab[lex + 1u], cnt = hi - lo;
  uint4 o = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
  if (cnt <= kSeedBucketRows && r < cnt) {
    o = wide_row_dev(b, b.sa[lo + r]);
    const uint32_t before = min(63u, o.y & 255u), after = min(63u, (o.y >> 8) & 255u);
    o.y = before | (after << 6) | (o.y & 0xFFFF0000u);
  }
  if (r == 0u) o.y |= (cnt <= kSeedBucketRows ? cnt : kSeedBucketOverflow) << 12;
  out[t] = o;
}

// ---- position lists of the overflowing k-mers (fm_index.cpp: seed_pos_lists) ----
// one thread per k-mer: an interval of more rows than a bucket holds marks its first row (start[]) and brackets itself
// in diff[] (+1 at its first row, -1 behind its last: the running sum of diff is "row i lies in such an interval")
__global__ void __launch_bounds__(256) pos_mark_kernel(const uint32_t* __restrict__ tab, uint32_t k, uint32_t* __restrict__ diff,
                                                       uint32_t* __restrict__ start) {
  const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (c >= (1ull << (2u * k))) return;
  const uint32_t lo = tab[c], hi = tab[c + 1u];
  if (hi - lo <= kSeedBucketRows) return;
  atomicAdd(&diff[lo], 1u);
  atomicAdd(&diff[hi], 0xFFFFFFFFu);
  start[lo] = 1u;
}

// flagged row i -> (list number : position) key at its place among the flagged rows, value = the row
__global__ void __launch_bounds__(256) pos_compact_kernel(const uint64_t* __restrict__ sa, uint32_t n_rows, const uint32_t* __restrict__ flag,
                                                          const uint32_t* __restrict__ idx, const uint32_t* __restrict__ seg, uint32_t pos_bits,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_rows || !flag[i]) return;
  const uint32_t j = idx[i];
  keys[j] = ((uint64_t)seg[i] << pos_bits) | (uint32_t)sa[i];
  vals[j] = i;
}

__global__ void __launch_bounds__(256) pos_rows_kernel(const TabLib b, const uint32_t* __restrict__ vals, uint32_t n_over, uint4* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j < n_over) out[j] = wide_row_dev(b, b.sa[vals[j]]);
}

// header of an overflowing bucket: words 2 / 3 of its row 0 = first row of the k-mer's position list, its length
__global__ void __launch_bounds__(256) pos_header_kernel(const uint32_t* __restrict__ tab, const uint32_t* __restrict__ idx, uint32_t k,
                                                         uint4* __restrict__ buckets) {
  const uint64_t code = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (code >= (1ull << (2u * k))) return;
  const uint32_t lex = lex_code(code, k);
  const uint32_t lo = tab[lex], cnt = tab[lex + 1u] - lo;
  if (cnt <= kSeedBucketRows) return;
  uint4 h = buckets[code * kSeedBucketRows];
  h.z = idx[lo];
  h.w = cnt;
  buckets[code * kSeedBucketRows] = h;
}

}  // namespace

// Position lists (fm_index.hpp) of the k-mers whose bucket overflows, and the headers in their buckets.  *out_rows:
// hipMalloc'ed here (null when no k-mer overflows), *out_n rows of 16 bytes.  Synchronises `stream`.
hipError_t build_seed_pos_lists_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, const uint32_t* tab, uint32_t k,
                                       uint32_t* buckets, uint32_t** out_rows, uint64_t* out_n, hipStream_t stream) {
  const TabLib b{text, sa, n, n + 1u, text_words};
  *out_rows = nullptr;
  *out_n = 0;
  uint32_t *flag = nullptr, *idx = nullptr, *seg = nullptr, *vals0 = nullptr, *vals1 = nullptr;
  uint64_t *keys0 = nullptr, *keys1 = nullptr;
  void *scan_tmp = nullptr, *sort_tmp = nullptr;
  uint32_t* rows = nullptr;
  auto done = [&](hipError_t e) {
    (void)hipFree(flag);
    (void)hipFree(idx);
    (void)hipFree(seg);
    (void)hipFree(vals0);
    (void)hipFree(vals1);
    (void)hipFree(keys0);
    (void)hipFree(keys1);
    (void)hipFree(scan_tmp);
    (void)hipFree(sort_tmp);
    if (e != hipSuccess) (void)hipFree(rows);
    return e;
  };
  hipError_t e;
  const uint32_t m = b.n_rows + 1u;  // (one element behind the last row: where the last interval's bracket closes)
  const uint32_t grid_rows = (b.n_rows + 255u) / 256u, grid_codes = (uint32_t)(((1ull << (2u * k)) + 255u) / 256u);
  if ((e = hipMalloc((void**)&flag, (size_t)m * 4u)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&idx, (size_t)m * 4u)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&seg, (size_t)m * 4u)) != hipSuccess) return done(e);
  if ((e = hipMalloc(&scan_tmp, prims::scan_temp_bytes(m))) != hipSuccess) return done(e);
  if ((e = hipMemsetAsync(flag, 0, (size_t)m * 4u, stream)) != hipSuccess) return done(e);
  if ((e = hipMemsetAsync(seg, 0, (size_t)m * 4u, stream)) != hipSuccess) return done(e);
  hipLaunchKernelGGL(pos_mark_kernel, dim3(grid_codes), dim3(256), 0, stream, tab, k, flag, seg);
  if ((e = hipGetLastError()) != hipSuccess) return done(e);
  if ((e = prims::inclusive_sum_u32(flag, flag, m, scan_tmp, stream)) != hipSuccess) return done(e);  // diff -> "in an overflowing interval"
  if ((e = prims::exclusive_sum_u32(flag, idx, m, scan_tmp, stream)) != hipSuccess) return done(e);   // its place among those rows
  if ((e = prims::inclusive_sum_u32(seg, seg, m, scan_tmp, stream)) != hipSuccess) return done(e);    // number of its list
  uint32_t last[2] = {0, 0};
  if ((e = hipMemcpyAsync(&last[0], idx + b.n_rows, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return done(e);
  if ((e = hipMemcpyAsync(&last[1], seg + b.n_rows, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return done(e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
  const uint32_t n_over = last[0], n_lists = last[1];
  if (!n_over) return done(hipSuccess);
  uint32_t pos_bits = 1, seg_bits = 1;
  while (pos_bits < 32u && (1ull << pos_bits) <= (uint64_t)n) ++pos_bits;
  while (seg_bits < 32u && (1ull << seg_bits) <= (uint64_t)n_lists) ++seg_bits;
  if ((e = hipMalloc((void**)&keys0, (size_t)n_over * 8u)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&keys1, (size_t)n_over * 8u)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&vals0, (size_t)n_over * 4u)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&vals1, (size_t)n_over * 4u)) != hipSuccess) return done(e);
  if ((e = hipMalloc(&sort_tmp, prims::radix_temp_bytes(n_over))) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&rows, (size_t)n_over * 16u)) != hipSuccess) return done(e);
  hipLaunchKernelGGL(pos_compact_kernel, dim3(grid_rows), dim3(256), 0, stream, sa, b.n_rows, flag, idx, seg, pos_bits, keys0, vals0);
  if ((e = hipGetLastError()) != hipSuccess) return done(e);
  bool second = false;
  if ((e = prims::radix_sort_pairs_u64(keys0, keys1, vals0, vals1, n_over, pos_bits + seg_bits, sort_tmp, stream, &second)) != hipSuccess) return done(e);
  hipLaunchKernelGGL(pos_rows_kernel, dim3((n_over + 255u) / 256u), dim3(256), 0, stream, b, second ? vals1 : vals0, n_over, reinterpret_cast<uint4*>(rows));
  if ((e = hipGetLastError()) != hipSuccess) return done(e);
  hipLaunchKernelGGL(pos_header_kernel, dim3(grid_codes), dim3(256), 0, stream, tab, idx, k, reinterpret_cast<uint4*>(buckets));
  if ((e = hipGetLastError()) != hipSuccess) return done(e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
  *out_rows = rows;
  *out_n = n_over;
  return done(hipSuccess);
}

size_t jump_tables_device_temp_bytes(uint32_t n_rows) {
  const size_t tiles = ((size_t)n_rows + kScanTile - 1) / kScanTile;
  return (size_t)n_rows * 4u + ((tiles * 4u + 255u) & ~(size_t)255u);
}

// ks[0..4): the tables' k in storage order (0 = absent), ftab: their 4^k + 1 boundaries back to back (device memory).
hipError_t build_jump_tables_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, const uint8_t ks[4], uint32_t* ftab,
                                    void* tmp, hipStream_t stream) {
  const TabLib b{text, sa, n, n + 1u, text_words};
  uint32_t kmax = 0;
  for (int t = 0; t < 4; ++t) kmax = std::max<uint32_t>(kmax, ks[t]);
  if (!kmax || kmax > 16u) return hipErrorInvalidValue;
  uint32_t* run_max = reinterpret_cast<uint32_t*>(tmp);
  uint32_t* tile_max = run_max + b.n_rows;
  const uint32_t n_tiles = (uint32_t)(((uint64_t)b.n_rows + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL((code_max_kernel<false>), dim3(n_tiles), dim3(kScanThreads), 0, stream, b, kmax, tile_max, run_max);
  TCK(hipGetLastError());
  hipLaunchKernelGGL(tile_max_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, tile_max, n_tiles);
  TCK(hipGetLastError());
  hipLaunchKernelGGL((code_max_kernel<true>), dim3(n_tiles), dim3(kScanThreads), 0, stream, b, kmax, tile_max, run_max);
  TCK(hipGetLastError());
  size_t off = 0;
  for (int t = 0; t < 4; ++t) {
    if (!ks[t]) continue;
    const uint32_t n_codes = 1u << (2u * ks[t]);
    hipLaunchKernelGGL(jump_fill_kernel, dim3((b.n_rows + 255u) / 256u), dim3(256), 0, stream, run_max, b.n_rows, 2u * (kmax - ks[t]), n_codes, ftab + off);
    TCK(hipGetLastError());
    off += (size_t)n_codes + 1u;
  }
  return hipSuccess;
}

hipError_t build_row_context_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, uint32_t* ctx, hipStream_t stream) {
  const TabLib b{text, sa, n, n + 1u, text_words};
  hipLaunchKernelGGL(row_context_kernel, dim3((b.n_rows + 255u) / 256u), dim3(256), 0, stream, b, ctx);
  return hipGetLastError();
}

hipError_t build_wide_rows_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, uint32_t* sa16, hipStream_t stream) {
  const TabLib b{text, sa, n, n + 1u, text_words};
  hipLaunchKernelGGL(wide_rows_kernel, dim3((b.n_rows + 255u) / 256u), dim3(256), 0, stream, b, reinterpret_cast<uint4*>(sa16));
  return hipGetLastError();
}

// tab: the device jump table of k bases (4^k + 1 boundaries); buckets: 4^k x 128 bytes
hipError_t build_seed_buckets_device(const uint32_t* text, uint32_t text_words, const uint64_t* sa, uint32_t n, const uint32_t* tab, uint32_t k,
                                     uint32_t* buckets, hipStream_t stream) {
  const TabLib b{text, sa, n, n + 1u, text_words};
  const uint64_t threads = (1ull << (2u * k)) * kSeedBucketRows;
  hipLaunchKernelGGL(seed_buckets_kernel, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, stream, b, tab, k, reinterpret_cast<uint4*>(buckets));
  return hipGetLastError();
}

}  // namespace mrg
