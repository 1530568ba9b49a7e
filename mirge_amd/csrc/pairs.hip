// Pair tables of a LARGE library, built on the device from the arrays already resident there
// (suffix-array rows + packed text): see fm_index.hpp (PairTables) for what they are and
// kernels.hip (fused_kernel<W, true>) for how they are searched.
//
// A one-mismatch policy splits the seed region into two pigeonhole pieces.  For reads of 22 nt and
// more those are >= 11 bases and the library's k = 11 jump table answers each with ~3 rows of an
// 11 Mbp library; for a 16..19-base seed region they are 8..9 bases and match 40..170 rows each.
// Three anchors of A = 5 bases (offsets 0, 5, 10) give three 10-base pair keys -- one mismatch
// leaves at least one pair clean -- and ~10 rows per lookup, whatever the read length.  The tables
// (4^10 + 1 boundaries per gap, 8-byte rows sorted by key) are built lazily, the first time a
// cascade needs them, with one stable radix sort per gap (deterministic layout: rows of one key stay
// in suffix-array order).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "kernels.hpp"
#include "prims.hpp"

namespace mrg {

namespace {

__device__ __forceinline__ uint64_t text_window(const uint32_t* __restrict__ text, uint32_t p) {
  const uint32_t i = p >> 4, sh = (p & 15) * 2;
  const uint32_t w0 = text[i], w1 = text[i + 1], w2 = text[i + 2];
  const uint64_t lo64 = (uint64_t)w0 | ((uint64_t)w1 << 32);
  return (lo64 >> sh) | ((((uint64_t)w2) << 1) << (63 - sh));
}

// key of every suffix-array row for gap d (0xFFFFFFFF = the two anchors do not fit the row's segment)
__global__ void pair_keys_kernel(const uint64_t* __restrict__ sa, const uint32_t* __restrict__ text, uint32_t n_rows,
                                 uint32_t anchor, uint32_t d, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const uint64_t row = sa[i];
  const uint32_t p = (uint32_t)row, after = (uint32_t)(row >> 40) & 255u;
  uint32_t key = 0xFFFFFFFFu;
  if (after >= d + anchor) {
    const uint32_t kb = 2u * anchor;
    const uint64_t amask = (1ull << kb) - 1ull;
    key = (uint32_t)((text_window(text, p) & amask) | ((text_window(text, p + d) & amask) << kb));
  }
  keys[i] = key;
  vals[i] = i;
}

__global__ void pair_hist_kernel(const uint32_t* __restrict__ keys_sorted, uint32_t n_rows, uint32_t* __restrict__ jump) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const uint32_t k = keys_sorted[i];
  if (k != 0xFFFFFFFFu) atomicAdd(&jump[k + 1u], 1u);
}

__global__ void pair_rows_kernel(const uint64_t* __restrict__ sa, const uint32_t* __restrict__ vals_sorted, uint32_t n_valid,
                                 uint64_t* __restrict__ rows) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_valid) rows[i] = sa[vals_sorted[i]];
}

}  // namespace

// jump: n_gaps tables of 4^(2 anchor) + 1 words; rows: capacity n_gaps * n_rows;
// row_off[t] = first row of list t (host array of n_gaps + 1 entries).
hipError_t build_pair_tables_device(const uint64_t* sa, const uint32_t* text, uint32_t n_rows, uint32_t anchor,
                                    uint32_t n_gaps, uint32_t* jump, uint64_t* rows, uint32_t* row_off,
                                    hipStream_t stream) {
  const uint64_t n_codes = 1ull << (4u * anchor);
  uint32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
  void* tmp = nullptr;
  hipError_t e = hipSuccess;
  auto done = [&](hipError_t rc) {
    (void)hipStreamSynchronize(stream);
    (void)hipFree(keys);
    (void)hipFree(vals);
    (void)hipFree(keys2);
    (void)hipFree(vals2);
    (void)hipFree(tmp);
    return rc;
  };
  if ((e = hipMalloc((void**)&keys, (size_t)n_rows * 4)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&vals, (size_t)n_rows * 4)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&keys2, (size_t)n_rows * 4)) != hipSuccess) return done(e);
  if ((e = hipMalloc((void**)&vals2, (size_t)n_rows * 4)) != hipSuccess) return done(e);
  const size_t need = std::max(prims::radix_temp_bytes(n_rows), prims::scan_temp_bytes(n_codes + 1));
  if ((e = hipMalloc(&tmp, need ? need : 16)) != hipSuccess) return done(e);
  const uint32_t block = 256, grid = (n_rows + block - 1) / block;
  uint32_t off = 0;
  for (uint32_t t = 0; t < n_gaps; ++t) {
    uint32_t* jt = jump + (size_t)t * (n_codes + 1);
    if ((e = hipMemsetAsync(jt, 0, (size_t)(n_codes + 1) * 4, stream)) != hipSuccess) return done(e);
    hipLaunchKernelGGL(pair_keys_kernel, dim3(grid), dim3(block), 0, stream, sa, text, n_rows, anchor, (t + 1u) * anchor, keys, vals);
    // (stable: rows of one key stay in suffix-array order; four passes of eight bits end in the buffers they began in)
    bool second = false;
    if ((e = prims::radix_sort_pairs_u32(keys, keys2, vals, vals2, n_rows, 32, tmp, stream, &second)) != hipSuccess) return done(e);
    const uint32_t* keys_sorted = second ? keys2 : keys;
    const uint32_t* vals_sorted = second ? vals2 : vals;
    hipLaunchKernelGGL(pair_hist_kernel, dim3(grid), dim3(block), 0, stream, keys_sorted, n_rows, jt);
    if ((e = prims::inclusive_sum_u32(jt, jt, n_codes + 1, tmp, stream)) != hipSuccess) return done(e);
    uint32_t n_valid = 0;
    if ((e = hipMemcpyAsync(&n_valid, jt + n_codes, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    row_off[t] = off;
    if (n_valid) hipLaunchKernelGGL(pair_rows_kernel, dim3((n_valid + block - 1) / block), dim3(block), 0, stream, sa, vals_sorted, n_valid, rows + off);
    off += n_valid;
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
  }
  row_off[n_gaps] = off;
  return done(hipSuccess);
}

}  // namespace mrg
