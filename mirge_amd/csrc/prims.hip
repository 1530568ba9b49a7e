// Device-wide prefix sums and a stable LSD radix sort, written for gfx950 (wave64) -- what collapse.hip, pairs.hip
// and ingest.hip used to call a library for.
//
// Prefix sum: tiles of 4096 elements (256 threads x 16 consecutive elements).  Level by level: tile sums -> the
// same scan over the sums -> tile-local scan + the tile's offset.  Three levels cover 2^36 elements.
//
// Radix sort (least significant digit first, 8 bits per pass, stable): a pass is
//   hist     per tile of 4096 keys the count of every digit, stored digit-major ([digit][tile]) so that ONE prefix
//            sum over the array yields, for (digit d, tile t), the number of keys with a smaller digit anywhere plus
//            the keys with digit d in earlier tiles = where tile t's first d-key goes;
//   scatter  a tile is 4 waves x 16 rows x 64 lanes in memory order.  A key's rank among the tile's keys with its
//            digit = (d-keys of earlier waves) + (d-keys of earlier rows of its wave) + (d-keys of lower lanes of
//            its row).  The lanes of a row that share a digit find each other with eight ballots (one per digit
//            bit); their first lane bumps the wave's LDS counter of that digit, everybody reads the value in front
//            of the bump.  No atomics, no sorting network: order inside a digit is memory order, which is what
//            makes the passes compose.
#include "prims.hpp"

#include <algorithm>

namespace mrg {
namespace prims {

namespace {

constexpr uint32_t kThreads = 256u, kPer = 16u, kTile = kThreads * kPer;

#define PCK(expr)                      \
  do {                                 \
    hipError_t e_ = (expr);            \
    if (e_ != hipSuccess) return e_;   \
  } while (0)

template <class T>
__device__ __forceinline__ T wave_incl_scan_t(T v, uint32_t lane) {
#pragma unroll
  for (uint32_t off = 1; off < 64u; off <<= 1) {
    const T o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  return v;
}

// exclusive prefix of `v` over the workgroup's 256 threads (4 waves); *total = the workgroup's sum
template <class T>
__device__ __forceinline__ T block_excl_scan(T v, T* wave_tot /* LDS, 4 entries */, T* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const T incl = wave_incl_scan_t<T>(v, lane);
  if (lane == 63u) wave_tot[wave] = incl;
  __syncthreads();
  T base = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < kThreads / 64u; ++w) {
    const T t = wave_tot[w];
    base += w < wave ? t : (T)0;
    all += t;
  }
  __syncthreads();
  *total = all;
  return base + incl - v;
}

template <class TIn, class T>
__global__ void __launch_bounds__(kThreads) scan_sums_kernel(const TIn* __restrict__ in, uint64_t n, T* __restrict__ sums) {
  __shared__ T wave_tot[kThreads / 64u];
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kPer;
  T s = 0;
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k)
    if (base + k < n) s += (T)in[base + k];
  T total;
  (void)block_excl_scan<T>(s, wave_tot, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// out[i] = offsets[tile] + (exclusive or inclusive) tile-local prefix; offsets == null: a single tile
template <class TIn, class T, bool INCLUSIVE>
__global__ void __launch_bounds__(kThreads) scan_apply_kernel(const TIn* __restrict__ in, T* __restrict__ out, uint64_t n,
                                                              const T* __restrict__ offsets) {
  __shared__ T wave_tot[kThreads / 64u];
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kPer;
  T v[kPer];
  T s = 0;
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    v[k] = base + k < n ? (T)in[base + k] : (T)0;
    s += v[k];
  }
  T total;
  T run = block_excl_scan<T>(s, wave_tot, &total) + (offsets ? offsets[blockIdx.x] : (T)0);
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    if (INCLUSIVE) run += v[k];
    if (base + k < n) out[base + k] = run;
    if (!INCLUSIVE) run += v[k];
  }
}

template <class T>
size_t scan_temp_elems(uint64_t n) {
  size_t total = 0;
  while (n > kTile) {
    n = (n + kTile - 1) / kTile;
    total += (size_t)((n + 63) & ~63ull);
  }
  return total;
}

template <class TIn, class T, bool INCLUSIVE>
hipError_t scan_impl(const TIn* in, T* out, uint64_t n, T* tmp, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint64_t tiles = (n + kTile - 1) / kTile;
  if (tiles == 1) {
    hipLaunchKernelGGL((scan_apply_kernel<TIn, T, INCLUSIVE>), dim3(1), dim3(kThreads), 0, stream, in, out, n, (const T*)nullptr);
    return hipGetLastError();
  }
  if (tiles > 0x7fffffffull) return hipErrorInvalidValue;
  T* sums = tmp;
  hipLaunchKernelGGL((scan_sums_kernel<TIn, T>), dim3((uint32_t)tiles), dim3(kThreads), 0, stream, in, n, sums);
  PCK(hipGetLastError());
  PCK((scan_impl<T, T, false>(sums, sums, tiles, tmp + ((tiles + 63) & ~63ull), stream)));
  hipLaunchKernelGGL((scan_apply_kernel<TIn, T, INCLUSIVE>), dim3((uint32_t)tiles), dim3(kThreads), 0, stream, in, out, n, (const T*)sums);
  return hipGetLastError();
}

// ---------------------------------------------------------------- radix sort
template <class K>
__global__ void __launch_bounds__(kThreads) radix_hist_kernel(const K* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t n_tiles,
                                                              uint32_t* __restrict__ counts_t) {
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)wave * (kPer * 64u) + lane;
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint64_t i = base + (uint64_t)k * 64u;
    if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  counts_t[(size_t)threadIdx.x * n_tiles + blockIdx.x] = hist[threadIdx.x];
}

template <class K, bool VALS>
__global__ void __launch_bounds__(kThreads) radix_scatter_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                 K* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                                 uint32_t shift, uint32_t n_tiles, const uint32_t* __restrict__ off_t) {
  __shared__ uint32_t wcnt[kThreads / 64u][256];  // d-keys of a wave so far, then: d-keys of the waves in front
  volatile uint32_t* vw = &wcnt[0][0];
  for (uint32_t i = threadIdx.x; i < (kThreads / 64u) * 256u; i += kThreads) wcnt[0][i] = 0u;
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)wave * (kPer * 64u) + lane;
  const uint64_t lt = lane ? (~0ull >> (64u - lane)) : 0ull;
  K key[kPer];
  uint32_t val[kPer];
  uint32_t rank[kPer];  // low 8 bits: the digit; above: rank among the wave's keys with that digit
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint64_t i = base + (uint64_t)k * 64u;
    const bool live = i < n;
    key[k] = live ? keys_in[i] : (K)0;
    val[k] = (VALS && live) ? vals_in[i] : 0u;
    const uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
    uint64_t peers = __ballot(live);
#pragma unroll
    for (uint32_t b = 0; b < 8u; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    uint32_t r = 0;
    if (live) {
      const uint32_t before = vw[wave * 256u + d];
      const uint32_t in_row = (uint32_t)__popcll(peers & lt);
      if (in_row == 0u) vw[wave * 256u + d] = before + (uint32_t)__popcll(peers);
      r = before + in_row;
    }
    rank[k] = (r << 8) | d;
  }
  __syncthreads();
  {
    // per digit: counts of the waves -> exclusive prefix over the waves + the tile's global offset
    const uint32_t d = threadIdx.x;
    uint32_t run = off_t[(size_t)d * n_tiles + blockIdx.x];
#pragma unroll
    for (uint32_t w = 0; w < kThreads / 64u; ++w) {
      const uint32_t c = wcnt[w][d];
      wcnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint64_t i = base + (uint64_t)k * 64u;
    if (i < n) {
      const uint32_t dst = wcnt[wave][rank[k] & 255u] + (rank[k] >> 8);
      keys_out[dst] = key[k];
      if (VALS) vals_out[dst] = val[k];
    }
  }
}

template <class K>
hipError_t radix_impl(K* keys0, K* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, uint32_t bits, void* tmp, hipStream_t stream,
                      bool* result_in_second) {
  *result_in_second = false;
  if (n == 0 || bits == 0) return hipSuccess;
  const uint32_t n_tiles = (n + kTile - 1) / kTile;
  uint32_t* counts = (uint32_t*)tmp;
  const size_t n_counts = (size_t)256 * n_tiles;
  uint32_t* scan_tmp = counts + ((n_counts + 63) & ~(size_t)63);
  K* kin = keys0;
  K* kout = keys1;
  uint32_t* vin = vals0;
  uint32_t* vout = vals1;
  bool second = false;
  for (uint32_t shift = 0; shift < bits; shift += 8u) {
    hipLaunchKernelGGL((radix_hist_kernel<K>), dim3(n_tiles), dim3(kThreads), 0, stream, kin, n, shift, n_tiles, counts);
    PCK(hipGetLastError());
    PCK((scan_impl<uint32_t, uint32_t, false>(counts, counts, n_counts, scan_tmp, stream)));
    if (vals0)
      hipLaunchKernelGGL((radix_scatter_kernel<K, true>), dim3(n_tiles), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, shift, n_tiles,
                         counts);
    else
      hipLaunchKernelGGL((radix_scatter_kernel<K, false>), dim3(n_tiles), dim3(kThreads), 0, stream, kin, (const uint32_t*)nullptr, kout,
                         (uint32_t*)nullptr, n, shift, n_tiles, counts);
    PCK(hipGetLastError());
    std::swap(kin, kout);
    std::swap(vin, vout);
    second = !second;
  }
  *result_in_second = second;
  return hipSuccess;
}

}  // namespace

size_t scan_temp_bytes(uint64_t n) { return (scan_temp_elems<uint64_t>(n) + 64) * sizeof(uint64_t); }

hipError_t exclusive_sum_u32(const uint32_t* in, uint32_t* out, uint64_t n, void* tmp, hipStream_t stream) {
  return scan_impl<uint32_t, uint32_t, false>(in, out, n, (uint32_t*)tmp, stream);
}
hipError_t inclusive_sum_u32(const uint32_t* in, uint32_t* out, uint64_t n, void* tmp, hipStream_t stream) {
  return scan_impl<uint32_t, uint32_t, true>(in, out, n, (uint32_t*)tmp, stream);
}
hipError_t exclusive_sum_u32_to_u64(const uint32_t* in, uint64_t* out, uint64_t n, void* tmp, hipStream_t stream) {
  return scan_impl<uint32_t, uint64_t, false>(in, out, n, (uint64_t*)tmp, stream);
}

size_t radix_temp_bytes(uint64_t n) {
  const uint64_t n_tiles = (n + kTile - 1) / kTile;
  const uint64_t n_counts = 256ull * n_tiles;
  return (size_t)(((n_counts + 63) & ~63ull) * 4u) + scan_temp_bytes(n_counts);
}
hipError_t radix_sort_pairs_u64(uint64_t* keys0, uint64_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, uint32_t bits, void* tmp,
                                hipStream_t stream, bool* result_in_second) {
  return radix_impl<uint64_t>(keys0, keys1, vals0, vals1, n, bits, tmp, stream, result_in_second);
}
hipError_t radix_sort_pairs_u32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, uint32_t bits, void* tmp,
                                hipStream_t stream, bool* result_in_second) {
  return radix_impl<uint32_t>(keys0, keys1, vals0, vals1, n, bits, tmp, stream, result_in_second);
}

}  // namespace prims
}  // namespace mrg
