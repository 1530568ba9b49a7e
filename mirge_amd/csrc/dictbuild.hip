// The exact-match dictionary of a LARGE library (dict_index.hpp), filled on the device from the library's packed text
// and segment tables -- what replaces, for this path, the `.ebwt` load of runAnnotationPipeline.py:643 at the
// start of every run: on the host the 137 Mbp mRNA library's 2^29 slots took 14 s of sixteen threads and an 8.6 GB
// upload; here they never leave the GPU.
//
// Same slot format and the same rules as the host build (dict_index.cpp: insert_position), and the same way of
// making them parallel: a worker owns a range of HOME slots and inserts, in text order, the positions homed there
// whose chain cannot leave the range; the positions homed within a chain's reach of the range's end go in a second
// launch, after every range is filled.  Positions with the same key share a home, so they stay in text order among
// themselves -- all that "the first match is the lowest (entry, offset)" needs, and why the fifteen positions an
// overflowed home still stores are its first.  The layout depends on the range size, what a lookup finds does not.
//   homes    text position -> home slot of its key (positions too close to their segment's end: none)
//   sort     stable radix sort of the positions by home (prims.hip): a range's positions become contiguous, in
//            text order inside a home
//   fill     one thread per range of kRangeSlots homes: lower bound of its first home, then insert after insert
//            (a few dependent loads each, inside the range's own 8 KB of the table)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_util.hpp"
#include "dict_index.hpp"
#include "kernels.hpp"
#include "prims.hpp"

namespace mrg {

namespace {

using namespace dev;

constexpr uint32_t kRangeSlots = 512u;  // homes per worker (a chain reaches kDictChainOverflow - 1 slots further)
constexpr uint32_t kFillThreads = 64u;

#define DCK(expr)                    \
  do {                               \
    hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return e_; \
  } while (0)

struct BuildLib {
  const uint32_t* text;  // 2 bits per base, zero padded (three words past the last base are readable)
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  uint32_t n, key_bases, log2_slots;
};

__device__ __forceinline__ uint32_t segment_of(const BuildLib& b, uint32_t p) {
  uint32_t sg = b.chunk_seg[p >> 5];
  while (b.seg_start[sg + 1] <= p) ++sg;
  return sg;
}

// the 32 bases from text position p on, bases past the end of the text read as A (dict_index.cpp: text_window)
__device__ __forceinline__ uint64_t window_at(const BuildLib& b, uint32_t p) {
  uint64_t win = text_window(b.text, p);
  if ((uint64_t)p + 32u > b.n) win &= low_bits(2u * (b.n - p));
  return win;
}

__device__ __forceinline__ uint32_t home_of_window(const BuildLib& b, uint64_t win) {
  const uint32_t key = b.key_bases >= 16u ? (uint32_t)win : ((uint32_t)win & ((1u << (2u * b.key_bases)) - 1u));
  return (key * kDictHashMul) >> (32u - b.log2_slots);
}

// home of every text position; a position whose key would leave its segment gets `none` (sorts behind every home)
__global__ void __launch_bounds__(256) dict_homes_kernel(const BuildLib b, uint32_t none, uint32_t* __restrict__ homes, uint32_t* __restrict__ pos) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= b.n) return;
  const uint32_t sg = segment_of(b, p);
  const bool valid = (uint64_t)p + b.key_bases <= b.seg_start[sg + 1];
  homes[p] = valid ? home_of_window(b, window_at(b, p)) : none;
  pos[p] = p;
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t v) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < v) lo = mid + 1u;
    else hi = mid;
  }
  return lo;
}

// dict_index.cpp: insert_position, on the device.  Returns 1 when the home's chain overflowed by this insertion.
__device__ __forceinline__ uint32_t insert_position_dev(const BuildLib& b, uint4* __restrict__ slots, uint32_t smask, uint32_t p, uint32_t home,
                                                        uint32_t& n_keys) {
  const uint32_t hw = slots[home].w;
  if (((hw >> kDictChainShift) & kDictChainMask) == kDictChainOverflow) return 0u;  // this home is served by the FM index
  const uint32_t sg = segment_of(b, p);
  const uint32_t s0 = b.seg_start[sg], s1 = b.seg_start[sg + 1];
  const uint64_t win = window_at(b, p);
  const uint32_t after = min(63u, s1 - p), room = min(32u, after);
  uint32_t d = 0;
  bool placed = false;
  for (; d < kDictChainOverflow; ++d) {
    const uint32_t at = (home + d) & smask;
    const uint4 s = slots[at];
    if (!(s.w & kDictOccBit)) {
      const uint32_t keep = s.w & (kDictChainMask << kDictChainShift);  // this slot's own chain field
      slots[at] = make_uint4((uint32_t)win, (uint32_t)(win >> 32), b.seg_ref[sg],
                             keep | after | kDictOccBit | ((b.seg_off[sg] + (p - s0)) << kDictOffShift));
      placed = true;
      break;
    }
    // an earlier position matches whatever this one could
    if (s.x == (uint32_t)win && s.y == (uint32_t)(win >> 32) && min(32u, s.w & kDictAfterMask) >= room) return 0u;
  }
  uint32_t* home_w = &reinterpret_cast<uint32_t*>(slots + home)[3];
  if (!placed) {
    *home_w = (*home_w & ~(kDictChainMask << kDictChainShift)) | (kDictChainOverflow << kDictChainShift);
    return 1u;
  }
  const uint32_t w = *home_w;
  if (d > ((w >> kDictChainShift) & kDictChainMask)) *home_w = (w & ~(kDictChainMask << kDictChainShift)) | (d << kDictChainShift);
  ++n_keys;
  return 0u;
}

// LATE = false: the positions of range r whose chain stays inside it; true: the rest, once every range is filled
template <bool LATE>
__global__ void __launch_bounds__(kFillThreads) dict_fill_kernel(const BuildLib b, const uint32_t* __restrict__ homes, const uint32_t* __restrict__ pos,
                                                                 uint32_t m, uint32_t n_ranges, uint4* __restrict__ slots,
                                                                 unsigned long long* __restrict__ counters) {
  const uint32_t r = blockIdx.x * kFillThreads + threadIdx.x;
  uint32_t n_keys = 0, n_over = 0;
  if (r < n_ranges) {
    const uint32_t smask = (1u << b.log2_slots) - 1u;
    const uint32_t lo = r * kRangeSlots, hi = lo + kRangeSlots, late_from = hi - kDictChainOverflow;
    // (homes are sorted: the range's late positions are the tail of its run)
    uint32_t i = lower_bound_u32(homes, m, LATE ? late_from : lo);
    for (; i < m; ++i) {
      const uint32_t h = homes[i];
      if (h >= (LATE ? hi : late_from)) break;
      n_over += insert_position_dev(b, slots, smask, pos[i], h, n_keys);
    }
  }
  // one atomic per wave and counter
  const uint64_t k = wave_sum((uint64_t)n_keys), o = wave_sum((uint64_t)n_over);
  if ((threadIdx.x & 63u) == 0u) {
    if (k) atomicAdd(&counters[0], (unsigned long long)k);
    if (o) atomicAdd(&counters[1], (unsigned long long)o);
  }
}

}  // namespace

bool exact_dict_device_ok(uint32_t n, uint32_t key_bases, uint32_t log2_slots) {
  return key_bases >= 8u && key_bases <= 16u && log2_slots >= 16u && log2_slots <= 31u && n >= key_bases && n < 0xFFFFFF00u;
}

size_t exact_dict_device_temp_bytes(uint32_t n) {
  return (size_t)n * 4u * 4u + prims::radix_temp_bytes(n) + 256u;
}

// slots: 2^log2_slots x 16 bytes of device memory (zeroed here); counts[0] = positions stored, [1] = homes whose chain
// overflowed.  tmp: exact_dict_device_temp_bytes(n).  Synchronises the stream (the counters come back).
hipError_t build_exact_dict_device(const uint32_t* text, uint32_t n, const uint32_t* seg_start, const uint32_t* seg_ref, const uint32_t* seg_off,
                                   const uint32_t* chunk_seg, uint32_t key_bases, uint32_t log2_slots, void* slots, void* tmp, uint64_t counts[2],
                                   hipStream_t stream) {
  if (!exact_dict_device_ok(n, key_bases, log2_slots)) return hipErrorInvalidValue;
  const BuildLib b{text, seg_start, seg_ref, seg_off, chunk_seg, n, key_bases, log2_slots};
  const uint64_t n_slots = 1ull << log2_slots;
  uint32_t* keys0 = reinterpret_cast<uint32_t*>(tmp);
  uint32_t* keys1 = keys0 + n;
  uint32_t* vals0 = keys1 + n;
  uint32_t* vals1 = vals0 + n;
  unsigned long long* counters = reinterpret_cast<unsigned long long*>(vals1 + n);  // (16-byte aligned: n * 16 bytes in front)
  void* sort_tmp = reinterpret_cast<char*>(counters) + 256;
  DCK(hipMemsetAsync(slots, 0, n_slots * sizeof(DictSlot), stream));
  DCK(hipMemsetAsync(counters, 0, 256, stream));
  // (a position without a key sorts behind every home: one more key bit)
  const uint32_t none = log2_slots >= 31u ? 0xFFFFFFFFu : (1u << log2_slots);
  hipLaunchKernelGGL(dict_homes_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, b, none, keys0, vals0);
  DCK(hipGetLastError());
  bool second = false;
  DCK(prims::radix_sort_pairs_u32(keys0, keys1, vals0, vals1, n, std::min(32u, log2_slots + 1u), sort_tmp, stream, &second));
  const uint32_t* homes = second ? keys1 : keys0;
  const uint32_t* pos = second ? vals1 : vals0;
  const uint32_t n_ranges = (uint32_t)(n_slots / kRangeSlots);
  const uint32_t grid = (n_ranges + kFillThreads - 1u) / kFillThreads;
  hipLaunchKernelGGL((dict_fill_kernel<false>), dim3(grid), dim3(kFillThreads), 0, stream, b, homes, pos, n, n_ranges, reinterpret_cast<uint4*>(slots),
                     counters);
  DCK(hipGetLastError());
  hipLaunchKernelGGL((dict_fill_kernel<true>), dim3(grid), dim3(kFillThreads), 0, stream, b, homes, pos, n, n_ranges, reinterpret_cast<uint4*>(slots),
                     counters);
  DCK(hipGetLastError());
  unsigned long long host_counts[2] = {0, 0};
  DCK(hipMemcpyAsync(host_counts, counters, sizeof(host_counts), hipMemcpyDeviceToHost, stream));
  DCK(hipStreamSynchronize(stream));
  counts[0] = host_counts[0];
  counts[1] = host_counts[1];
  return hipSuccess;
}

}  // namespace mrg
