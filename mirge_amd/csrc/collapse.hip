// Collapse of raw reads into unique sequences with per-sample counts on the GPU.
//
// Reference role: quantReads (utils/quantReads.py:3-24) -- a Python dict keyed by the
// read string, one increment per FASTQ record, plus the read-length histogram
// (readLengthDic).  Here: stable LSD radix sort (hipCUB) of read ids by
// (sample, packed words, N mask, length), head flags on the sorted order, a scan for
// the unique id, and run lengths for the counts.  No atomics on read keys, so a
// sequence that makes up 30 % of a sample (miRNA-seq is that skewed) costs nothing
// extra, and the result is deterministic: uniques come out ordered by (length, bases).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "kernels.hpp"

namespace mrg {

namespace {

constexpr int kT = 256;

__global__ void iota_kernel(uint32_t* idx, uint32_t n) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < n) idx[i] = i;
}

// key[i] = column[idx[i]]  (column of uint64, uint16 or uint8, widened)
template <class T>
__global__ void gather_key_kernel(const T* __restrict__ col, const uint32_t* __restrict__ idx,
                                  uint64_t* __restrict__ key, uint32_t n) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < n) key[i] = (uint64_t)col[idx[i]];
}

// fused key for reads of at most 29 nt in one word: the length in six bits right above the 2 max_len
// bits of the bases (the radix sort then runs over 2 max_len + 6 bits, not 64: seven 8-bit passes
// instead of eight for 22-nt reads)
__global__ void gather_fused_key_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                        const uint32_t* __restrict__ idx, uint64_t* __restrict__ key,
                                        uint32_t n, uint32_t len_shift) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < n) {
    const uint32_t r = idx[i];
    key[i] = words[r] | ((uint64_t)lens[r] << len_shift);
  }
}

// keys-only path (one sample, one word, no N, <= 29 nt): key of read i, in place of the index gather
__global__ void fused_key_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                 uint64_t* __restrict__ key, uint32_t n, uint32_t len_shift) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < n) key[i] = words[i] | ((uint64_t)lens[i] << len_shift);
}

// unique key -> unique read (bases below len_shift, length above)
__global__ void split_key_kernel(const uint64_t* __restrict__ ukey, const uint32_t* __restrict__ n_runs,
                                 uint64_t* __restrict__ u_words, uint8_t* __restrict__ u_lens, uint32_t len_shift) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < *n_runs) {
    const uint64_t k = ukey[i];
    u_words[i] = k & ((1ull << len_shift) - 1ull);
    u_lens[i] = (uint8_t)(k >> len_shift);
  }
}

// keys-only path with several samples: the sample id in the low `sb` bits of the key
__global__ void fused_key_sample_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                        const uint16_t* __restrict__ sample, uint64_t* __restrict__ key, uint32_t n,
                                        uint32_t len_shift, uint32_t sb) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i < n) key[i] = ((words[i] | ((uint64_t)lens[i] << len_shift)) << sb) | (uint64_t)sample[i];
}

// runs of (read, sample): 1 where a new read starts
__global__ void run_heads_kernel(const uint64_t* __restrict__ rkey, const uint32_t* __restrict__ n_runs,
                                 uint32_t* __restrict__ flag, uint32_t n, uint32_t sb) {
  uint32_t j = blockIdx.x * kT + threadIdx.x;
  if (j >= n) return;
  flag[j] = (j < *n_runs && (j == 0 || (rkey[j] >> sb) != (rkey[j - 1] >> sb))) ? 1u : 0u;
}

__global__ void emit_runs_kernel(const uint64_t* __restrict__ rkey, const uint32_t* __restrict__ rcount,
                                 const uint32_t* __restrict__ n_runs, const uint32_t* __restrict__ uid_incl,
                                 uint64_t* __restrict__ u_words, uint8_t* __restrict__ u_lens, uint32_t* __restrict__ quant,
                                 uint32_t n_samples, uint32_t len_shift, uint32_t sb) {
  uint32_t j = blockIdx.x * kT + threadIdx.x;
  if (j >= *n_runs) return;
  const uint64_t k = rkey[j];
  const uint32_t u = uid_incl[j] - 1u;
  quant[(size_t)u * n_samples + (uint32_t)(k & ((1ull << sb) - 1ull))] = rcount[j];
  if (j == 0 || (k >> sb) != (rkey[j - 1] >> sb)) {
    const uint64_t rk = k >> sb;
    u_words[u] = rk & ((1ull << len_shift) - 1ull);
    u_lens[u] = (uint8_t)(rk >> len_shift);
  }
}

struct CollapseCols {
  const uint64_t* words;  // [W][n]
  const uint64_t* nmask;  // [W][n] or null
  const uint8_t* lens;
  const uint16_t* sample;  // null when one sample
  uint32_t W, n;
};

__device__ __forceinline__ bool same_read(const CollapseCols& c, uint32_t a, uint32_t b) {
  if (c.lens[a] != c.lens[b]) return false;
  for (uint32_t w = 0; w < c.W; ++w) {
    if (c.words[(size_t)w * c.n + a] != c.words[(size_t)w * c.n + b]) return false;
    if (c.nmask && c.nmask[(size_t)w * c.n + a] != c.nmask[(size_t)w * c.n + b]) return false;
  }
  return true;
}

// head flags on the sorted order: new unique read / new (read, sample) run
__global__ void head_flags_kernel(CollapseCols c, const uint32_t* __restrict__ idx,
                                  uint32_t* __restrict__ new_read, uint8_t* __restrict__ new_run) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i >= c.n) return;
  bool nr = true, ns = true;
  if (i > 0) {
    const uint32_t a = idx[i], b = idx[i - 1];
    nr = !same_read(c, a, b);
    ns = nr || (c.sample && c.sample[a] != c.sample[b]);
  }
  new_read[i] = nr ? 1u : 0u;
  new_run[i] = ns ? 1u : 0u;
}

// one thread per (read, sample) run: its length is the count
__global__ void emit_kernel(CollapseCols c, const uint32_t* __restrict__ idx,
                            const uint32_t* __restrict__ uid_incl, const uint32_t* __restrict__ run_start,
                            const uint32_t* __restrict__ n_runs_ptr, uint32_t n_samples, uint64_t cap,
                            uint64_t* __restrict__ u_words, uint8_t* __restrict__ u_lens,
                            uint64_t* __restrict__ u_nmask, uint32_t* __restrict__ quant) {
  const uint32_t k = blockIdx.x * kT + threadIdx.x;
  const uint32_t n_runs = *n_runs_ptr;
  if (k >= n_runs) return;
  const uint32_t i = run_start[k];
  const uint32_t end = (k + 1 < n_runs) ? run_start[k + 1] : c.n;
  const uint32_t r = idx[i];
  const uint32_t uid = uid_incl[i] - 1;
  const uint32_t s = c.sample ? c.sample[r] : 0u;
  quant[(size_t)uid * n_samples + s] = end - i;
  // the first run of a unique read also writes the read itself
  if (i == 0 || uid_incl[i - 1] != uid_incl[i]) {
    u_lens[uid] = c.lens[r];
    for (uint32_t w = 0; w < c.W; ++w) {
      u_words[(size_t)w * cap + uid] = c.words[(size_t)w * c.n + r];
      if (u_nmask) u_nmask[(size_t)w * cap + uid] = c.nmask ? c.nmask[(size_t)w * c.n + r] : 0ull;
    }
  }
}

// readLengthDic (QNT:17-21): reads per (length, sample).  Nearly every read has the
// same length, so the bins are privatised in LDS per workgroup (one hot global
// address would serialise at ~11 ns per atomic).
template <bool LDSH>
__global__ void length_hist_kernel(const uint8_t* __restrict__ lens, const uint16_t* __restrict__ sample,
                                   uint32_t n, uint32_t n_samples, unsigned long long* __restrict__ hist) {
  extern __shared__ uint32_t lhist[];
  const uint32_t bins = 256u * n_samples;
  if (LDSH) {
    for (uint32_t b = threadIdx.x; b < bins; b += kT) lhist[b] = 0u;
    __syncthreads();
  }
  for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) {
    const uint32_t b = (uint32_t)lens[i] * n_samples + (sample ? sample[i] : 0u);
    if (LDSH) atomicAdd(&lhist[b], 1u);
    else atomicAdd(&hist[b], 1ull);
  }
  if (LDSH) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += kT)
      if (lhist[b]) atomicAdd(&hist[b], (unsigned long long)lhist[b]);
  }
}

// temporaries come out of the caller's arena (the context keeps one: a hipMalloc + hipFree of a
// few GB per call is milliseconds) and fall back to hipMalloc when it is absent or full
struct Arena {
  char* base = nullptr;
  size_t size = 0, used = 0;
};
static thread_local Arena* g_arena = nullptr;

struct DevBuf {
  void* p = nullptr;
  bool owned = false;
  ~DevBuf() {
    if (p && owned) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) {
    bytes = bytes ? bytes : 16;
    const size_t rounded = (bytes + 255) & ~(size_t)255;
    if (g_arena && g_arena->used + rounded <= g_arena->size) {
      p = g_arena->base + g_arena->used;
      g_arena->used += rounded;
      return hipSuccess;
    }
    owned = true;
    return hipMalloc(&p, bytes);
  }
  template <class T>
  T* as() {
    return reinterpret_cast<T*>(p);
  }
};

#define CK(expr)                       \
  do {                                 \
    hipError_t e_ = (expr);            \
    if (e_ != hipSuccess) return e_;   \
  } while (0)

// longest read and largest sample id of the batch: the fused sort keys put the length right above
// 2 max_len base bits and the sample id below them, so both bounds must HOLD, not be hoped for
__global__ void __launch_bounds__(256) bounds_kernel(const uint8_t* __restrict__ lens, const uint16_t* __restrict__ sample, uint32_t n,
                                                     uint32_t* __restrict__ out) {
  uint32_t ml = 0, ms = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    ml = max(ml, (uint32_t)lens[i]);
    if (sample) ms = max(ms, (uint32_t)sample[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ml = max(ml, (uint32_t)__shfl_down(ml, off, 64));
    ms = max(ms, (uint32_t)__shfl_down(ms, off, 64));
  }
  if ((threadIdx.x & 63u) == 0u) {
    atomicMax(&out[0], ml);
    atomicMax(&out[1], ms);
  }
}

}  // namespace

hipError_t collapse_reads(const uint64_t* d_reads, uint32_t W, const uint8_t* d_lens,
                          const uint64_t* d_nmask, const uint16_t* d_sample, uint32_t n,
                          uint32_t n_samples, uint32_t max_len, uint64_t cap, uint64_t* d_u_words,
                          uint8_t* d_u_lens, uint64_t* d_u_nmask, uint32_t* d_quant,
                          uint64_t* d_len_hist, uint32_t* h_n_unique, hipStream_t stream, void* arena_base,
                          uint64_t arena_bytes) {
  Arena arena;
  arena.base = (char*)arena_base;
  arena.size = arena_base ? (size_t)arena_bytes : 0;
  struct ArenaScope {
    ArenaScope(Arena* a) { g_arena = a->base ? a : nullptr; }
    ~ArenaScope() { g_arena = nullptr; }
  } scope(&arena);
  *h_n_unique = 0;
  CK(hipMemsetAsync(d_len_hist, 0, (size_t)256 * n_samples * 8, stream));
  if (n == 0) return hipStreamSynchronize(stream);
  const uint32_t grid = (n + kT - 1) / kT;
  {
    // the caller's max_len is only a hint: a read longer than it would overlap the length bits of a
    // fused key and merge or split sequences silently.  One pass over the lengths (and sample ids):
    // an out-of-range sample id is an error, an exceeded max_len sends the batch down the general path.
    DevBuf b;
    CK(b.alloc(8));
    CK(hipMemsetAsync(b.p, 0, 8, stream));
    hipLaunchKernelGGL(bounds_kernel, dim3(min(grid, 2048u)), dim3(256), 0, stream, d_lens, n_samples > 1 ? d_sample : nullptr, n, b.as<uint32_t>());
    uint32_t h[2] = {0, 0};
    CK(hipMemcpyAsync(h, b.p, 8, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    if (n_samples > 1 && h[1] >= n_samples) return hipErrorInvalidDevicePointer;  // (mapped to MRG_ERR_ARG by the C-ABI)
    if (max_len && h[0] > max_len) max_len = 0;
  }
  if (W == 1 && max_len > 0 && max_len <= 29 && !d_nmask && !(d_sample && n_samples > 1) && n_samples == 1) {
    // One sample, reads of one word without N: the packed read + its length IS the sort key and the
    // unique read; nothing has to be carried through the sort or gathered afterwards.  Sort the
    // keys alone over their 2 max_len + 6 bits, run-length encode them (unique keys + multiplicities
    // = quant), split the unique keys into words and lengths.
    const uint32_t len_shift = 2u * max_len;
    DevBuf k0, k1, ukey, runs, tmp;
    CK(k0.alloc((size_t)n * 8));
    CK(k1.alloc((size_t)n * 8));
    CK(ukey.alloc((size_t)n * 8));
    CK(runs.alloc(4));
    hipcub::DoubleBuffer<uint64_t> keys(k0.as<uint64_t>(), k1.as<uint64_t>());
    size_t tb_sort = 0, tb_rle = 0;
    CK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb_sort, keys, (int)n, 0, (int)(len_shift + 6u), stream));
    CK(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb_rle, k0.as<uint64_t>(), ukey.as<uint64_t>(), d_quant,
                                             runs.as<uint32_t>(), (int)n, stream));
    const size_t tbytes = tb_sort > tb_rle ? tb_sort : tb_rle;
    CK(tmp.alloc(tbytes));
    hipLaunchKernelGGL(fused_key_kernel, dim3(grid), dim3(kT), 0, stream, d_reads, d_lens, keys.Current(), n, len_shift);
    size_t tb = tbytes;
    CK(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, keys, (int)n, 0, (int)(len_shift + 6u), stream));
    uint32_t n_unique = 0;
    if ((uint64_t)n <= cap) {
      // (every run fits: multiplicities straight into quant)
      tb = tbytes;
      CK(hipcub::DeviceRunLengthEncode::Encode(tmp.p, tb, keys.Current(), ukey.as<uint64_t>(), d_quant, runs.as<uint32_t>(),
                                               (int)n, stream));
      CK(hipMemcpyAsync(&n_unique, runs.as<uint32_t>(), 4, hipMemcpyDeviceToHost, stream));
      hipLaunchKernelGGL(split_key_kernel, dim3(grid), dim3(kT), 0, stream, ukey.as<uint64_t>(), runs.as<uint32_t>(), d_u_words,
                         d_u_lens, len_shift);
      const uint32_t lds = 256u * 4u;
      hipLaunchKernelGGL(length_hist_kernel<true>, dim3(min(grid, 1024u)), dim3(kT), lds, stream, d_lens,
                         (const uint16_t*)nullptr, n, 1u, reinterpret_cast<unsigned long long*>(d_len_hist));
      CK(hipGetLastError());
      CK(hipStreamSynchronize(stream));
      *h_n_unique = n_unique;
      return hipSuccess;
    }
    // (an output capacity below n: the general path below checks it)
  }
  uint32_t sb = 0;
  while ((1u << sb) < n_samples) ++sb;
  if (W == 1 && max_len > 0 && !d_nmask && d_sample && n_samples > 1 && 2u * max_len + 6u + sb <= 64u && (uint64_t)n <= cap) {
    // Several samples: the same, with the sample id in the low bits of the key.  The runs of the
    // sorted keys are (read, sample) pairs with their counts; a read's runs are neighbours.
    const uint32_t len_shift = 2u * max_len;
    DevBuf k0, k1, rkey, rcount, runs, flag, uidb, tmp;
    CK(k0.alloc((size_t)n * 8));
    CK(k1.alloc((size_t)n * 8));
    CK(rkey.alloc((size_t)n * 8));
    CK(rcount.alloc((size_t)n * 4));
    CK(flag.alloc((size_t)n * 4));
    CK(uidb.alloc((size_t)n * 4));
    CK(runs.alloc(4));
    hipcub::DoubleBuffer<uint64_t> keys(k0.as<uint64_t>(), k1.as<uint64_t>());
    const int bits = (int)(len_shift + 6u + sb);
    size_t tb_sort = 0, tb_rle = 0, tb_scan = 0;
    CK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb_sort, keys, (int)n, 0, bits, stream));
    CK(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb_rle, k0.as<uint64_t>(), rkey.as<uint64_t>(), rcount.as<uint32_t>(),
                                             runs.as<uint32_t>(), (int)n, stream));
    CK(hipcub::DeviceScan::InclusiveSum(nullptr, tb_scan, flag.as<uint32_t>(), uidb.as<uint32_t>(), (int)n, stream));
    const size_t tbytes = std::max(tb_sort, std::max(tb_rle, tb_scan));
    CK(tmp.alloc(tbytes));
    hipLaunchKernelGGL(fused_key_sample_kernel, dim3(grid), dim3(kT), 0, stream, d_reads, d_lens, d_sample, keys.Current(), n,
                       len_shift, sb);
    size_t tb = tbytes;
    CK(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, keys, (int)n, 0, bits, stream));
    tb = tbytes;
    CK(hipcub::DeviceRunLengthEncode::Encode(tmp.p, tb, keys.Current(), rkey.as<uint64_t>(), rcount.as<uint32_t>(),
                                             runs.as<uint32_t>(), (int)n, stream));
    uint32_t h_runs = 0;
    CK(hipMemcpyAsync(&h_runs, runs.as<uint32_t>(), 4, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    const uint32_t rgrid = (h_runs + kT - 1) / kT;
    hipLaunchKernelGGL(run_heads_kernel, dim3(rgrid), dim3(kT), 0, stream, rkey.as<uint64_t>(), runs.as<uint32_t>(),
                       flag.as<uint32_t>(), h_runs, sb);
    tb = tbytes;
    CK(hipcub::DeviceScan::InclusiveSum(tmp.p, tb, flag.as<uint32_t>(), uidb.as<uint32_t>(), (int)h_runs, stream));
    uint32_t n_unique = 0;
    CK(hipMemcpyAsync(&n_unique, uidb.as<uint32_t>() + (h_runs - 1), 4, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    CK(hipMemsetAsync(d_quant, 0, (size_t)n_unique * n_samples * 4, stream));
    hipLaunchKernelGGL(emit_runs_kernel, dim3(rgrid), dim3(kT), 0, stream, rkey.as<uint64_t>(), rcount.as<uint32_t>(),
                       runs.as<uint32_t>(), uidb.as<uint32_t>(), d_u_words, d_u_lens, d_quant, n_samples, len_shift, sb);
    const uint32_t lds = 256u * n_samples * 4u;
    auto* hist = reinterpret_cast<unsigned long long*>(d_len_hist);
    if (lds <= 48u * 1024u)
      hipLaunchKernelGGL(length_hist_kernel<true>, dim3(min(grid, 1024u)), dim3(kT), lds, stream, d_lens, d_sample, n, n_samples,
                         hist);
    else
      hipLaunchKernelGGL(length_hist_kernel<false>, dim3(min(grid, 1024u)), dim3(kT), 0, stream, d_lens, d_sample, n, n_samples,
                         hist);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(stream));
    *h_n_unique = n_unique;
    return hipSuccess;
  }
  DevBuf idx0, idx1, key0, key1, flags_read, flags_run, uid, starts, n_runs, temp;
  CK(idx0.alloc((size_t)n * 4));
  CK(idx1.alloc((size_t)n * 4));
  CK(key0.alloc((size_t)n * 8));
  CK(key1.alloc((size_t)n * 8));
  CK(flags_read.alloc((size_t)n * 4));
  CK(flags_run.alloc((size_t)n));
  CK(uid.alloc((size_t)n * 4));
  CK(starts.alloc((size_t)n * 4));
  CK(n_runs.alloc(4));

  hipcub::DoubleBuffer<uint64_t> keys(key0.as<uint64_t>(), key1.as<uint64_t>());
  hipcub::DoubleBuffer<uint32_t> vals(idx0.as<uint32_t>(), idx1.as<uint32_t>());
  size_t temp_bytes = 0, need = 0;
  CK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, vals, (int)n, 0, 64, stream));
  temp_bytes = need;
  CK(hipcub::DeviceScan::InclusiveSum(nullptr, need, flags_read.as<uint32_t>(), uid.as<uint32_t>(), (int)n, stream));
  if (need > temp_bytes) temp_bytes = need;
  CK(hipcub::DeviceSelect::Flagged(nullptr, need, hipcub::CountingInputIterator<uint32_t>(0),
                                   flags_run.as<uint8_t>(), starts.as<uint32_t>(), n_runs.as<uint32_t>(),
                                   (int)n, stream));
  if (need > temp_bytes) temp_bytes = need;
  CK(temp.alloc(temp_bytes));

  hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(kT), 0, stream, vals.Current(), n);

  auto sort_pass = [&](int bits) -> hipError_t {
    size_t tb = temp_bytes;
    return hipcub::DeviceRadixSort::SortPairs(temp.p, tb, keys, vals, (int)n, 0, bits, stream);
  };
  // least significant column first (stable sorts): sample, words, N mask, length
  if (d_sample && n_samples > 1) {
    hipLaunchKernelGGL(gather_key_kernel<uint16_t>, dim3(grid), dim3(kT), 0, stream, d_sample, vals.Current(),
                       keys.Current(), n);
    CK(sort_pass(16));
  }
  const bool fused = (W == 1 && max_len > 0 && max_len <= 29 && !d_nmask);
  if (fused) {
    hipLaunchKernelGGL(gather_fused_key_kernel, dim3(grid), dim3(kT), 0, stream, d_reads, d_lens,
                       vals.Current(), keys.Current(), n, 2u * max_len);
    CK(sort_pass((int)(2u * max_len + 6u)));
  } else {
    for (uint32_t w = 0; w < W; ++w) {
      hipLaunchKernelGGL(gather_key_kernel<uint64_t>, dim3(grid), dim3(kT), 0, stream,
                         d_reads + (size_t)w * n, vals.Current(), keys.Current(), n);
      CK(sort_pass(64));
    }
    if (d_nmask)
      for (uint32_t w = 0; w < W; ++w) {
        hipLaunchKernelGGL(gather_key_kernel<uint64_t>, dim3(grid), dim3(kT), 0, stream,
                           d_nmask + (size_t)w * n, vals.Current(), keys.Current(), n);
        CK(sort_pass(64));
      }
    hipLaunchKernelGGL(gather_key_kernel<uint8_t>, dim3(grid), dim3(kT), 0, stream, d_lens, vals.Current(),
                       keys.Current(), n);
    CK(sort_pass(8));
  }

  CollapseCols c{d_reads, d_nmask, d_lens, (d_sample && n_samples > 1) ? d_sample : nullptr, W, n};
  hipLaunchKernelGGL(head_flags_kernel, dim3(grid), dim3(kT), 0, stream, c, vals.Current(),
                     flags_read.as<uint32_t>(), flags_run.as<uint8_t>());
  {
    size_t tb = temp_bytes;
    CK(hipcub::DeviceScan::InclusiveSum(temp.p, tb, flags_read.as<uint32_t>(), uid.as<uint32_t>(), (int)n, stream));
    tb = temp_bytes;
    CK(hipcub::DeviceSelect::Flagged(temp.p, tb, hipcub::CountingInputIterator<uint32_t>(0),
                                     flags_run.as<uint8_t>(), starts.as<uint32_t>(), n_runs.as<uint32_t>(),
                                     (int)n, stream));
  }
  uint32_t n_unique = 0;
  CK(hipMemcpyAsync(&n_unique, uid.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  if (n_unique > cap) return hipErrorInvalidValue;
  CK(hipMemsetAsync(d_quant, 0, (size_t)n_unique * n_samples * 4, stream));
  hipLaunchKernelGGL(emit_kernel, dim3(grid), dim3(kT), 0, stream, c, vals.Current(), uid.as<uint32_t>(),
                     starts.as<uint32_t>(), n_runs.as<uint32_t>(), n_samples, cap, d_u_words, d_u_lens,
                     d_u_nmask, d_quant);
  {
    const uint16_t* smp = (d_sample && n_samples > 1) ? d_sample : nullptr;
    auto* hist = reinterpret_cast<unsigned long long*>(d_len_hist);
    const uint32_t lds = 256u * n_samples * 4u;
    if (lds <= 48u * 1024u)
      hipLaunchKernelGGL(length_hist_kernel<true>, dim3(min(grid, 1024u)), dim3(kT), lds, stream, d_lens, smp,
                         n, n_samples, hist);
    else
      hipLaunchKernelGGL(length_hist_kernel<false>, dim3(min(grid, 1024u)), dim3(kT), 0, stream, d_lens, smp,
                         n, n_samples, hist);
  }
  CK(hipGetLastError());
  CK(hipStreamSynchronize(stream));
  *h_n_unique = n_unique;
  return hipSuccess;
}


// Exclusive prefix sum of n+1 uint32 values into uint64 (out[n] = total when in[n] == 0):
// offsets of the per-read alignment lists of mrg_list_best.
struct U32ToU64 {
  __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};
hipError_t exclusive_sum_u32_u64(const uint32_t* in, uint64_t* out, uint64_t n_plus_1, hipStream_t stream) {
  hipcub::TransformInputIterator<uint64_t, U32ToU64, const uint32_t*> it(in, U32ToU64());
  size_t need = 0;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, it, out, (int)n_plus_1, stream);
  if (e != hipSuccess) return e;
  void* tmp = nullptr;
  e = hipMalloc(&tmp, need ? need : 16);
  if (e != hipSuccess) return e;
  e = hipcub::DeviceScan::ExclusiveSum(tmp, need, it, out, (int)n_plus_1, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  return e != hipSuccess ? e : e2;
}

}  // namespace mrg
