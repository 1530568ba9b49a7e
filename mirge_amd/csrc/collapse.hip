// Collapse of raw reads into unique sequences with per-sample counts on the GPU (gfx950, wave64).
//
// Reference role: quantReads (utils/quantReads.py:3-24) -- a Python dict keyed by the read string, one increment
// per FASTQ record, plus the read-length histogram (readLengthDic).  Uniques come out ordered by (length, bases),
// deterministically.  Every kernel here is this library's own (rounds 1-4 called a radix-sort library: sorting 100 M
// records to find 18 M distinct ones ignores what small-RNA data is -- a few sequences are most of a sample).
//
// FAST PATH (one-word reads without N, 2 max_len + sample bits <= 58, at most 16 distinct lengths): duplication-aware.
//   bounds + length histogram   one pass over lengths / sample ids (what readLengthDic needs anyway): longest read,
//                  largest sample id, which lengths occur;
//   K0 l1_hist     raw reads per (workgroup chunk, L1 bucket).  An L1 bucket = (length, the 8 most significant bits
//                  of the 2 L-bit packed read): buckets are ORDERED as the output is.  A prefix sum over
//                  [bucket][workgroup] gives every (bucket, workgroup) a private region sized for its raw reads --
//                  an upper bound of what K1 writes there, so K1 needs no global atomic at all;
//   K1 aggregate   a workgroup streams its chunk through an LDS hash table (4096 slots, 64-bit compare-and-swap on
//                  the key, add on the count): the copies of a sequence inside a flush interval become ONE (rest of
//                  key, count) pair of 8 bytes -- the miRNA that is 30 % of a sample leaves a chunk as a handful of
//                  pairs, not as 30 M same-address atomics.  The table is flushed (pairs appended to their L1
//                  regions through LDS cursors) when it is half full or 15 batches old (the count field is 14 bits);
//   K2 subdivide   one workgroup per L1 bucket: the bucket's pairs (at most 1024 segments, one per K1 workgroup) are
//                  counted by their next b2 <= 8 bits and copied into contiguous FINAL buckets of ~1000 pairs;
//   K3 reduce      one workgroup per final bucket: pairs into an LDS table whose slot is the MONOTONE function "top
//                  12 bits of the remaining key", linear probing without wrap-around.  Runs of occupied slots are then
//                  ordered among themselves, so the bucket is sorted once every run is (a few entries each: the head
//                  thread of a run insertion-sorts it) -- no sorting network, no second hash.  Counts of equal keys
//                  were summed by the inserts.  Sorted (key, count) entries go back to the bucket's region;
//   K4 emit        prefix sum of the buckets' read counts, entries -> u_words / u_lens / quant[u][sample].
//   Anything that does not fit (a bucket with more distinct keys than its table, a run of more than 256 slots:
//   sequences that share 20 leading bits of their key by the thousand) raises a flag and the batch takes the general
//   path: slower, never wrong.
//
// GENERAL PATH (several words per read, N masks, reads beyond 29 nt): stable LSD radix sort (prims.hip) of read ids
// by (sample, packed words, N mask, length) column by column, head flags on the sorted order, prefix sums for the
// unique id and the run starts, one thread per (read, sample) run for the counts.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "device_util.hpp"
#include "kernels.hpp"
#include "prims.hpp"

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
                                  uint32_t* __restrict__ new_read, uint32_t* __restrict__ new_run) {
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

// run_start[k] = sorted position of the first record of run k (run_incl = inclusive prefix of the run heads)
__global__ void run_starts_kernel(const uint32_t* __restrict__ run_incl, uint32_t n, uint32_t* __restrict__ run_start) {
  uint32_t i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = run_incl[i];
  if (i == 0 || run_incl[i - 1] != r) run_start[r - 1u] = i;
}

// one thread per (read, sample) run: its length is the count
__global__ void emit_kernel(CollapseCols c, const uint32_t* __restrict__ idx,
                            const uint32_t* __restrict__ uid_incl, const uint32_t* __restrict__ run_start,
                            uint32_t n_runs, uint32_t n_samples, uint64_t cap,
                            uint64_t* __restrict__ u_words, uint8_t* __restrict__ u_lens,
                            uint64_t* __restrict__ u_nmask, uint32_t* __restrict__ quant) {
  const uint32_t k = blockIdx.x * kT + threadIdx.x;
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

// largest sample id of the batch (the longest read comes out of the length histogram)
__global__ void __launch_bounds__(256) max_sample_kernel(const uint16_t* __restrict__ sample, uint32_t n, uint32_t* __restrict__ out) {
  uint32_t ms = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) ms = max(ms, (uint32_t)sample[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ms = max(ms, (uint32_t)__shfl_down(ms, off, 64));
  if ((threadIdx.x & 63u) == 0u) atomicMax(out, ms);
}

// ---------------------------------------------------------------------------------------------------------------
// fast path
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kFastThreads = 1024u;     // K0, K1, K2
constexpr uint32_t kMaxChunks = 1024u;       // K1 workgroups = segments of an L1 bucket
constexpr uint32_t kMaxLenSlots = 16u;       // distinct read lengths a batch may hold
constexpr uint32_t kL1Bits = 8u;
constexpr uint32_t kAggSlots = 4096u;        // K1's LDS table
constexpr uint32_t kAggFlushAt = 2048u;      // ... flushed once it holds this many distinct keys
constexpr uint32_t kAggMaxBatches = 15u;     // ... or after 15 x 1024 reads: a pair's count has 14 bits
constexpr uint32_t kCountShift = 50u;        // pair = count << 50 | rest of the key
constexpr uint64_t kRestMask = (1ull << kCountShift) - 1ull;
constexpr uint32_t kRedThreads = 256u;       // K3, K4
constexpr uint32_t kRedHomeBits = 12u;
constexpr uint32_t kRedSlots = (1u << kRedHomeBits) + 256u;  // no wrap-around: probing may run 256 slots past the last home
constexpr uint32_t kRedMaxRun = 256u;
constexpr uint64_t kEmpty = ~0ull;

struct FastShape {
  uint32_t n, n_chunks, chunk;  // reads, K1 workgroups, reads per workgroup (a multiple of 1024)
  uint32_t n_slots, n_bins;     // lengths present, n_slots x 256
  uint32_t sb;                  // sample bits below the bases in a key
  uint8_t slot_of_len[64];      // length -> slot (0xFF: absent)
  uint8_t len_of_slot[kMaxLenSlots];
};

__device__ __forceinline__ uint32_t l1_shift(uint32_t L) { return 2u * L > kL1Bits ? 2u * L - kL1Bits : 0u; }

// K0: raw reads per (chunk, L1 bucket), stored [bucket][chunk]
__global__ void __launch_bounds__(kFastThreads) l1_hist_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                                               FastShape f, uint32_t* __restrict__ counts_t) {
  extern __shared__ uint32_t hist[];
  for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads) hist[b] = 0u;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * f.chunk;
  const uint64_t hi = min((uint64_t)f.n, lo + f.chunk);
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kFastThreads) {
    const uint32_t L = lens[i];
    const uint32_t bin = (uint32_t)f.slot_of_len[L & 63u] * 256u + (uint32_t)(words[i] >> l1_shift(L));
    atomicAdd(&hist[bin], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads) counts_t[(size_t)b * f.n_chunks + blockIdx.x] = hist[b];
}

// K1: aggregate a chunk through an LDS hash table, append (rest, count) pairs to the chunk's L1 regions
__global__ void __launch_bounds__(kFastThreads) aggregate_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                                                 const uint16_t* __restrict__ sample, FastShape f,
                                                                 const uint32_t* __restrict__ off_t, uint32_t* __restrict__ fill_t,
                                                                 uint64_t* __restrict__ pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* tkey = reinterpret_cast<unsigned long long*>(smem);
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(smem + kAggSlots * 8u);
  uint32_t* cursor = tcnt + kAggSlots;
  uint32_t* ctl = cursor + f.n_bins;  // [0] distinct keys in the table
  for (uint32_t s = threadIdx.x; s < kAggSlots; s += kFastThreads) {
    tkey[s] = kEmpty;
    tcnt[s] = 0u;
  }
  for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads) cursor[b] = off_t[(size_t)b * f.n_chunks + blockIdx.x];
  if (threadIdx.x == 0) ctl[0] = 0u;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * f.chunk;
  const uint64_t hi = min((uint64_t)f.n, lo + f.chunk);
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t batches = 0;
  for (uint64_t base = lo; base < hi; base += kFastThreads) {
    const uint64_t i = base + threadIdx.x;
    bool fresh = false;
    if (i < hi) {
      const uint32_t L = lens[i];
      const uint64_t v = (words[i] << f.sb) | (sample ? (uint64_t)sample[i] : 0ull);
      const unsigned long long key = ((unsigned long long)L << 58) | v;
      uint32_t s = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 52);
      for (;;) {
        const unsigned long long old = atomicCAS(&tkey[s], kEmpty, key);
        if (old == kEmpty || old == key) {
          atomicAdd(&tcnt[s], 1u);
          fresh = old == kEmpty;
          break;
        }
        s = (s + 1u) & (kAggSlots - 1u);
      }
    }
    const uint64_t fm = __ballot(fresh);
    if (lane == 0 && fm) atomicAdd(&ctl[0], (uint32_t)__popcll(fm));
    ++batches;
    __syncthreads();
    const bool last = base + kFastThreads >= hi;
    if (ctl[0] > kAggFlushAt || batches == kAggMaxBatches || last) {
      // flush: every occupied slot becomes a pair in its L1 bucket's region
      for (uint32_t s = threadIdx.x; s < kAggSlots; s += kFastThreads) {
        const unsigned long long key = tkey[s];
        if (key != kEmpty) {
          const uint32_t L = (uint32_t)(key >> 58);
          const uint64_t v = key & ((1ull << 58) - 1ull);
          const uint32_t sh = l1_shift(L) + f.sb;
          const uint32_t bin = (uint32_t)f.slot_of_len[L] * 256u + (uint32_t)(v >> sh);
          const uint32_t dst = atomicAdd(&cursor[bin], 1u);
          pairs[dst] = ((uint64_t)tcnt[s] << kCountShift) | (v & ((1ull << sh) - 1ull));
          tkey[s] = kEmpty;
          tcnt[s] = 0u;
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) ctl[0] = 0u;
      batches = 0;
      __syncthreads();
    }
  }
  for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads)
    fill_t[(size_t)b * f.n_chunks + blockIdx.x] = cursor[b] - off_t[(size_t)b * f.n_chunks + blockIdx.x];
}

// K2: one workgroup per L1 bucket: its pairs (one segment per K1 workgroup) counted by their next b2 bits and copied
// into contiguous final buckets.  fb_* are indexed by final bucket id = L1 bucket x 256 + sub.
__global__ void __launch_bounds__(kFastThreads) subdivide_kernel(FastShape f, const uint32_t* __restrict__ off_t,
                                                                 const uint32_t* __restrict__ fill_t, const uint64_t* __restrict__ pairs_in,
                                                                 uint64_t* __restrict__ pairs_out, uint32_t* __restrict__ fb_start,
                                                                 uint32_t* __restrict__ fb_count, uint8_t* __restrict__ l1_b2,
                                                                 uint32_t* __restrict__ work, uint32_t* __restrict__ n_work) {
  __shared__ uint32_t seg_start[kMaxChunks], seg_fill[kMaxChunks];
  __shared__ uint32_t sub_hist[256], sub_start[256], sub_cursor[256];
  __shared__ uint32_t wtot[kFastThreads / 64u];
  __shared__ uint32_t ctl[4];
  const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t fill = 0;
  if (tid < f.n_chunks) {
    seg_start[tid] = off_t[(size_t)b * f.n_chunks + tid];
    fill = fill_t[(size_t)b * f.n_chunks + tid];
    seg_fill[tid] = fill;
  }
  if (tid < 256u) sub_hist[tid] = 0u, sub_cursor[tid] = 0u;
  // pairs of the bucket
  uint32_t t = fill;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if (lane == 0) wtot[wave] = t;
  __syncthreads();
  uint32_t P = 0;
#pragma unroll
  for (uint32_t w = 0; w < kFastThreads / 64u; ++w) P += wtot[w];
  if (P == 0) return;
  const uint32_t L = f.len_of_slot[b >> 8];
  const uint32_t r1 = l1_shift(L) + f.sb;
  uint32_t b2 = 0;
  while (b2 < 8u && b2 < r1 && ((P + 1023u) >> 10) > (1u << b2)) ++b2;
  const uint32_t sh = r1 - b2;
  for (uint32_t w = wave; w < f.n_chunks; w += kFastThreads / 64u) {
    const uint32_t s0 = seg_start[w], fl = seg_fill[w];
    for (uint32_t j = lane; j < fl; j += 64u) {
      const uint64_t rest = pairs_in[s0 + j] & kRestMask;
      atomicAdd(&sub_hist[(uint32_t)(rest >> sh) & 255u], 1u);
    }
  }
  __syncthreads();
  // final buckets: exclusive prefix of the 256 counts (the first four waves), descriptors, work list
  const uint32_t base = seg_start[0];
  uint32_t cnt = 0, incl = 0;
  if (tid < 256u) {
    cnt = sub_hist[tid];
    incl = dev::wave_incl_scan(cnt);
    if (lane == 63u) wtot[wave] = incl;
  }
  const uint64_t live = __ballot(tid < 256u && cnt != 0u);
  if (tid < 256u && lane == 0) ctl[wave] = (uint32_t)__popcll(live);
  __syncthreads();
  if (tid == 0) {
    const uint32_t n_live = ctl[0] + ctl[1] + ctl[2] + ctl[3];
    const uint32_t at = atomicAdd(n_work, n_live);
    ctl[3] = at + ctl[0] + ctl[1] + ctl[2];
    ctl[2] = at + ctl[0] + ctl[1];
    ctl[1] = at + ctl[0];
    ctl[0] = at;
    l1_b2[b] = (uint8_t)b2;
  }
  __syncthreads();
  if (tid < 256u) {
    uint32_t pre = incl - cnt;
    for (uint32_t w = 0; w < wave; ++w) pre += wtot[w];
    sub_start[tid] = pre;
    fb_start[(size_t)b * 256u + tid] = base + pre;
    fb_count[(size_t)b * 256u + tid] = cnt;
    if (cnt) work[ctl[wave] + (uint32_t)__popcll(live & ((1ull << lane) - 1ull))] = b * 256u + tid;
  }
  __syncthreads();
  for (uint32_t w = wave; w < f.n_chunks; w += kFastThreads / 64u) {
    const uint32_t s0 = seg_start[w], fl = seg_fill[w];
    for (uint32_t j = lane; j < fl; j += 64u) {
      const uint64_t pr = pairs_in[s0 + j];
      const uint32_t sub = (uint32_t)((pr & kRestMask) >> sh) & 255u;
      pairs_out[base + sub_start[sub] + atomicAdd(&sub_cursor[sub], 1u)] = pr;
    }
  }
}

// K3: one final bucket per trip: pairs -> LDS table addressed by a monotone function of the key -> runs sorted ->
// sorted (full key, count) entries back to the bucket's region in `ent_key` / `ent_cnt`
__global__ void __launch_bounds__(kRedThreads) reduce_kernel(FastShape f, const uint32_t* __restrict__ work,
                                                             const uint32_t* __restrict__ n_work_p, const uint32_t* __restrict__ fb_start,
                                                             const uint32_t* __restrict__ fb_count, const uint8_t* __restrict__ l1_b2,
                                                             const uint64_t* __restrict__ pairs, uint64_t* __restrict__ ent_key,
                                                             uint32_t* __restrict__ ent_cnt, uint32_t* __restrict__ fb_entries,
                                                             uint32_t* __restrict__ fb_reads, uint32_t* __restrict__ overflow) {
  __shared__ unsigned long long tkey[kRedSlots];
  __shared__ uint32_t tcnt[kRedSlots];
  __shared__ uint32_t wtot[2][kRedThreads / 64u];
  constexpr uint32_t kPerThread = kRedSlots / kRedThreads;  // 17 consecutive slots per thread
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_work = *n_work_p;
  for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const uint32_t fb = work[wi];
    const uint32_t l1 = fb >> 8, sub = fb & 255u;
    const uint32_t L = f.len_of_slot[l1 >> 8];
    const uint32_t r1 = l1_shift(L) + f.sb, b2 = l1_b2[l1], r2 = r1 - b2;
    const uint32_t start = fb_start[fb], P = fb_count[fb];
    for (uint32_t s = tid; s < kRedSlots; s += kRedThreads) {
      tkey[s] = kEmpty;
      tcnt[s] = 0u;
    }
    __syncthreads();
    bool lost = false;
    for (uint32_t j = tid; j < P; j += kRedThreads) {
      const uint64_t pr = pairs[start + j];
      const unsigned long long rest = pr & kRestMask;
      const uint64_t rest2 = rest & ((1ull << r2) - 1ull);
      uint32_t s = r2 >= kRedHomeBits ? (uint32_t)(rest2 >> (r2 - kRedHomeBits)) : (uint32_t)(rest2 << (kRedHomeBits - r2));
      for (;;) {
        if (s >= kRedSlots) {
          lost = true;
          break;
        }
        const unsigned long long old = atomicCAS(&tkey[s], kEmpty, rest);
        if (old == kEmpty || old == rest) {
          atomicAdd(&tcnt[s], (uint32_t)(pr >> kCountShift));
          break;
        }
        ++s;
      }
    }
    __syncthreads();
    // runs of occupied slots are ordered among themselves (monotone home, no wrap): sort every run in place
    const uint32_t s_lo = tid * kPerThread;
    for (uint32_t s = s_lo; s < s_lo + kPerThread; ++s) {
      if (tkey[s] == kEmpty || (s > 0 && tkey[s - 1] != kEmpty)) continue;
      uint32_t e = s + 1;
      while (e < kRedSlots && tkey[e] != kEmpty) ++e;
      if (e - s > kRedMaxRun) {
        lost = true;
        continue;
      }
      for (uint32_t i = s + 1; i < e; ++i) {
        const unsigned long long k = tkey[i];
        const uint32_t c = tcnt[i];
        uint32_t j = i;
        while (j > s && tkey[j - 1] > k) {
          tkey[j] = tkey[j - 1];
          tcnt[j] = tcnt[j - 1];
          --j;
        }
        tkey[j] = k;
        tcnt[j] = c;
      }
    }
    if (lost) atomicOr(overflow, 1u);
    __syncthreads();
    // entries and reads (an entry whose bases differ from the entry in front of it) of the thread's slots
    uint32_t n_ent = 0, n_rd = 0;
    for (uint32_t s = s_lo; s < s_lo + kPerThread; ++s) {
      const unsigned long long k = tkey[s];
      if (k == kEmpty) continue;
      ++n_ent;
      n_rd += (s == 0 || tkey[s - 1] == kEmpty || (tkey[s - 1] >> f.sb) != (k >> f.sb)) ? 1u : 0u;
    }
    const uint32_t ie = dev::wave_incl_scan(n_ent), ir = dev::wave_incl_scan(n_rd);
    if (lane == 63u) {
      wtot[0][wave] = ie;
      wtot[1][wave] = ir;
    }
    __syncthreads();
    uint32_t pos = ie - n_ent, tot_e = 0, tot_r = 0;
#pragma unroll
    for (uint32_t w = 0; w < kRedThreads / 64u; ++w) {
      pos += w < wave ? wtot[0][w] : 0u;
      tot_e += wtot[0][w];
      tot_r += wtot[1][w];
    }
    const uint64_t prefix = ((uint64_t)(l1 & 255u) << r1) | ((uint64_t)sub << r2);  // the key bits the bucket stands for
    for (uint32_t s = s_lo; s < s_lo + kPerThread; ++s) {
      const unsigned long long k = tkey[s];
      if (k == kEmpty) continue;
      ent_key[start + pos] = prefix | (k & ((1ull << r2) - 1ull));
      ent_cnt[start + pos] = tcnt[s];
      ++pos;
    }
    if (tid == 0) {
      fb_entries[fb] = tot_e;
      fb_reads[fb] = tot_r;
    }
    __syncthreads();
  }
}

// K4: entries of a final bucket -> the output arrays at the bucket's first read (read_base = exclusive prefix of fb_reads)
__global__ void __launch_bounds__(kRedThreads) emit_fast_kernel(FastShape f, const uint32_t* __restrict__ work,
                                                                const uint32_t* __restrict__ n_work_p, const uint32_t* __restrict__ fb_start,
                                                                const uint32_t* __restrict__ fb_entries, const uint32_t* __restrict__ read_base,
                                                                const uint64_t* __restrict__ ent_key, const uint32_t* __restrict__ ent_cnt,
                                                                uint32_t n_samples, uint64_t* __restrict__ u_words, uint8_t* __restrict__ u_lens,
                                                                uint32_t* __restrict__ quant) {
  __shared__ uint32_t wtot[kRedThreads / 64u];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_work = *n_work_p;
  const uint64_t smask = (1ull << f.sb) - 1ull;
  for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const uint32_t fb = work[wi];
    const uint32_t L = f.len_of_slot[fb >> 16];
    const uint32_t start = fb_start[fb], E = fb_entries[fb];
    uint32_t u0 = read_base[fb];  // index of the first read of this trip
    for (uint32_t j0 = 0; j0 < E; j0 += kRedThreads) {
      const uint32_t j = j0 + tid;
      uint64_t v = 0;
      bool head = false;
      if (j < E) {
        v = ent_key[start + j];
        head = j == 0 || (ent_key[start + j - 1] >> f.sb) != (v >> f.sb);
      }
      const uint32_t h = head ? 1u : 0u;
      const uint32_t incl = dev::wave_incl_scan(h);
      if (lane == 63u) wtot[wave] = incl;
      __syncthreads();
      uint32_t idx = incl, tot = 0;
#pragma unroll
      for (uint32_t w = 0; w < kRedThreads / 64u; ++w) {
        idx += w < wave ? wtot[w] : 0u;
        tot += wtot[w];
      }
      if (j < E) {
        const uint32_t u = u0 + idx - 1u;
        if (head) {
          u_words[u] = v >> f.sb;
          u_lens[u] = (uint8_t)L;
        }
        quant[(size_t)u * n_samples + (uint32_t)(v & smask)] = ent_cnt[start + j];
      }
      u0 += tot;
      __syncthreads();
    }
  }
}

// The fast path.  *took = false: the batch does not fit it (or overflowed a table): nothing was written that the
// general path does not overwrite.  h_len_hist: the length histogram of the batch ([256][S], already on the host).
hipError_t collapse_fast(const uint64_t* d_reads, const uint8_t* d_lens, const uint16_t* d_sample, uint32_t n, uint32_t n_samples,
                         const std::vector<uint64_t>& h_len_hist, uint64_t cap, uint64_t* d_u_words, uint8_t* d_u_lens, uint32_t* d_quant,
                         uint32_t* h_n_unique, int n_cu, hipStream_t stream, bool* took) {
  *took = false;
  FastShape f;
  std::fill(f.slot_of_len, f.slot_of_len + 64, (uint8_t)0xFF);
  f.n = n;
  f.sb = 0;
  while ((1u << f.sb) < n_samples) ++f.sb;
  f.n_slots = 0;
  uint32_t max_len = 0;
  for (uint32_t L = 0; L < 256u; ++L) {
    uint64_t c = 0;
    for (uint32_t s = 0; s < n_samples; ++s) c += h_len_hist[(size_t)L * n_samples + s];
    if (!c) continue;
    if (L > 29u || f.n_slots == kMaxLenSlots) return hipSuccess;
    f.slot_of_len[L] = (uint8_t)f.n_slots;
    f.len_of_slot[f.n_slots++] = (uint8_t)L;
    max_len = L;
  }
  if (2u * max_len + f.sb > 58u || f.n_slots == 0) return hipSuccess;
  f.n_bins = f.n_slots * 256u;
  f.n_chunks = std::max<uint32_t>(1u, std::min<uint32_t>(kMaxChunks, (n + 8191u) / 8192u));
  f.chunk = (uint32_t)((((uint64_t)n + f.n_chunks - 1) / f.n_chunks + 1023u) & ~1023ull);
  const size_t n_ct = (size_t)f.n_bins * f.n_chunks, n_fb = (size_t)f.n_bins * 256u;
  DevBuf off_t, fill_t, bufA, bufB, cnt, fbs, misc, work, stmp;
  CK(off_t.alloc(n_ct * 4));
  CK(fill_t.alloc(n_ct * 4));
  CK(bufA.alloc((size_t)n * 8));
  CK(bufB.alloc((size_t)n * 8));
  CK(cnt.alloc((size_t)n * 4));
  CK(fbs.alloc(n_fb * 4 * 5));  // start, count, entries, reads, read_base
  CK(misc.alloc(f.n_bins + 64));  // l1_b2 bytes, then (64-byte aligned) n_work, overflow
  CK(work.alloc(n_fb * 4));
  CK(stmp.alloc(std::max(prims::scan_temp_bytes(n_ct), prims::scan_temp_bytes(n_fb))));
  uint32_t* fb_start = fbs.as<uint32_t>();
  uint32_t* fb_count = fb_start + n_fb;
  uint32_t* fb_entries = fb_count + n_fb;
  uint32_t* fb_reads = fb_entries + n_fb;
  uint32_t* read_base = fb_reads + n_fb;
  uint8_t* l1_b2 = misc.as<uint8_t>();
  uint32_t* n_work = reinterpret_cast<uint32_t*>(misc.as<uint8_t>() + ((f.n_bins + 15u) & ~15u));
  uint32_t* overflow = n_work + 1;
  CK(hipMemsetAsync(fb_count, 0, n_fb * 4 * 3, stream));  // count, entries, reads
  CK(hipMemsetAsync(misc.p, 0, f.n_bins + 64, stream));
  const uint16_t* smp = n_samples > 1 ? d_sample : nullptr;
  hipLaunchKernelGGL(l1_hist_kernel, dim3(f.n_chunks), dim3(kFastThreads), f.n_bins * 4u, stream, d_reads, d_lens, f, off_t.as<uint32_t>());
  CK(hipGetLastError());
  CK(prims::exclusive_sum_u32(off_t.as<uint32_t>(), off_t.as<uint32_t>(), n_ct, stmp.p, stream));
  const uint32_t agg_lds = kAggSlots * 12u + f.n_bins * 4u + 16u;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggregate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)agg_lds));
  hipLaunchKernelGGL(aggregate_kernel, dim3(f.n_chunks), dim3(kFastThreads), agg_lds, stream, d_reads, d_lens, smp, f, off_t.as<uint32_t>(),
                     fill_t.as<uint32_t>(), bufA.as<uint64_t>());
  CK(hipGetLastError());
  hipLaunchKernelGGL(subdivide_kernel, dim3(f.n_bins), dim3(kFastThreads), 0, stream, f, off_t.as<uint32_t>(), fill_t.as<uint32_t>(),
                     bufA.as<uint64_t>(), bufB.as<uint64_t>(), fb_start, fb_count, l1_b2, work.as<uint32_t>(), n_work);
  CK(hipGetLastError());
  const uint32_t red_grid = (uint32_t)std::max(1, n_cu) * 6u;
  hipLaunchKernelGGL(reduce_kernel, dim3(red_grid), dim3(kRedThreads), 0, stream, f, work.as<uint32_t>(), n_work, fb_start, fb_count, l1_b2,
                     bufB.as<uint64_t>(), bufA.as<uint64_t>(), cnt.as<uint32_t>(), fb_entries, fb_reads, overflow);
  CK(hipGetLastError());
  CK(prims::exclusive_sum_u32(fb_reads, read_base, n_fb, stmp.p, stream));
  uint32_t h_over = 0, h_last[2] = {0, 0};
  CK(hipMemcpyAsync(&h_over, overflow, 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&h_last[0], read_base + (n_fb - 1), 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&h_last[1], fb_reads + (n_fb - 1), 4, hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  if (h_over) return hipSuccess;  // (general path)
  const uint64_t n_unique = (uint64_t)h_last[0] + h_last[1];
  if (n_unique > cap) return hipErrorInvalidValue;
  if (n_samples > 1) CK(hipMemsetAsync(d_quant, 0, (size_t)n_unique * n_samples * 4, stream));
  hipLaunchKernelGGL(emit_fast_kernel, dim3(red_grid), dim3(kRedThreads), 0, stream, f, work.as<uint32_t>(), n_work, fb_start, fb_entries,
                     read_base, bufA.as<uint64_t>(), cnt.as<uint32_t>(), n_samples, d_u_words, d_u_lens, d_quant);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(stream));
  *h_n_unique = (uint32_t)n_unique;
  *took = true;
  return hipSuccess;
}

}  // namespace

hipError_t collapse_reads(const uint64_t* d_reads, uint32_t W, const uint8_t* d_lens,
                          const uint64_t* d_nmask, const uint16_t* d_sample, uint32_t n,
                          uint32_t n_samples, uint32_t max_len, uint64_t cap, uint64_t* d_u_words,
                          uint8_t* d_u_lens, uint64_t* d_u_nmask, uint32_t* d_quant,
                          uint64_t* d_len_hist, uint32_t* h_n_unique, hipStream_t stream, void* arena_base,
                          uint64_t arena_bytes, int n_cu, bool allow_fast) {
  Arena arena;
  arena.base = (char*)arena_base;
  arena.size = arena_base ? (size_t)arena_bytes : 0;
  struct ArenaScope {
    ArenaScope(Arena* a) { g_arena = a->base ? a : nullptr; }
    ~ArenaScope() { g_arena = nullptr; }
  } scope(&arena);
  *h_n_unique = 0;
  (void)max_len;  // (a hint of rounds 1-4: the length histogram below says what the batch holds)
  CK(hipMemsetAsync(d_len_hist, 0, (size_t)256 * n_samples * 8, stream));
  if (n == 0) return hipStreamSynchronize(stream);
  const uint32_t grid = (n + kT - 1) / kT;
  const uint16_t* smp = (d_sample && n_samples > 1) ? d_sample : nullptr;
  if (n_cu <= 0) n_cu = 256;
  // ---- readLengthDic + the largest sample id: what the batch holds decides the path; an out-of-range sample id is
  // an error (it would index past a row of quant) ----
  std::vector<uint64_t> h_hist((size_t)256 * n_samples);
  if (smp) {
    // (first, and on its own: the histogram below indexes its bins with the sample id)
    uint32_t h_ms = 0;
    DevBuf b;
    CK(b.alloc(4));
    CK(hipMemsetAsync(b.p, 0, 4, stream));
    hipLaunchKernelGGL(max_sample_kernel, dim3(min(grid, 2048u)), dim3(256), 0, stream, smp, n, b.as<uint32_t>());
    CK(hipMemcpyAsync(&h_ms, b.p, 4, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    if (h_ms >= n_samples) return hipErrorInvalidDevicePointer;  // (mapped to MRG_ERR_ARG by the C-ABI)
  }
  {
    auto* hist = reinterpret_cast<unsigned long long*>(d_len_hist);
    const uint32_t lds = 256u * n_samples * 4u;
    if (lds <= 48u * 1024u)
      hipLaunchKernelGGL(length_hist_kernel<true>, dim3(min(grid, 1024u)), dim3(kT), lds, stream, d_lens, smp, n, n_samples, hist);
    else
      hipLaunchKernelGGL(length_hist_kernel<false>, dim3(min(grid, 1024u)), dim3(kT), 0, stream, d_lens, smp, n, n_samples, hist);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(h_hist.data(), d_len_hist, h_hist.size() * 8, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
  }
  if (W == 1 && !d_nmask && allow_fast) {
    bool took = false;
    const size_t mark = arena.used;
    CK(collapse_fast(d_reads, d_lens, smp, n, n_samples, h_hist, cap, d_u_words, d_u_lens, d_quant, h_n_unique, n_cu, stream, &took));
    if (took) return hipSuccess;
    arena.used = mark;
  }

  // ---- general path ----
  DevBuf idx0, idx1, key0, key1, flags_read, flags_run, starts, temp;
  CK(idx0.alloc((size_t)n * 4));
  CK(idx1.alloc((size_t)n * 4));
  CK(key0.alloc((size_t)n * 8));
  CK(key1.alloc((size_t)n * 8));
  CK(flags_read.alloc((size_t)n * 4));
  CK(flags_run.alloc((size_t)n * 4));
  CK(starts.alloc((size_t)n * 4));
  CK(temp.alloc(std::max(prims::radix_temp_bytes(n), prims::scan_temp_bytes(n))));
  uint64_t* keys[2] = {key0.as<uint64_t>(), key1.as<uint64_t>()};
  uint32_t* vals[2] = {idx0.as<uint32_t>(), idx1.as<uint32_t>()};
  int cur = 0;  // which pair of buffers holds the current order
  hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(kT), 0, stream, vals[cur], n);
  // (the key column of a pass is gathered into keys[cur] through vals[cur])
  auto sort_pass = [&](uint32_t bits) -> hipError_t {
    bool second = false;
    hipError_t e = prims::radix_sort_pairs_u64(keys[cur], keys[cur ^ 1], vals[cur], vals[cur ^ 1], n, bits, temp.p, stream, &second);
    if (second) cur ^= 1;
    return e;
  };
  // least significant column first (stable sorts): sample, words, N mask, length
  if (smp) {
    hipLaunchKernelGGL(gather_key_kernel<uint16_t>, dim3(grid), dim3(kT), 0, stream, smp, vals[cur], keys[cur], n);
    uint32_t sbits = 1;
    while ((1u << sbits) < n_samples) ++sbits;
    CK(sort_pass(sbits));
  }
  for (uint32_t w = 0; w < W; ++w) {
    hipLaunchKernelGGL(gather_key_kernel<uint64_t>, dim3(grid), dim3(kT), 0, stream, d_reads + (size_t)w * n, vals[cur], keys[cur], n);
    CK(sort_pass(64));
  }
  if (d_nmask)
    for (uint32_t w = 0; w < W; ++w) {
      hipLaunchKernelGGL(gather_key_kernel<uint64_t>, dim3(grid), dim3(kT), 0, stream, d_nmask + (size_t)w * n, vals[cur], keys[cur], n);
      CK(sort_pass(64));
    }
  hipLaunchKernelGGL(gather_key_kernel<uint8_t>, dim3(grid), dim3(kT), 0, stream, d_lens, vals[cur], keys[cur], n);
  CK(sort_pass(8));

  CollapseCols c{d_reads, d_nmask, d_lens, smp, W, n};
  uint32_t* uid = flags_read.as<uint32_t>();   // head flags, then their inclusive prefix = unique id + 1
  uint32_t* runs = flags_run.as<uint32_t>();   // run heads, then their inclusive prefix = run number + 1
  hipLaunchKernelGGL(head_flags_kernel, dim3(grid), dim3(kT), 0, stream, c, vals[cur], uid, runs);
  CK(hipGetLastError());
  CK(prims::inclusive_sum_u32(uid, uid, n, temp.p, stream));
  CK(prims::inclusive_sum_u32(runs, runs, n, temp.p, stream));
  uint32_t n_unique = 0, n_runs = 0;
  CK(hipMemcpyAsync(&n_unique, uid + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&n_runs, runs + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  if (n_unique > cap) return hipErrorInvalidValue;
  hipLaunchKernelGGL(run_starts_kernel, dim3(grid), dim3(kT), 0, stream, runs, n, starts.as<uint32_t>());
  CK(hipMemsetAsync(d_quant, 0, (size_t)n_unique * n_samples * 4, stream));
  hipLaunchKernelGGL(emit_kernel, dim3((n_runs + kT - 1) / kT), dim3(kT), 0, stream, c, vals[cur], uid, starts.as<uint32_t>(), n_runs,
                     n_samples, cap, d_u_words, d_u_lens, d_u_nmask, d_quant);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(stream));
  *h_n_unique = n_unique;
  return hipSuccess;
}

// Exclusive prefix sum of n+1 uint32 values into uint64 (out[n] = total when in[n] == 0):
// offsets of the per-read alignment lists of mrg_list_best.
hipError_t exclusive_sum_u32_u64(const uint32_t* in, uint64_t* out, uint64_t n_plus_1, hipStream_t stream) {
  void* tmp = nullptr;
  hipError_t e = hipMalloc(&tmp, prims::scan_temp_bytes(n_plus_1));
  if (e != hipSuccess) return e;
  e = prims::exclusive_sum_u32_to_u64(in, out, n_plus_1, tmp, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  return e != hipSuccess ? e : e2;
}

}  // namespace mrg
