// Collapse of raw reads into unique sequences with per-sample counts on the GPU (gfx950, wave64).
//
// Reference role: quantReads (utils/quantReads.py:3-24) -- a Python dict keyed by the read string, one increment
// per FASTQ record, plus the read-length histogram (readLengthDic).  Uniques come out ordered by (length, bases),
// deterministically.  Every kernel here is this library's own (rounds 1-4 called a radix-sort library: sorting 100 M
// records to find 18 M distinct ones ignores what small-RNA data is -- a few sequences are most of a sample).
//
// FAST PATH (one-word reads without N of at most 29 nt, 2 max_len + sample bits <= 58, at most 16 distinct lengths,
// at most 16 samples): duplication-aware.  What the measurements of round 5 said about this chip decides its shape:
// scattered 8-byte stores run at ~100 G/s whatever they carry (so every scatter is staged in LDS and written run by
// run), a wave waits for its slowest lane (so every data-dependent loop runs once per lane, never once per slot), and
// 64 same-address LDS atomics are 64 serial ones (so the hot keys are counted apart).
//   K0 prepass     ONE pass over the batch: readLengthDic (what the call returns anyway), the largest sample id, the raw
//                  reads per (workgroup chunk, L1 bucket), and a SAMPLE (the first 256 reads of each of the 1024 chunks)
//                  into a global hash table.  An L1 bucket = (length, the 8 most significant bits of the 2 L-bit packed
//                  read): buckets are ORDERED as the output is.  A prefix sum over [bucket][chunk] gives every (bucket,
//                  chunk) a private region sized for its raw reads -- an upper bound of what K1 writes there, so K1
//                  needs no global atomic at all;
//   hot keys       the (at most 1024) sample entries seen most often: the sequences that are most of a small-RNA sample;
//   K1 split       a workgroup copies the hot keys into an LDS table (buckets of four keys, one 32-byte read per
//                  lookup) with a counter each and streams its chunk: a read that is a hot key bumps a counter -- the
//                  miRNA that is 8 % of the sample leaves a chunk as ONE (rest of key, count) pair, not as 8 000 --,
//                  every other read becomes a pair with count 1.  A trip's pairs are grouped by L1 bucket in LDS and
//                  appended to the chunk's regions run by run; what a region has left at the end is filled with zero
//                  pairs, so that K2 streams whole buckets;
//   K2 subdivide   a workgroup = (L1 bucket, one of 32 parts of its region): the pairs counted by their next b2 <= 8 bits
//                  (K2a), one prefix sum over [final bucket][part], and the pairs copied -- staged in LDS by final
//                  bucket, written run by run -- into contiguous FINAL buckets of ~1000 pairs, in key order (K2b);
//   K3 reduce      one workgroup per final bucket: pairs into an LDS hash table (equal keys summed by the inserts), its
//                  occupied slots counting-sorted by the top 8 bits the bucket leaves open, an entry's place = its bin's
//                  start + the members of its bin with a smaller key (a handful: it looks at each).  Sorted (key, count)
//                  entries go back to the bucket's region;
//   K4 emit        prefix sum of the buckets' read counts, entries -> u_words / u_lens / quant[u][sample].
//   A final bucket with more distinct keys than its table (~1500: an L1 bucket with more than 400 000 distinct reads --
//   a batch whose reads all END in the same 4 bases) raises a flag and the batch takes the general path: slower,
//   never wrong.
//
// GENERAL PATH (several words per read, N masks, reads beyond 29 nt): stable LSD radix sort (prims.hip) of read ids
// by (sample, packed words, N mask, length) column by column, head flags on the sorted order, prefix sums for the
// unique id and the run starts, one thread per (read, sample) run for the counts.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
constexpr uint32_t kFastMaxLen = 29u;
constexpr uint32_t kLenBins = (kFastMaxLen + 1u) * 256u;  // K0 counts by (length, top 8 bits), whatever lengths occur
constexpr uint32_t kL1Bits = 8u;
constexpr uint32_t kAggSlotBits = 11u;
constexpr uint32_t kAggSlots = 1u << kAggSlotBits;  // K1's LDS table of hot keys
constexpr uint32_t kHotKeys = 1024u;         // ... holds at most this many
constexpr uint32_t kSampleLanes = 64u;       // K0: the first 4 x 64 reads of a chunk are the sample (256 K reads of 1024 chunks)
constexpr uint32_t kSampleBits = 19u;        // slots of the sample's global hash table
constexpr uint32_t kPairMaxCount = 16383u;   // a pair's count has 14 bits
constexpr uint32_t kCountShift = 50u;        // pair = count << 50 | rest of the key (count 0: an unused position)
constexpr uint64_t kRestMask = (1ull << kCountShift) - 1ull;
constexpr uint32_t kRedThreads = 256u;       // K3, K4
constexpr uint32_t kRedSlotBits = 11u;
constexpr uint32_t kRedSlots = 1u << kRedSlotBits;  // K3's LDS table: a final bucket may hold ~1500 distinct keys
constexpr uint64_t kEmpty = ~0ull;

struct FastShape {
  uint32_t n, n_chunks, chunk;  // reads, K1 workgroups, reads per workgroup (a multiple of 4096)
  uint32_t n_slots, n_bins;     // lengths present, n_slots x 256
  uint32_t sb;                  // sample bits below the bases in a key
  uint32_t groups;              // K2 workgroups per L1 bucket
  uint8_t slot_of_len[64];      // length -> slot (0xFF: absent)
  uint8_t len_of_slot[kMaxLenSlots];
};

__device__ __forceinline__ uint32_t l1_shift(uint32_t L) { return 2u * L > kL1Bits ? 2u * L - kL1Bits : 0u; }

// K0: one pass over the batch: readLengthDic (reads per (length, sample): LENH), the largest sample id, and the raw
// reads per (chunk, length, top 8 bits of the packed read), stored [length x 256 + top][chunk] -- for reads of at most
// 29 nt; a longer read only counts in the length histogram (the batch then takes the general path).
template <bool LENH>
__global__ void __launch_bounds__(kFastThreads) prepass_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                                               const uint16_t* __restrict__ sample, uint32_t n, uint32_t chunk,
                                                               uint32_t n_chunks, uint32_t n_samples, uint32_t* __restrict__ counts_t,
                                                               unsigned long long* __restrict__ len_hist, uint32_t* __restrict__ max_sample,
                                                               unsigned long long* __restrict__ smp_key, uint32_t* __restrict__ smp_cnt,
                                                               uint32_t sb) {
  extern __shared__ uint32_t smem_u32[];
  uint32_t* hist = smem_u32;             // kLenBins
  uint32_t* lhist = smem_u32 + kLenBins;  // 256 x n_samples (LENH)
  for (uint32_t b = threadIdx.x; b < kLenBins; b += kFastThreads) hist[b] = 0u;
  if (LENH)
    for (uint32_t b = threadIdx.x; b < 256u * n_samples; b += kFastThreads) lhist[b] = 0u;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * chunk;
  const uint64_t hi = min((uint64_t)n, lo + chunk);
  uint32_t ms = 0;
  // four consecutive reads per lane and trip (chunk and lo are multiples of 4: 4- and 32-byte aligned vector loads)
  for (uint64_t i0 = lo; i0 < hi; i0 += 4ull * kFastThreads) {
    const uint64_t i = i0 + 4ull * threadIdx.x;  // (may lie beyond hi: the lane then only takes part in the ballots)
    uint32_t L4;
    uint64_t w[4];
    uint32_t sm[4] = {0u, 0u, 0u, 0u};
    if (i + 4u <= hi) {
      L4 = *reinterpret_cast<const uint32_t*>(lens + i);
      const uint4 a = *reinterpret_cast<const uint4*>(words + i), b = *reinterpret_cast<const uint4*>(words + i + 2);
      w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32);
      w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
      w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32);
      w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
      if (sample) {
        const uint2 s2 = *reinterpret_cast<const uint2*>(sample + i);
        sm[0] = s2.x & 0xFFFFu, sm[1] = s2.x >> 16, sm[2] = s2.y & 0xFFFFu, sm[3] = s2.y >> 16;
      }
    } else {
      L4 = 0u;
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        w[k] = 0ull;
        if (i + k < hi) {
          L4 |= (uint32_t)lens[i + k] << (8u * k);
          w[k] = words[i + k];
          if (sample) sm[k] = sample[i + k];
        }
      }
    }
    if (i0 == lo && threadIdx.x < kSampleLanes) {
      // the SAMPLE: the first 4 x kSampleLanes reads of every chunk into a global hash table (key -> occurrences):
      // what occurs there more than once is probably frequent everywhere (K1's hot table is chosen from it)
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t L = (L4 >> (8u * k)) & 255u;
        if (i + k >= hi || L > kFastMaxLen) continue;
        const unsigned long long key = ((unsigned long long)L << 58) | ((w[k] << sb) | (uint64_t)min(sm[k], n_samples - 1u));
        uint32_t s = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64u - kSampleBits));
        for (uint32_t tries = 0; tries < 64u; ++tries) {
          const unsigned long long old = atomicCAS(&smp_key[s], kEmpty, key);
          if (old == kEmpty || old == key) {
            atomicAdd(&smp_cnt[s], 1u);
            break;
          }
          s = (s + 1u) & ((1u << kSampleBits) - 1u);
        }
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      if (i + k >= hi) break;
      const uint32_t L = (L4 >> (8u * k)) & 255u;
      // (& 255: a word with bits beyond its 2 L -- not a packed read -- lands in SOME bucket of its length, never outside)
      if (L <= kFastMaxLen) atomicAdd(&hist[L * 256u + ((uint32_t)(w[k] >> l1_shift(L)) & 255u)], 1u);
      ms = max(ms, sm[k]);
    }
    if (LENH) {
      // nearly every read has the same (length, sample): the lanes that agree with the first active lane add ONCE
      // (64 same-address LDS atomics are 64 serial ones); whoever differs adds for itself
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        const bool live = i + k < hi;
        const uint32_t bin = ((L4 >> (8u * k)) & 255u) * n_samples + min(sm[k], n_samples - 1u);
        const uint64_t lm = __ballot(live);
        if (!lm) continue;
        const uint32_t first = __shfl(bin, __ffsll((long long)lm) - 1, 64);
        const uint64_t same = __ballot(live && bin == first);
        if (live && bin != first) atomicAdd(&lhist[bin], 1u);
        if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)lm) - 1)) atomicAdd(&lhist[first], (uint32_t)__popcll(same));
      }
    }
  }
  __syncthreads();
  // (the table was zeroed: 7 680 scattered 4-byte stores per chunk -- 7.8 M a batch -- for the 256 that are not zero)
  for (uint32_t b = threadIdx.x; b < kLenBins; b += kFastThreads)
    if (hist[b]) counts_t[(size_t)b * n_chunks + blockIdx.x] = hist[b];
  if (LENH)
    for (uint32_t b = threadIdx.x; b < 256u * n_samples; b += kFastThreads)
      if (lhist[b]) atomicAdd(&len_hist[b], (unsigned long long)lhist[b]);
  if (sample) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ms = max(ms, (uint32_t)__shfl_down(ms, off, 64));
    if ((threadIdx.x & 63u) == 0u && ms) atomicMax(max_sample, ms);
  }
}

// The hot keys: sample entries seen at least T times, T = the lowest count that keeps at most kHotKeys of them
// (hist: entries by min(count, 255), filled by hot_hist_kernel; every workgroup of hot_pick_kernel derives T for itself).
__global__ void __launch_bounds__(256) hot_hist_kernel(const uint32_t* __restrict__ smp_cnt, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < (1u << kSampleBits); s += gridDim.x * 256u) {
    const uint32_t c = smp_cnt[s];
    if (c >= 2u) atomicAdd(&h[min(c, 255u)], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) hot_pick_kernel(const unsigned long long* __restrict__ smp_key, const uint32_t* __restrict__ smp_cnt,
                                                       const uint32_t* __restrict__ hist, unsigned long long* __restrict__ hot,
                                                       uint32_t* __restrict__ n_hot) {
  __shared__ uint32_t T;
  if (threadIdx.x == 0) {
    uint32_t keep = 0, t = 256u;
    for (uint32_t c = 255u; c >= 2u; --c) {
      if (keep + hist[c] > kHotKeys) break;
      keep += hist[c];
      t = c;
    }
    T = t;
  }
  __syncthreads();
  for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < (1u << kSampleBits); s += gridDim.x * 256u) {
    if (smp_cnt[s] >= T) {
      const uint32_t at = atomicAdd(n_hot, 1u);
      if (at < kHotKeys) hot[at] = smp_key[s];
    }
  }
}

// K1: split a chunk into what its hot table absorbs and what leaves at once.  The workgroup copies the hot keys into
// an LDS hash table (read-only from then on) with a counter each; a read whose key is there bumps the counter -- the
// sequence that is a percent of the sample leaves the chunk ONCE, at the end, with its count -- every other read
// becomes a (rest of key, 1) pair in its L1 bucket's region on the spot: no flush, no barrier in the loop.
// A region is sized for the chunk's raw reads of the bucket: what the pairs leave of it is filled with zero pairs, so
// that K2 streams whole buckets.  off_t: exclusive prefix of K0's counts (by length x 256 + top, one entry past the end).
__global__ void __launch_bounds__(kFastThreads) split_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                                             const uint16_t* __restrict__ sample, FastShape f,
                                                             const uint32_t* __restrict__ off_t, const unsigned long long* __restrict__ hot,
                                                             const uint32_t* __restrict__ n_hot_p, uint32_t* __restrict__ bin_pairs,
                                                             uint32_t* __restrict__ cnt_t, uint64_t* __restrict__ pairs) {
  constexpr uint32_t kPer = 4u, kTrip = kPer * kFastThreads;  // reads per lane and trip; reads per trip
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* tkey = reinterpret_cast<unsigned long long*>(smem);         // kAggSlots: the hot keys
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem + kAggSlots * 8u);            // kTrip: a trip's pairs, grouped by bucket
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(smem + kAggSlots * 8u + kTrip * 8u);  // kAggSlots: what a hot key counted
  uint32_t* cursor = tcnt + kAggSlots;   // n_bins: where the next pair of a bucket goes
  uint32_t* tcount = cursor + f.n_bins;  // n_bins: pairs of the trip per bucket
  uint32_t* tstart = tcount + f.n_bins;  // n_bins: their first position in `stage`
  uint32_t* wtot = tstart + f.n_bins;    // 16 + 1
  uint16_t* stage_bin = reinterpret_cast<uint16_t*>(wtot + 32);  // kTrip
  for (uint32_t s = threadIdx.x; s < kAggSlots; s += kFastThreads) {
    tkey[s] = kEmpty;
    tcnt[s] = 0u;
  }
  for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads) {
    cursor[b] = off_t[((size_t)f.len_of_slot[b >> 8] * 256u + (b & 255u)) * f.n_chunks + blockIdx.x];
    tcount[b] = 0u;
  }
  __syncthreads();
  // buckets of four keys (one 32-byte read looks at a whole bucket): a key sits in the first free slot of its home
  // bucket or, when that is full, of the buckets behind it -- a lookup that meets a bucket with a free slot is over
  const uint32_t n_hot = min(*n_hot_p, kHotKeys);
  for (uint32_t h = threadIdx.x; h < n_hot; h += kFastThreads) {
    const unsigned long long key = hot[h];
    uint32_t s = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (66u - kAggSlotBits)) * 4u;
    while (atomicCAS(&tkey[s], kEmpty, key) != kEmpty) s = (s + 1u) & (kAggSlots - 1u);
  }
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * f.chunk;
  const uint64_t hi = min((uint64_t)f.n, lo + f.chunk);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t per_thread = (f.n_bins + kFastThreads - 1u) / kFastThreads;  // buckets a thread sums in the trip's prefix
  for (uint64_t t0 = lo; t0 < hi; t0 += kTrip) {
    const uint64_t i0 = t0 + (uint64_t)kPer * threadIdx.x;
    uint64_t w[kPer];
    uint32_t L4 = 0;
    uint32_t sm[kPer] = {0u, 0u, 0u, 0u};
    if (i0 + kPer <= hi) {
      L4 = *reinterpret_cast<const uint32_t*>(lens + i0);
      const uint4 a = *reinterpret_cast<const uint4*>(words + i0), b = *reinterpret_cast<const uint4*>(words + i0 + 2);
      w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32);
      w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
      w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32);
      w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
      if (sample) {
        const uint2 s2 = *reinterpret_cast<const uint2*>(sample + i0);
        sm[0] = s2.x & 0xFFFFu, sm[1] = s2.x >> 16, sm[2] = s2.y & 0xFFFFu, sm[3] = s2.y >> 16;
      }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        w[k] = 0ull;
        if (i0 + k < hi) {
          L4 |= (uint32_t)lens[i0 + k] << (8u * k);
          w[k] = words[i0 + k];
          if (sample) sm[k] = sample[i0 + k];
        }
      }
    }
    uint64_t pr[kPer];   // 0: absorbed by the hot table (or no read)
    uint32_t pb[kPer];   // bucket << 16 | rank among the trip's pairs of that bucket
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      pr[k] = 0ull;
      pb[k] = 0u;
      if (i0 + k >= hi) continue;
      const uint32_t L = (L4 >> (8u * k)) & 255u;
      const uint64_t v = (w[k] << f.sb) | (uint64_t)sm[k];
      const unsigned long long key = ((unsigned long long)L << 58) | v;
      uint32_t s = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (66u - kAggSlotBits)) * 4u;
      bool hit = false;
      for (;;) {
        const ulonglong2 p0 = *reinterpret_cast<const ulonglong2*>(tkey + s), p1 = *reinterpret_cast<const ulonglong2*>(tkey + s + 2u);
        const uint32_t m = (p0.x == key ? 1u : 0u) | (p0.y == key ? 2u : 0u) | (p1.x == key ? 4u : 0u) | (p1.y == key ? 8u : 0u);
        if (m) {
          hit = true;
          s += (uint32_t)__ffs((int)m) - 1u;
          break;
        }
        if (p1.y == kEmpty) break;  // (slots fill from the front: a free last slot = nothing of this home went further)
        s = (s + 4u) & (kAggSlots - 1u);
      }
      if (hit) {
        atomicAdd(&tcnt[s], 1u);
      } else {
        const uint32_t sh = l1_shift(L) + f.sb;
        const uint32_t bin = (uint32_t)f.slot_of_len[L & 63u] * 256u + ((uint32_t)(v >> sh) & 255u);
        pr[k] = (1ull << kCountShift) | (v & ((1ull << sh) - 1ull));
        pb[k] = (bin << 16) | atomicAdd(&tcount[bin], 1u);
      }
    }
    __syncthreads();
    // the trip's pairs grouped by bucket in LDS, then written run by run: neighbouring lanes write neighbouring pairs
    // (scattered 8-byte stores run at ~100 G/s on this chip whatever they carry)
    {
      uint32_t sum = 0;
      for (uint32_t q = 0; q < per_thread; ++q) {
        const uint32_t b = threadIdx.x * per_thread + q;
        sum += b < f.n_bins ? tcount[b] : 0u;
      }
      const uint32_t incl = dev::wave_incl_scan(sum);
      if (lane == 63u) wtot[wave] = incl;
      __syncthreads();
      uint32_t st = incl - sum;
      for (uint32_t ww = 0; ww < wave; ++ww) st += wtot[ww];
      for (uint32_t q = 0; q < per_thread; ++q) {
        const uint32_t b = threadIdx.x * per_thread + q;
        if (b < f.n_bins) {
          tstart[b] = st;
          st += tcount[b];
        }
      }
      if (threadIdx.x == kFastThreads - 1u) wtot[16] = st;
    }
    __syncthreads();
    const uint32_t total = wtot[16];
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k)
      if (pr[k]) {
        const uint32_t at = tstart[pb[k] >> 16] + (pb[k] & 0xFFFFu);
        stage[at] = pr[k];
        stage_bin[at] = (uint16_t)(pb[k] >> 16);
      }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total; i += kFastThreads) {
      const uint32_t bin = stage_bin[i];
      pairs[cursor[bin] + (i - tstart[bin])] = stage[i];
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads) {
      cursor[b] += tcount[b];
      tcount[b] = 0u;
    }
    // (the next trip's first barrier orders this against its uses)
  }
  __syncthreads();
  // the hot entries with what they counted
  for (uint32_t s = threadIdx.x; s < kAggSlots; s += kFastThreads) {
    const unsigned long long key = tkey[s];
    uint32_t c = key != kEmpty ? tcnt[s] : 0u;
    if (!c) continue;
    const uint32_t KL = (uint32_t)(key >> 58);
    const uint64_t v = key & ((1ull << 58) - 1ull);
    const uint32_t sh = l1_shift(KL) + f.sb;
    const uint32_t bin = (uint32_t)f.slot_of_len[KL] * 256u + ((uint32_t)(v >> sh) & 255u);
    const uint64_t rest = v & ((1ull << sh) - 1ull);
    while (c) {  // (a count beyond the pair's 14 bits leaves in pieces: each stands for at least one read of the region)
      const uint32_t piece = min(c, kPairMaxCount);
      pairs[atomicAdd(&cursor[bin], 1u)] = ((uint64_t)piece << kCountShift) | rest;
      c -= piece;
    }
  }
  __syncthreads();
  // pairs the chunk left in each of its regions (K2 walks exactly those: a region is sized for the chunk's RAW reads of the
  // bucket, the hot table and the per-chunk aggregation leave ~35 % of it unused), and the buckets' totals
  for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads) {
    const size_t at = ((size_t)f.len_of_slot[b >> 8] * 256u + (b & 255u)) * f.n_chunks + blockIdx.x;
    const uint32_t first = off_t[at], cur = cursor[b];
    cnt_t[at] = cur - first;
    if (cur > first) atomicAdd(&bin_pairs[b], cur - first);
  }
}

// K2a / K2b: the pairs of an L1 bucket counted by their next b2 bits, then copied into contiguous final buckets.  A
// workgroup = (L1 bucket, one of `groups` runs of chunks); a wave walks the regions of its chunks (every 16th of the
// run), exactly the pairs K1 left there (cnt_t), 64 at a stride.  The counts are stored [final bucket][group], so ONE
// prefix sum over the array places every (final bucket, group) run: final buckets in key order, dense.
// b2 = enough bits for ~1024 pairs per final bucket.
__device__ __forceinline__ uint32_t sub_bits_of(uint32_t P, uint32_t r1) {
  uint32_t b2 = 0;
  while (b2 < 8u && b2 < r1 && ((P + 1023u) >> 10) > (1u << b2)) ++b2;
  return b2;
}

template <bool SCATTER>
__global__ void __launch_bounds__(kFastThreads) subdivide_kernel(FastShape f, const uint32_t* __restrict__ off_t, const uint32_t* __restrict__ cnt_t,
                                                                 const uint32_t* __restrict__ bin_pairs, const uint64_t* __restrict__ pairs_in,
                                                                 uint32_t* __restrict__ hist_t /* SCATTER: its exclusive prefix */,
                                                                 uint64_t* __restrict__ pairs_out, uint8_t* __restrict__ l1_b2) {
  // SCATTER stages a tile of up to 8192 pairs in LDS, grouped by final bucket, and writes it out run by run: the lanes of
  // a store instruction then write neighbouring pairs (a 128-byte line takes ONE request, not sixteen 8-byte ones:
  // scattered 8-byte stores run at ~100 G/s on this chip whatever they carry)
  constexpr uint32_t kSlots = 8u;  // strides of 64 pairs a wave takes per tile
  constexpr uint32_t kWaves = kFastThreads / 64u;
  constexpr uint32_t kTilePos = kSlots * kFastThreads;
  __shared__ uint32_t sub[256];    // counts (hist pass); SCATTER: where the next tile's run of a final bucket goes
  __shared__ uint32_t tcount[256], tstart[256];
  __shared__ uint32_t wtot[kWaves];
  __shared__ uint32_t seg_first[kMaxChunks], seg_n[kMaxChunks];  // the regions of this workgroup's chunks
  __shared__ uint32_t n_tiles_s;
  __shared__ uint64_t stage[SCATTER ? kTilePos : 1];
  const uint32_t b = blockIdx.x / f.groups, g = blockIdx.x % f.groups;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t P = bin_pairs[b];
  if (P == 0) return;
  if (tid < 256u) sub[tid] = SCATTER ? hist_t[((size_t)b * 256u + tid) * f.groups + g] : 0u;
  if (tid == 0) n_tiles_s = 0u;
  const uint32_t L = f.len_of_slot[b >> 8];
  const uint32_t r1 = l1_shift(L) + f.sb;
  const uint32_t b2 = sub_bits_of(P, r1);
  const uint32_t sh = r1 - b2;
  if (!SCATTER && g == 0 && tid == 0) l1_b2[b] = (uint8_t)b2;
  const size_t at = ((size_t)L * 256u + (b & 255u)) * f.n_chunks;
  const uint32_t cpg = (f.n_chunks + f.groups - 1u) / f.groups;  // chunks per group
  const uint32_t c_lo = min(f.n_chunks, g * cpg), n_seg = min(f.n_chunks, c_lo + cpg) - c_lo;
  for (uint32_t j = tid; j < n_seg; j += kFastThreads) {
    seg_first[j] = off_t[at + c_lo + j];
    seg_n[j] = cnt_t[at + c_lo + j];
  }
  __syncthreads();
  {
    // strides of the wave's segments (wave, wave + 16, ...) -> tiles it needs; the workgroup walks the largest number
    uint32_t strides = 0;
    for (uint32_t j = wave; j < n_seg; j += kWaves) strides += (seg_n[j] + 63u) >> 6;
    if (lane == 0 && strides) atomicMax(&n_tiles_s, (strides + kSlots - 1u) / kSlots);
  }
  __syncthreads();
  const uint32_t n_tiles = n_tiles_s;
  uint32_t sj = wave, spos = 0;  // the wave's current segment and how far it is in (wave-uniform)
  for (uint32_t tile = 0; tile < n_tiles; ++tile) {
    uint64_t pr[kSlots];
    uint32_t rk[kSlots];  // SCATTER: rank of the pair among the tile's pairs of its final bucket
#pragma unroll
    for (uint32_t k = 0; k < kSlots; ++k) {
      while (sj < n_seg && spos >= seg_n[sj]) {
        sj += kWaves;
        spos = 0u;
      }
      pr[k] = 0ull;
      if (sj < n_seg) {
        if (spos + lane < seg_n[sj]) pr[k] = pairs_in[seg_first[sj] + spos + lane];
        spos += 64u;
      }
    }
    if (SCATTER) {
      if (tid < 256u) tcount[tid] = 0u;
      __syncthreads();
    }
#pragma unroll
    for (uint32_t k = 0; k < kSlots; ++k)
      if (pr[k] >> kCountShift) rk[k] = atomicAdd(SCATTER ? &tcount[(uint32_t)((pr[k] & kRestMask) >> sh) & 255u] : &sub[(uint32_t)((pr[k] & kRestMask) >> sh) & 255u], 1u);
    if (SCATTER) {
      __syncthreads();
      // tile-local starts of the 256 runs
      uint32_t c = 0, incl = 0;
      if (tid < 256u) {
        c = tcount[tid];
        incl = dev::wave_incl_scan(c);
        if (lane == 63u) wtot[wave] = incl;
      }
      __syncthreads();
      uint32_t total = 0;
      if (tid < 256u) {
        uint32_t st = incl - c;
        for (uint32_t w = 0; w < wave; ++w) st += wtot[w];
        tstart[tid] = st;
      }
      total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
      __syncthreads();
#pragma unroll
      for (uint32_t k = 0; k < kSlots; ++k)
        if (pr[k] >> kCountShift) stage[tstart[(uint32_t)((pr[k] & kRestMask) >> sh) & 255u] + rk[k]] = pr[k];
      __syncthreads();
      for (uint32_t i = tid; i < total; i += kFastThreads) {
        const uint64_t p = stage[i];
        const uint32_t sb2 = (uint32_t)((p & kRestMask) >> sh) & 255u;
        pairs_out[sub[sb2] + (i - tstart[sb2])] = p;
      }
      __syncthreads();
      if (tid < 256u) sub[tid] += tcount[tid];
      // (the next trip's first barrier orders this against its reads of sub / writes of tcount)
    }
  }
  if (!SCATTER) {
    __syncthreads();
    if (tid < 256u) hist_t[((size_t)b * 256u + tid) * f.groups + g] = sub[tid];
  }
}

// the non-empty final buckets, in order: flag -> prefix -> list
// (offs has one entry past the last (final bucket, group): the number of pairs; flag one past the last bucket: after its
// prefix sum, the length of the list)
__global__ void __launch_bounds__(256) fb_flags_kernel(const uint32_t* __restrict__ offs, uint32_t n_fb, uint32_t groups,
                                                       uint32_t* __restrict__ flag) {
  const uint32_t fb = blockIdx.x * 256u + threadIdx.x;
  if (fb > n_fb) return;
  flag[fb] = (fb < n_fb && offs[(size_t)(fb + 1u) * groups] > offs[(size_t)fb * groups]) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) fb_list_kernel(const uint32_t* __restrict__ flag_excl, const uint32_t* __restrict__ offs, uint32_t n_fb,
                                                      uint32_t groups, uint32_t* __restrict__ work) {
  const uint32_t fb = blockIdx.x * 256u + threadIdx.x;
  if (fb >= n_fb) return;
  if (offs[(size_t)(fb + 1u) * groups] > offs[(size_t)fb * groups]) work[flag_excl[fb]] = fb;
}

// K3: one final bucket per trip:
//   pairs -> LDS hash table (scrambled hash, short probes; equal keys summed by the inserts)
//   -> the distinct entries, compacted
//   -> counting sort by the top 8 bits of what the bucket leaves open of the key (256 bins, a handful of entries each)
//   -> an entry's place = its bin's start + the entries of its bin with a smaller key (it looks at each of them)
//   -> (full key, count) written at the bucket's start + that place.
// Every loop a lane runs is as long as ITS entry needs (probe length, bin size), one entry per lane: a wave never
// waits nine times for its longest run, which is what a slot-by-slot sweep of a monotone table cost (4 ms).
__global__ void __launch_bounds__(kRedThreads) reduce_kernel(FastShape f, const uint32_t* __restrict__ work,
                                                             const uint32_t* __restrict__ n_work_p, const uint32_t* __restrict__ offs,
                                                             const uint8_t* __restrict__ l1_b2, const uint64_t* __restrict__ pairs,
                                                             uint64_t* __restrict__ ent_key, uint32_t* __restrict__ ent_cnt,
                                                             uint32_t* __restrict__ fb_entries, uint32_t* __restrict__ fb_reads,
                                                             uint32_t* __restrict__ overflow) {
  __shared__ unsigned long long tkey[kRedSlots];   // the table
  __shared__ uint32_t tcnt[kRedSlots];
  __shared__ uint16_t sidx[kRedSlots];             // the occupied slots grouped by bin
  __shared__ uint32_t bin_cnt[256], bin_start[257];
  __shared__ uint32_t wtot[kRedThreads / 64u];
  constexpr uint32_t kRedMaxProbe = 128u;
  constexpr uint32_t kPerThread = kRedSlots / kRedThreads;  // consecutive slots per thread
  constexpr uint32_t kLoads = 4u;                           // pairs a lane requests before it inserts any
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_work = *n_work_p;
  // the bucket of the NEXT trip is looked up while this trip's works (three dependent loads otherwise head every trip)
  uint32_t n_fb = 0, n_start = 0, n_end = 0, n_b2 = 0;
  if (blockIdx.x < n_work) {
    n_fb = work[blockIdx.x];
    n_start = offs[(size_t)n_fb * f.groups];
    n_end = offs[(size_t)(n_fb + 1u) * f.groups];
    n_b2 = l1_b2[n_fb >> 8];
  }
  for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const uint32_t fb = n_fb, start = n_start, P = n_end - n_start, b2 = n_b2;
    if (wi + gridDim.x < n_work) {
      n_fb = work[wi + gridDim.x];
      n_start = offs[(size_t)n_fb * f.groups];
      n_end = offs[(size_t)(n_fb + 1u) * f.groups];
      n_b2 = l1_b2[n_fb >> 8];
    }
    const uint32_t l1 = fb >> 8, sub = fb & 255u;
    const uint32_t L = f.len_of_slot[l1 >> 8];
    const uint32_t r1 = l1_shift(L) + f.sb, r2 = r1 - b2;
    for (uint32_t s = tid; s < kRedSlots; s += kRedThreads) {
      tkey[s] = kEmpty;
      tcnt[s] = 0u;
    }
    bin_cnt[tid] = 0u;
    __syncthreads();
    bool lost = false;
    for (uint32_t j0 = tid; j0 < P; j0 += kRedThreads * kLoads) {
      uint64_t pr[kLoads];
#pragma unroll
      for (uint32_t u = 0; u < kLoads; ++u) pr[u] = j0 + u * kRedThreads < P ? pairs[start + j0 + u * kRedThreads] : 0ull;
#pragma unroll
      for (uint32_t u = 0; u < kLoads; ++u) {
        if (j0 + u * kRedThreads >= P) break;
        const unsigned long long rest = pr[u] & kRestMask;
        uint32_t s = (uint32_t)((rest * 0x9E3779B97F4A7C15ull) >> (64u - kRedSlotBits));
        // (no count of the occupied slots while inserting -- one same-address atomic per new key was a third of the
        // kernel's LDS time --: a probe that walks kRedMaxProbe slots says the table is not this bucket's size, and the
        // occupancy is checked once, after the inserts)
        for (uint32_t step = 0;; ++step) {
          const unsigned long long old = atomicCAS(&tkey[s], kEmpty, rest);
          if (old == kEmpty || old == rest) {
            atomicAdd(&tcnt[s], (uint32_t)(pr[u] >> kCountShift));
            break;
          }
          s = (s + 1u) & (kRedSlots - 1u);
          if (step >= kRedMaxProbe) {  // (never spin in a full table)
            lost = true;
            break;
          }
        }
      }
    }
    if (lost) atomicOr(overflow, 1u);
    __syncthreads();
    // counting sort of the occupied slots by bin = the top 8 bits of the key's open part (monotone in the key)
    const uint32_t bsh = r2 > 8u ? r2 - 8u : 0u;
    const uint64_t open_mask = (1ull << r2) - 1ull;
    const uint32_t s_lo = tid * kPerThread;
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      const unsigned long long key = tkey[s_lo + k];
      if (key != kEmpty) atomicAdd(&bin_cnt[(uint32_t)((key & open_mask) >> bsh) & 255u], 1u);
    }
    __syncthreads();
    {
      const uint32_t c = bin_cnt[tid];
      const uint32_t incl = dev::wave_incl_scan(c);
      if (lane == 63u) wtot[wave] = incl;
      __syncthreads();
      uint32_t st = incl - c;
#pragma unroll
      for (uint32_t w = 0; w < kRedThreads / 64u; ++w) st += w < wave ? wtot[w] : 0u;
      bin_start[tid] = st;
      if (tid == 255u) bin_start[256] = st + c;
      bin_cnt[tid] = st;  // (now the bin's write cursor)
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      const unsigned long long key = tkey[s_lo + k];
      if (key != kEmpty) sidx[atomicAdd(&bin_cnt[(uint32_t)((key & open_mask) >> bsh) & 255u], 1u)] = (uint16_t)(s_lo + k);
    }
    __syncthreads();
    const uint32_t D = bin_start[256];
    if (tid == 0 && D > kRedSlots - kRedSlots / 4u) atomicOr(overflow, 1u);  // (three quarters full: not this table's bucket)
    // an entry's place in its bin = members with a smaller key; it is the first entry of its read when no member with
    // a smaller key has the same bases (entries of one read differ in their sample bits only: same bin, or the key is
    // shorter than 8 + sample bits and the bins hold one key each)
    const uint64_t prefix = ((uint64_t)(l1 & 255u) << r1) | ((uint64_t)sub << r2);  // the key bits the bucket stands for
    uint32_t n_rd = 0;
    for (uint32_t e = tid; e < D; e += kRedThreads) {
      const uint32_t slot = sidx[e];
      const unsigned long long k = tkey[slot];
      const uint32_t bn = (uint32_t)((k & open_mask) >> bsh) & 255u;
      const uint32_t b_lo = bin_start[bn], b_hi = bin_start[bn + 1u];
      uint32_t rank = 0;
      bool head = true;
      for (uint32_t j = b_lo; j < b_hi; ++j) {
        const unsigned long long o = tkey[sidx[j]];
        rank += o < k ? 1u : 0u;
        head = head && !(o < k && (o >> f.sb) == (k >> f.sb));
      }
      if (f.sb && bsh < f.sb && head) {
        // (tiny keys: the read's other samples may sit in the bins in front)
        for (uint32_t j = 0; j < b_lo; ++j) head = head && (tkey[sidx[j]] >> f.sb) != (k >> f.sb);
      }
      n_rd += head ? 1u : 0u;
      const uint32_t at = start + b_lo + rank;
      ent_key[at] = prefix | (k & open_mask);
      ent_cnt[at] = tcnt[slot];
    }
    const uint32_t ir = dev::wave_incl_scan(n_rd);
    __syncthreads();
    if (lane == 63u) wtot[wave] = ir;
    __syncthreads();
    if (tid == 0) {
      fb_entries[fb] = D;
      fb_reads[fb] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    }
    __syncthreads();
  }
}

// K4: entries of a final bucket -> the output arrays at the bucket's first read (read_base = exclusive prefix of fb_reads)
__global__ void __launch_bounds__(kRedThreads) emit_fast_kernel(FastShape f, const uint32_t* __restrict__ work,
                                                                const uint32_t* __restrict__ n_work_p, const uint32_t* __restrict__ offs,
                                                                const uint32_t* __restrict__ fb_entries, const uint32_t* __restrict__ read_base,
                                                                const uint64_t* __restrict__ ent_key, const uint32_t* __restrict__ ent_cnt,
                                                                uint32_t n_samples, uint64_t* __restrict__ u_words, uint8_t* __restrict__ u_lens,
                                                                uint32_t* __restrict__ quant) {
  __shared__ uint32_t wtot[kRedThreads / 64u];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t smask = (1ull << f.sb) - 1ull;
  const uint32_t n_work = *n_work_p;
  for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const uint32_t fb = work[wi];
    const uint32_t L = f.len_of_slot[fb >> 16];
    const uint32_t start = offs[(size_t)fb * f.groups], E = fb_entries[fb];
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

struct FastPlan {
  uint32_t n_chunks = 0, chunk = 0;
  DevBuf off_t;  // K0's counts, then their exclusive prefix: (kLenBins x n_chunks + 1) entries
  DevBuf cnt_t;  // K1: pairs a chunk left in each of its regions (same indexing)
  DevBuf max_sample;
  DevBuf sample;  // keys (8 B x 2^kSampleBits), counts (4 B x 2^kSampleBits), count histogram (256 x 4 B), n_hot, hot keys
};

// K0 (+ the prefix sum of its counts): issued before the host knows what the batch holds -- the length histogram it
// fills says that
hipError_t fast_prepass(const uint64_t* d_reads, const uint8_t* d_lens, const uint16_t* smp, uint32_t n, uint32_t n_samples,
                        uint64_t* d_len_hist, hipStream_t stream, FastPlan* plan, DevBuf* stmp) {
  plan->n_chunks = std::max<uint32_t>(1u, std::min<uint32_t>(kMaxChunks, (n + 8191u) / 8192u));
  plan->chunk = (uint32_t)((((uint64_t)n + plan->n_chunks - 1) / plan->n_chunks + 4095u) & ~4095ull);
  const size_t n_ct = (size_t)kLenBins * plan->n_chunks + 1;
  CK(plan->off_t.alloc(n_ct * 4));
  CK(plan->cnt_t.alloc(n_ct * 4));
  CK(plan->max_sample.alloc(4));
  CK(stmp->alloc(prims::scan_temp_bytes(n_ct)));
  const size_t n_smp = (size_t)1 << kSampleBits;
  CK(plan->sample.alloc(n_smp * 12 + 1024 + 64 + (size_t)kHotKeys * 8));
  unsigned long long* smp_key = plan->sample.as<unsigned long long>();
  uint32_t* smp_cnt = reinterpret_cast<uint32_t*>(smp_key + n_smp);
  uint32_t* smp_hist = smp_cnt + n_smp;
  uint32_t* n_hot = smp_hist + 256;
  unsigned long long* hot = reinterpret_cast<unsigned long long*>(n_hot + 16);
  CK(hipMemsetAsync(smp_key, 0xFF, n_smp * 8, stream));
  CK(hipMemsetAsync(smp_cnt, 0, n_smp * 4 + 1024 + 64, stream));
  CK(hipMemsetAsync(plan->max_sample.p, 0, 4, stream));
  CK(hipMemsetAsync(plan->off_t.p, 0, n_ct * 4, stream));
  uint32_t sb = 0;
  while ((1u << sb) < n_samples) ++sb;
  const uint32_t lds = (kLenBins + 256u * n_samples) * 4u;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(prepass_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(prepass_kernel<true>, dim3(plan->n_chunks), dim3(kFastThreads), lds, stream, d_reads, d_lens, smp, n, plan->chunk,
                     plan->n_chunks, n_samples, plan->off_t.as<uint32_t>(), reinterpret_cast<unsigned long long*>(d_len_hist),
                     plan->max_sample.as<uint32_t>(), smp_key, smp_cnt, sb);
  CK(hipGetLastError());
  hipLaunchKernelGGL(hot_hist_kernel, dim3(256), dim3(256), 0, stream, smp_cnt, smp_hist);
  hipLaunchKernelGGL(hot_pick_kernel, dim3(256), dim3(256), 0, stream, smp_key, smp_cnt, smp_hist, hot, n_hot);
  CK(hipGetLastError());
  return prims::exclusive_sum_u32(plan->off_t.as<uint32_t>(), plan->off_t.as<uint32_t>(), n_ct, stmp->p, stream);
}

// The fast path behind the prepass.  *took = false: the batch does not fit it (or overflowed a table): nothing was
// written that the general path does not overwrite.  h_len_hist: the length histogram of the batch ([256][S], on the host).
hipError_t collapse_fast(const uint64_t* d_reads, const uint8_t* d_lens, const uint16_t* smp, uint32_t n, uint32_t n_samples,
                         const std::vector<uint64_t>& h_len_hist, FastPlan& plan, uint64_t cap, uint64_t* d_u_words, uint8_t* d_u_lens,
                         uint32_t* d_quant, uint32_t* h_n_unique, int n_cu, hipStream_t stream, bool* took) {
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
    if (L > kFastMaxLen || f.n_slots == kMaxLenSlots) return hipSuccess;
    f.slot_of_len[L] = (uint8_t)f.n_slots;
    f.len_of_slot[f.n_slots++] = (uint8_t)L;
    max_len = L;
  }
  if (2u * max_len + f.sb > 58u || f.n_slots == 0) return hipSuccess;
  f.n_bins = f.n_slots * 256u;
  f.n_chunks = plan.n_chunks;
  f.chunk = plan.chunk;
  f.groups = getenv("MIRGE_COLLAPSE_GROUPS") ? (uint32_t)std::max(1, atoi(getenv("MIRGE_COLLAPSE_GROUPS"))) : 32u;
  const size_t n_fb = (size_t)f.n_bins * 256u, n_ht = n_fb * f.groups;
  DevBuf bufA, bufB, cnt, hist_t, fbs, misc, work, stmp;
  CK(bufA.alloc((size_t)n * 8 + 16));
  CK(bufB.alloc((size_t)n * 8 + 16));
  CK(cnt.alloc((size_t)n * 4));
  CK(hist_t.alloc((n_ht + 1) * 4));
  CK(fbs.alloc((n_fb + 1) * 4 * 4));   // entries, reads, read_base, work-list flags (one more: the list's length)
  CK(misc.alloc(f.n_bins * 5 + 64));   // pairs per L1 bucket, b2 per L1 bucket, the overflow flag
  CK(work.alloc(n_fb * 4));
  CK(stmp.alloc(prims::scan_temp_bytes(n_ht + 1)));
  uint32_t* fb_entries = fbs.as<uint32_t>();
  uint32_t* fb_reads = fb_entries + (n_fb + 1);
  uint32_t* read_base = fb_reads + (n_fb + 1);
  uint32_t* fb_flag = read_base + (n_fb + 1);
  uint32_t* bin_pairs = misc.as<uint32_t>();
  uint32_t* overflow = bin_pairs + f.n_bins;
  uint8_t* l1_b2 = reinterpret_cast<uint8_t*>(overflow + 4);
  const uint32_t* off_t = plan.off_t.as<uint32_t>();
  CK(hipMemsetAsync(fb_entries, 0, (n_fb + 1) * 4 * 2, stream));  // entries, reads (empty buckets count nothing)
  CK(hipMemsetAsync(misc.p, 0, f.n_bins * 5 + 64, stream));
  CK(hipMemsetAsync(hist_t.p, 0, (n_ht + 1) * 4, stream));        // (workgroups of empty L1 buckets write nothing)
  const uint32_t agg_lds = kAggSlots * 12u + 4096u * 10u + f.n_bins * 12u + 128u;
  const size_t n_smp = (size_t)1 << kSampleBits;
  const uint32_t* n_hot = reinterpret_cast<const uint32_t*>(plan.sample.as<unsigned long long>() + n_smp) + n_smp + 256;
  const unsigned long long* hot = reinterpret_cast<const unsigned long long*>(n_hot + 16);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)agg_lds));
  hipLaunchKernelGGL(split_kernel, dim3(f.n_chunks), dim3(kFastThreads), agg_lds, stream, d_reads, d_lens, smp, f, off_t, hot, n_hot, bin_pairs,
                     plan.cnt_t.as<uint32_t>(), bufA.as<uint64_t>());
  CK(hipGetLastError());
  const uint32_t sub_grid = f.n_bins * f.groups;
  uint32_t* offs = hist_t.as<uint32_t>();  // counts [final bucket][group] (+ one 0), then their exclusive prefix (+ the number of pairs)
  hipLaunchKernelGGL(subdivide_kernel<false>, dim3(sub_grid), dim3(kFastThreads), 0, stream, f, off_t, plan.cnt_t.as<uint32_t>(), bin_pairs,
                     bufA.as<uint64_t>(), offs, (uint64_t*)nullptr, l1_b2);
  CK(hipGetLastError());
  CK(prims::exclusive_sum_u32(offs, offs, n_ht + 1, stmp.p, stream));
  hipLaunchKernelGGL(subdivide_kernel<true>, dim3(sub_grid), dim3(kFastThreads), 0, stream, f, off_t, plan.cnt_t.as<uint32_t>(), bin_pairs,
                     bufA.as<uint64_t>(), offs,
                     bufB.as<uint64_t>(), l1_b2);
  CK(hipGetLastError());
  const uint32_t fgrid = (uint32_t)((n_fb + 1 + 255) / 256);
  hipLaunchKernelGGL(fb_flags_kernel, dim3(fgrid), dim3(256), 0, stream, offs, (uint32_t)n_fb, f.groups, fb_flag);
  CK(prims::exclusive_sum_u32(fb_flag, fb_flag, n_fb + 1, stmp.p, stream));
  hipLaunchKernelGGL(fb_list_kernel, dim3(fgrid), dim3(256), 0, stream, fb_flag, offs, (uint32_t)n_fb, f.groups, work.as<uint32_t>());
  CK(hipGetLastError());
  const uint32_t* n_work = fb_flag + n_fb;
  const uint32_t dbg = getenv("MIRGE_COLLAPSE_DBG") ? (uint32_t)atoi(getenv("MIRGE_COLLAPSE_DBG")) : 0u;
  if (dbg & 8u) {
    uint32_t h[2] = {0, 0};
    std::vector<uint32_t> ho(n_ht + 1);
    (void)hipMemcpyAsync(ho.data(), offs, (n_ht + 1) * 4, hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    uint32_t mx = 0, big = 0;
    for (size_t fb = 0; fb < n_fb; ++fb) {
      const uint32_t c = ho[(fb + 1) * f.groups] - ho[fb * f.groups];
      mx = std::max(mx, c);
      big += c > 4096u ? 1u : 0u;
    }
    fprintf(stderr, "collapse_fast: largest final bucket %u pairs, %u buckets above 4096\n", mx, big);
    (void)hipMemcpyAsync(&h[0], offs + n_ht, 4, hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(&h[1], n_work, 4, hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    fprintf(stderr, "collapse_fast: %u reads -> %u pairs in %u final buckets (%u chunks, %u L1 bins, %u groups)\n", n, h[0], h[1],
            f.n_chunks, f.n_bins, f.groups);
  }
  const uint32_t red_grid = (uint32_t)std::min<uint64_t>(n_fb, (uint64_t)std::max(1, n_cu) * 12u);
  hipLaunchKernelGGL(reduce_kernel, dim3(red_grid), dim3(kRedThreads), 0, stream, f, work.as<uint32_t>(), n_work, offs, l1_b2,
                     bufB.as<uint64_t>(), bufA.as<uint64_t>(), cnt.as<uint32_t>(), fb_entries, fb_reads, overflow);
  CK(hipGetLastError());
  CK(prims::exclusive_sum_u32(fb_reads, read_base, n_fb + 1, stmp.p, stream));
  uint32_t h_over = 0, h_unique = 0;
  CK(hipMemcpyAsync(&h_over, overflow, 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&h_unique, read_base + n_fb, 4, hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  if (h_over) return hipSuccess;  // (general path)
  const uint64_t n_unique = h_unique;
  if (n_unique > cap) return hipErrorInvalidValue;
  if (n_samples > 1) CK(hipMemsetAsync(d_quant, 0, (size_t)n_unique * n_samples * 4, stream));
  hipLaunchKernelGGL(emit_fast_kernel, dim3(red_grid), dim3(kRedThreads), 0, stream, f, work.as<uint32_t>(), n_work, offs,
                     fb_entries, read_base, bufA.as<uint64_t>(), cnt.as<uint32_t>(), n_samples, d_u_words, d_u_lens, d_quant);
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
  // (the fast path loads reads 16 bytes, lengths 4 and sample ids 8 bytes at a time: a caller's sliced arrays -- d_lens + 1 --
  // take the general path: advisor, round 5)
  const bool aligned = ((uintptr_t)d_reads % 16u == 0u) && ((uintptr_t)d_lens % 4u == 0u) && (!smp || (uintptr_t)smp % 8u == 0u);
  const bool try_fast = W == 1 && !d_nmask && allow_fast && n_samples <= 16u && aligned;
  if (try_fast) {
    // one pass for the histogram, the sample bound and the raw counts the fast path partitions by (K0)
    const size_t mark = arena.used;
    FastPlan plan;
    DevBuf stmp;
    CK(fast_prepass(d_reads, d_lens, smp, n, n_samples, d_len_hist, stream, &plan, &stmp));
    uint32_t h_ms = 0;
    CK(hipMemcpyAsync(&h_ms, plan.max_sample.p, 4, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(h_hist.data(), d_len_hist, h_hist.size() * 8, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    if (smp && h_ms >= n_samples) return hipErrorInvalidDevicePointer;  // (mapped to MRG_ERR_ARG by the C-ABI)
    bool took = false;
    CK(collapse_fast(d_reads, d_lens, smp, n, n_samples, h_hist, plan, cap, d_u_words, d_u_lens, d_quant, h_n_unique, n_cu, stream, &took));
    if (took) return hipSuccess;
    arena.used = mark;
  } else {
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
    auto* hist = reinterpret_cast<unsigned long long*>(d_len_hist);
    const uint32_t lds = 256u * n_samples * 4u;
    if (lds <= 48u * 1024u)
      hipLaunchKernelGGL(length_hist_kernel<true>, dim3(min(grid, 1024u)), dim3(kT), lds, stream, d_lens, smp, n, n_samples, hist);
    else
      hipLaunchKernelGGL(length_hist_kernel<false>, dim3(min(grid, 1024u)), dim3(kT), 0, stream, d_lens, smp, n, n_samples, hist);
    CK(hipGetLastError());
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
