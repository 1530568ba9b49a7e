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
//                  11 bits of the remaining key", linear probing without wrap-around.  Runs of occupied slots are then
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
constexpr uint32_t kAggSlots = 4096u;        // K1's LDS table
constexpr uint32_t kAggFlushAt = 2048u;      // ... flushed once it holds this many distinct keys
constexpr uint32_t kAggKeep = 1024u;         // ... of which about this many (the most frequent) stay
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
                                                               unsigned long long* __restrict__ len_hist, uint32_t* __restrict__ max_sample) {
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
  for (uint32_t b = threadIdx.x; b < kLenBins; b += kFastThreads) counts_t[(size_t)b * n_chunks + blockIdx.x] = hist[b];
  if (LENH)
    for (uint32_t b = threadIdx.x; b < 256u * n_samples; b += kFastThreads)
      if (lhist[b]) atomicAdd(&len_hist[b], (unsigned long long)lhist[b]);
  if (sample) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ms = max(ms, (uint32_t)__shfl_down(ms, off, 64));
    if ((threadIdx.x & 63u) == 0u && ms) atomicMax(max_sample, ms);
  }
}

// K1: aggregate a chunk through an LDS hash table, append (rest, count) pairs to the chunk's L1 regions.
// The table is flushed when it holds more than kAggFlushAt distinct keys -- but only its COLD entries leave: the
// entries seen at least T times since they entered stay (T doubles while more than kAggKeep would stay, halves when
// few do), so a sequence that is a percent of the sample leaves the chunk once, at the end, with its whole count
// instead of once per flush.  (Emptied slots are not tombstoned: a key whose probe chain was cut is inserted a
// second time, the two entries are summed by K3 like any two pairs of one key.)  A region is sized for the chunk's
// raw reads of the bucket: what the pairs leave of it is filled with zero pairs, so that K2 streams whole buckets.
// off_t: exclusive prefix of K0's counts (indexed by length x 256 + top, one entry past the end).
__global__ void __launch_bounds__(kFastThreads) aggregate_kernel(const uint64_t* __restrict__ words, const uint8_t* __restrict__ lens,
                                                                 const uint16_t* __restrict__ sample, FastShape f,
                                                                 const uint32_t* __restrict__ off_t, uint32_t* __restrict__ bin_pairs,
                                                                 uint64_t* __restrict__ pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* tkey = reinterpret_cast<unsigned long long*>(smem);
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(smem + kAggSlots * 8u);
  uint32_t* cursor = tcnt + kAggSlots;  // n_bins: where the next pair of a bucket goes
  uint32_t* ctl = cursor + f.n_bins;    // [0] distinct keys in the table, [1] entries that stay, [2] T
  for (uint32_t s = threadIdx.x; s < kAggSlots; s += kFastThreads) {
    tkey[s] = kEmpty;
    tcnt[s] = 0u;
  }
  for (uint32_t b = threadIdx.x; b < f.n_bins; b += kFastThreads)
    cursor[b] = off_t[((size_t)f.len_of_slot[b >> 8] * 256u + (b & 255u)) * f.n_chunks + blockIdx.x];
  if (threadIdx.x == 0) ctl[0] = 0u, ctl[1] = 0u, ctl[2] = 2u;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * f.chunk;
  const uint64_t hi = min((uint64_t)f.n, lo + f.chunk);
  const uint32_t lane = threadIdx.x & 63u;
  // the next batch's read is requested before this batch's goes into the table
  uint64_t nw = 0;
  uint32_t nl = 0, ns = 0;
  if (lo + threadIdx.x < hi) {
    nw = words[lo + threadIdx.x];
    nl = lens[lo + threadIdx.x];
    if (sample) ns = sample[lo + threadIdx.x];
  }
  for (uint64_t base = lo; base < hi; base += kFastThreads) {
    const uint64_t i = base + threadIdx.x;
    const uint64_t w = nw;
    const uint32_t L = nl, smp = ns;
    if (i + kFastThreads < hi) {
      nw = words[i + kFastThreads];
      nl = lens[i + kFastThreads];
      if (sample) ns = sample[i + kFastThreads];
    }
    bool fresh = false;
    if (i < hi) {
      const uint64_t v = (w << f.sb) | (uint64_t)smp;
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
    __syncthreads();
    const bool last = base + kFastThreads >= hi;
    if (ctl[0] > kAggFlushAt || last) {
      for (;;) {
        const uint32_t T = last ? 0xFFFFFFFFu : ctl[2];
        // every entry seen fewer than T times becomes a pair in its L1 bucket's region
        for (uint32_t s = threadIdx.x; s < kAggSlots; s += kFastThreads) {
          const unsigned long long key = tkey[s];
          uint32_t c = key != kEmpty ? tcnt[s] : 0u;
          const bool stays = c >= T;
          const uint64_t sm_ = __ballot(stays);
          if (lane == 0 && sm_) atomicAdd(&ctl[1], (uint32_t)__popcll(sm_));
          if (c == 0u || stays) continue;
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
          tkey[s] = kEmpty;
          tcnt[s] = 0u;
        }
        __syncthreads();
        const uint32_t kept = ctl[1];
        __syncthreads();
        if (threadIdx.x == 0) {
          ctl[0] = kept;
          ctl[1] = 0u;
          if (kept > kAggKeep) ctl[2] = T * 2u;                       // too many stay: a higher bar from now on
          else if (kept < kAggKeep / 4u && T > 2u) ctl[2] = T / 2u;   // the table runs cold: a lower one
        }
        __syncthreads();
        if (last || kept <= kAggKeep + kAggKeep / 2u) break;  // (else: sweep again with the higher bar)
      }
    }
  }
  // unused positions of the chunk's regions -> zero pairs; pairs of the chunk per bucket -> the buckets' totals
  const uint32_t wave = threadIdx.x >> 6;
  for (uint32_t b = wave; b < f.n_bins; b += kFastThreads / 64u) {
    const size_t at = ((size_t)f.len_of_slot[b >> 8] * 256u + (b & 255u)) * f.n_chunks + blockIdx.x;
    const uint32_t first = off_t[at], end = off_t[at + 1], cur = cursor[b];
    for (uint32_t j = cur + lane; j < end; j += 64u) pairs[j] = 0ull;
    if (lane == 0 && cur > first) atomicAdd(&bin_pairs[b], cur - first);
  }
}

// K2a / K2b: the pairs of an L1 bucket (its whole region: zero pairs skipped) counted by their next b2 bits, then copied
// into contiguous final buckets.  A workgroup = (L1 bucket, one of `groups` equal parts of its region); the counts are
// stored [final bucket][group], so ONE prefix sum over the array places every (final bucket, group) run: final buckets
// in key order, dense.  b2 = enough bits for ~1024 pairs per final bucket.
__device__ __forceinline__ uint32_t sub_bits_of(uint32_t P, uint32_t r1) {
  uint32_t b2 = 0;
  while (b2 < 8u && b2 < r1 && ((P + 1023u) >> 10) > (1u << b2)) ++b2;
  return b2;
}

template <bool SCATTER>
__global__ void __launch_bounds__(kFastThreads) subdivide_kernel(FastShape f, const uint32_t* __restrict__ off_t,
                                                                 const uint32_t* __restrict__ bin_pairs, const uint64_t* __restrict__ pairs_in,
                                                                 uint32_t* __restrict__ hist_t /* SCATTER: its exclusive prefix */,
                                                                 uint64_t* __restrict__ pairs_out, uint8_t* __restrict__ l1_b2) {
  __shared__ uint32_t sub[256];  // counts, or (SCATTER) write cursors
  const uint32_t b = blockIdx.x / f.groups, g = blockIdx.x % f.groups;
  const uint32_t tid = threadIdx.x;
  const uint32_t P = bin_pairs[b];
  if (P == 0) return;
  if (tid < 256u) sub[tid] = SCATTER ? hist_t[((size_t)b * 256u + tid) * f.groups + g] : 0u;
  __syncthreads();
  const uint32_t L = f.len_of_slot[b >> 8];
  const uint32_t r1 = l1_shift(L) + f.sb;
  const uint32_t b2 = sub_bits_of(P, r1);
  const uint32_t sh = r1 - b2;
  if (!SCATTER && g == 0 && tid == 0) l1_b2[b] = (uint8_t)b2;
  const size_t at = ((size_t)L * 256u + (b & 255u)) * f.n_chunks;
  const uint32_t r_lo = off_t[at], r_hi = off_t[at + f.n_chunks];
  const uint32_t a0 = r_lo & ~1u;                                              // (the pair in front of an odd start is another bucket's)
  const uint32_t span = (((r_hi - a0) + f.groups - 1u) / f.groups + 1u) & ~1u;  // (even: 16-byte loads stay aligned)
  const uint32_t lo = a0 + g * span, hi = min(r_hi, lo + span);
  constexpr uint32_t kUnroll = 4u;  // 16-byte loads a lane keeps in flight
  for (uint32_t i0 = lo + 2u * tid; i0 < hi; i0 += 2u * kFastThreads * kUnroll) {
    uint4 q[kUnroll];
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u) {
      const uint32_t i = i0 + u * 2u * kFastThreads;
      q[u] = i < hi ? *reinterpret_cast<const uint4*>(pairs_in + i) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u) {
      const uint32_t i = i0 + u * 2u * kFastThreads;
      const uint64_t p0 = (uint64_t)q[u].x | ((uint64_t)q[u].y << 32), p1 = (uint64_t)q[u].z | ((uint64_t)q[u].w << 32);
      if (i >= r_lo && i < hi && (p0 >> kCountShift)) {
        const uint32_t at0 = atomicAdd(&sub[(uint32_t)((p0 & kRestMask) >> sh) & 255u], 1u);
        if (SCATTER) pairs_out[at0] = p0;
      }
      if (i + 1u < hi && (p1 >> kCountShift)) {
        const uint32_t at1 = atomicAdd(&sub[(uint32_t)((p1 & kRestMask) >> sh) & 255u], 1u);
        if (SCATTER) pairs_out[at1] = p1;
      }
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
  __shared__ unsigned long long tkey[kRedSlots];   // the table, then the compacted entries
  __shared__ uint32_t tcnt[kRedSlots];
  __shared__ unsigned long long skey[kRedSlots];   // the entries grouped by bin
  __shared__ uint32_t scnt[kRedSlots];
  __shared__ uint32_t bin_cnt[256], bin_start[257];
  __shared__ uint32_t wtot[2][kRedThreads / 64u];
  __shared__ uint32_t n_in;
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
    if (tid == 0) n_in = 0u;
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
        for (;;) {
          const unsigned long long old = atomicCAS(&tkey[s], kEmpty, rest);
          if (old == kEmpty || old == rest) {
            atomicAdd(&tcnt[s], (uint32_t)(pr[u] >> kCountShift));
            if (old == kEmpty && atomicAdd(&n_in, 1u) >= kRedSlots - kRedSlots / 4u) lost = true;  // (three quarters full: not this table's bucket)
            break;
          }
          s = (s + 1u) & (kRedSlots - 1u);
          if (n_in >= kRedSlots - kRedSlots / 4u) {  // (never spin in a full table)
            lost = true;
            break;
          }
        }
      }
    }
    if (lost) atomicOr(overflow, 1u);
    __syncthreads();
    // the occupied slots, compacted in place (every thread holds its slots in registers across the barrier)
    const uint32_t s_lo = tid * kPerThread;
    unsigned long long mk[kPerThread];
    uint32_t mc[kPerThread];
    uint32_t n_mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k) {
      mk[k] = tkey[s_lo + k];
      mc[k] = tcnt[s_lo + k];
      n_mine += mk[k] != kEmpty ? 1u : 0u;
    }
    const uint32_t ie = dev::wave_incl_scan(n_mine);
    if (lane == 63u) wtot[0][wave] = ie;
    __syncthreads();
    uint32_t pos = ie - n_mine, D = 0;
#pragma unroll
    for (uint32_t w = 0; w < kRedThreads / 64u; ++w) {
      pos += w < wave ? wtot[0][w] : 0u;
      D += wtot[0][w];
    }
#pragma unroll
    for (uint32_t k = 0; k < kPerThread; ++k)
      if (mk[k] != kEmpty) {
        tkey[pos] = mk[k];
        tcnt[pos] = mc[k];
        ++pos;
      }
    __syncthreads();
    // counting sort of the entries by bin = the top 8 bits of the key's open part (monotone in the key)
    const uint32_t bsh = r2 > 8u ? r2 - 8u : 0u;
    const uint64_t open_mask = (1ull << r2) - 1ull;
    for (uint32_t e = tid; e < D; e += kRedThreads) atomicAdd(&bin_cnt[(uint32_t)((tkey[e] & open_mask) >> bsh) & 255u], 1u);
    __syncthreads();
    {
      const uint32_t c = bin_cnt[tid];
      const uint32_t incl = dev::wave_incl_scan(c);
      if (lane == 63u) wtot[1][wave] = incl;
      __syncthreads();
      uint32_t st = incl - c;
#pragma unroll
      for (uint32_t w = 0; w < kRedThreads / 64u; ++w) st += w < wave ? wtot[1][w] : 0u;
      bin_start[tid] = st;
      if (tid == 255u) bin_start[256] = st + c;
      bin_cnt[tid] = st;  // (now the bin's write cursor)
    }
    __syncthreads();
    for (uint32_t e = tid; e < D; e += kRedThreads) {
      const unsigned long long k = tkey[e];
      const uint32_t at = atomicAdd(&bin_cnt[(uint32_t)((k & open_mask) >> bsh) & 255u], 1u);
      skey[at] = k;
      scnt[at] = tcnt[e];
    }
    __syncthreads();
    // an entry's place in its bin = members with a smaller key; it is the first entry of its read when no member with
    // a smaller key has the same bases (entries of one read differ in their sample bits only: same bin, or the key is
    // shorter than 8 + sample bits and the bins hold one key each)
    const uint64_t prefix = ((uint64_t)(l1 & 255u) << r1) | ((uint64_t)sub << r2);  // the key bits the bucket stands for
    uint32_t n_rd = 0;
    for (uint32_t e = tid; e < D; e += kRedThreads) {
      const unsigned long long k = skey[e];
      const uint32_t bn = (uint32_t)((k & open_mask) >> bsh) & 255u;
      const uint32_t b_lo = bin_start[bn], b_hi = bin_start[bn + 1u];
      uint32_t rank = 0;
      bool head = true;
      for (uint32_t j = b_lo; j < b_hi; ++j) {
        const unsigned long long o = skey[j];
        rank += o < k ? 1u : 0u;
        head = head && !(o < k && (o >> f.sb) == (k >> f.sb));
      }
      if (f.sb && bsh < f.sb && head) {
        // (tiny keys: the read's other samples may sit in the bins in front)
        for (uint32_t j = 0; j < b_lo; ++j) head = head && (skey[j] >> f.sb) != (k >> f.sb);
      }
      n_rd += head ? 1u : 0u;
      const uint32_t at = start + b_lo + rank;
      ent_key[at] = prefix | (k & open_mask);
      ent_cnt[at] = scnt[e];
    }
    const uint32_t ir = dev::wave_incl_scan(n_rd);
    if (lane == 63u) wtot[0][wave] = ir;
    __syncthreads();
    if (tid == 0) {
      fb_entries[fb] = D;
      fb_reads[fb] = wtot[0][0] + wtot[0][1] + wtot[0][2] + wtot[0][3];
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
  DevBuf max_sample;
};

// K0 (+ the prefix sum of its counts): issued before the host knows what the batch holds -- the length histogram it
// fills says that
hipError_t fast_prepass(const uint64_t* d_reads, const uint8_t* d_lens, const uint16_t* smp, uint32_t n, uint32_t n_samples,
                        uint64_t* d_len_hist, hipStream_t stream, FastPlan* plan, DevBuf* stmp) {
  plan->n_chunks = std::max<uint32_t>(1u, std::min<uint32_t>(kMaxChunks, (n + 8191u) / 8192u));
  plan->chunk = (uint32_t)((((uint64_t)n + plan->n_chunks - 1) / plan->n_chunks + 4095u) & ~4095ull);
  const size_t n_ct = (size_t)kLenBins * plan->n_chunks + 1;
  CK(plan->off_t.alloc(n_ct * 4));
  CK(plan->max_sample.alloc(4));
  CK(stmp->alloc(prims::scan_temp_bytes(n_ct)));
  CK(hipMemsetAsync(plan->max_sample.p, 0, 4, stream));
  CK(hipMemsetAsync(plan->off_t.as<uint32_t>() + (n_ct - 1), 0, 4, stream));
  const uint32_t lds = (kLenBins + 256u * n_samples) * 4u;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(prepass_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(prepass_kernel<true>, dim3(plan->n_chunks), dim3(kFastThreads), lds, stream, d_reads, d_lens, smp, n, plan->chunk,
                     plan->n_chunks, n_samples, plan->off_t.as<uint32_t>(), reinterpret_cast<unsigned long long*>(d_len_hist),
                     plan->max_sample.as<uint32_t>());
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
  f.groups = getenv("MIRGE_COLLAPSE_GROUPS") ? (uint32_t)std::max(1, atoi(getenv("MIRGE_COLLAPSE_GROUPS"))) : 8u;
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
  const uint32_t agg_lds = kAggSlots * 12u + f.n_bins * 4u + 64u;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggregate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)agg_lds));
  hipLaunchKernelGGL(aggregate_kernel, dim3(f.n_chunks), dim3(kFastThreads), agg_lds, stream, d_reads, d_lens, smp, f, off_t, bin_pairs,
                     bufA.as<uint64_t>());
  CK(hipGetLastError());
  const uint32_t sub_grid = f.n_bins * f.groups;
  uint32_t* offs = hist_t.as<uint32_t>();  // counts [final bucket][group] (+ one 0), then their exclusive prefix (+ the number of pairs)
  hipLaunchKernelGGL(subdivide_kernel<false>, dim3(sub_grid), dim3(kFastThreads), 0, stream, f, off_t, bin_pairs, bufA.as<uint64_t>(), offs,
                     (uint64_t*)nullptr, l1_b2);
  CK(hipGetLastError());
  CK(prims::exclusive_sum_u32(offs, offs, n_ht + 1, stmp.p, stream));
  hipLaunchKernelGGL(subdivide_kernel<true>, dim3(sub_grid), dim3(kFastThreads), 0, stream, f, off_t, bin_pairs, bufA.as<uint64_t>(), offs,
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
  const bool try_fast = W == 1 && !d_nmask && allow_fast && n_samples <= 16u;
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
