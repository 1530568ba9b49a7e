// gfx950 (CDNA4, wave64) dictionary kernels of the annotation cascade: batches of one-word reads
// (<= 32 nt) without N, i.e. every trimmed small-RNA read set.
//
// What they replace in the reference: the same bowtie runs of runAnnotationPipeline.py:577-599 /
// :688 as kernels.hip.  For a read no longer than bowtie's seed (28 nt) every policy of the
// cascade is a statement about the WHOLE read -- `-n 0` / `-v 0`: it occurs letter for letter in
// an entry; `-n 1` / `-v 1`: with one substitution -- so the search is a dictionary problem:
//   exact_dict_kernel   one slot load of the library's exact-match dictionary per read
// Integer/index work: no MFMA.  The budget that matters is VALU issue per read and the number of
// dependent random loads per read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_util.hpp"
#include "dict_index.hpp"
#include "kernels.hpp"

namespace mrg {

namespace {

using namespace dev;

constexpr uint32_t kBlock = 1024u;

// Longest input segment of a segmented survivor list -> ctl[1] (all threads call; two barriers)
__device__ __forceinline__ void longest_segment(const uint32_t* in_count, uint32_t in_nseg, uint32_t* ctl1) {
  uint32_t mx = 0;
  for (uint32_t sgi = threadIdx.x; sgi < in_nseg; sgi += blockDim.x) mx = max(mx, in_count[sgi]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(ctl1, mx);
}

// The FM-index answer for one read of a pass WITHOUT seed mismatches: rows of the prefix interval
// of the largest jump table the seed region is long enough for, each verified against the text.
// Slow (a short prefix has many rows) and rare: reads shorter than the dictionary key, and keys
// of an overflowed chain.  Returns the best (mm << 32 | text position) or ~0.
__device__ __forceinline__ uint64_t fm_exact_fallback(const ExactParams& p, uint64_t rd, int32_t L, uint32_t& best_seg,
                                                   uint32_t& best_before, uint32_t& n_rows) {
  const int32_t R = min(L, p.seed_len);
  uint32_t tab_off = 0;
  const uint32_t k = p.tabs.k[0] ? pick_table(p.tabs, R, tab_off) : 0u;
  uint32_t lo = 0, hi = p.n + 1;
  if (k) {
    const uint32_t* tab = p.ftab + tab_off + lex_code(rd & low_bits(2 * k), k);
    lo = tab[0];
    hi = tab[1];
  }
  n_rows = hi > lo ? hi - lo : 0u;
  uint64_t best = ~0ull;
  const uint64_t lmask = low_bits(2 * (uint32_t)L), smask = low_bits(2 * (uint32_t)R);
  for (uint32_t i = lo; i < hi; ++i) {
    const uint64_t row = p.sa[i];
    const uint32_t before = (uint32_t)(row >> 32) & 255u, after = (uint32_t)(row >> 40) & 255u;
    if ((uint32_t)L > after) continue;
    const uint32_t s = (uint32_t)row;
    const uint64_t m = mismatch_bits(text_window(p.text, s), rd) & lmask;
    if ((m & smask) != 0ull) continue;
    const uint32_t mm_total = (uint32_t)__popcll(m);
    if ((int32_t)mm_total > p.max_mm_total) continue;
    const uint64_t key = ((uint64_t)mm_total << 32) | s;
    if (key < best) {
      best = key;
      best_seg = (uint32_t)(row >> 48);
      best_before = before;
    }
  }
  return best;
}

}  // namespace

// U reads per lane and chunk: the slot loads of a lane's reads are in flight together (one read per
// lane and trip is latency-bound: two dependent memory trips per 2048 reads of a CU).
// FIRST: the pass streams the whole read set (identity list, 16-byte aligned arrays): a lane takes U
// consecutive reads with 16-byte loads and writes EVERY output of its reads with wide stores -- the
// "unannotated" values for the reads it does not claim, which whatever pass claims them later
// overwrites.  KBITS: the library's 9-mer presence bitmap is staged in LDS and a read whose first or
// last 9 bases do not occur in the library never asks for its slot.
constexpr int kExactU = 4;
static_assert(kBlock * kExactU == kExactChunk, "kernels.hpp: kExactChunk");
typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));
typedef int int4_t __attribute__((ext_vector_type(4)));

template <bool FIRST, bool KBITS>
__global__ void __launch_bounds__(kBlock, 8) exact_dict_kernel(const ExactParams p) {
  constexpr int U = kExactU;
  __shared__ uint32_t ctl[2];  // [0] survivors appended by this workgroup, [1] longest input segment
  __shared__ unsigned long long wg_cnt[5];
  __shared__ __attribute__((aligned(16))) uint32_t skbits[KBITS ? kKmerBitsWords : 4];
  if (threadIdx.x == 0) {
    ctl[0] = 0u;
    ctl[1] = 0u;
  }
  if (threadIdx.x < 5) wg_cnt[threadIdx.x] = 0ull;
  if (KBITS) {
    const uint4* src = reinterpret_cast<const uint4*>(p.kbits);
    uint4* dst = reinterpret_cast<uint4*>(skbits);
    for (uint32_t i = threadIdx.x; i < kKmerBitsWords / 4; i += kBlock) dst[i] = src[i];
  }
  __syncthreads();
  if (!FIRST && p.idx_in) longest_segment(p.in_count, p.in_nseg, &ctl[1]);
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63;
  uint32_t c_processed = 0, c_aligned = 0, c_cands = 0, c_lookups = 0;
  const uint32_t smask = (1u << p.log2_slots) - 1u, hshift = 32u - p.log2_slots;
  const uint32_t kmask = p.key_bases >= 16u ? 0xFFFFFFFFu : ((1u << (2u * p.key_bases)) - 1u);
  SegTables segs{p.seg_start, p.seg_ref, p.seg_off, p.chunk_seg, p.simple_segs};
  const bool have_list = !FIRST && p.idx_in;

  constexpr uint32_t kChunk = kBlock * U;
  const uint32_t in_nseg = have_list ? p.in_nseg : 1u;
  const uint32_t depth_chunks = have_list ? (ctl[1] + kChunk - 1) / kChunk : (p.n_total + kChunk - 1) / kChunk;
  const uint32_t n_chunks = in_nseg * depth_chunks;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t sgi = chunk % in_nseg, depth = chunk / in_nseg;
    const uint32_t t0 = depth * kChunk + threadIdx.x * U;
    const uint32_t count = have_list ? p.in_count[sgi] : p.n_total;
    const bool full = t0 + U <= count;
    uint32_t r[U], L0[U];
    uint64_t rd[U];
    bool active[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      active[u] = t0 + u < count;
      r[u] = t0 + u;
      L0[u] = 0;
      rd[u] = 0;
    }
    if (FIRST && full) {
      const ull2_t a = *reinterpret_cast<const ull2_t*>(p.reads + t0);
      const ull2_t b = *reinterpret_cast<const ull2_t*>(p.reads + t0 + 2);
      const uint32_t l4 = *reinterpret_cast<const uint32_t*>(p.lens + t0);
      rd[0] = a[0];
      rd[1] = a[1];
      rd[2] = b[0];
      rd[3] = b[1];
#pragma unroll
      for (int u = 0; u < U; ++u) L0[u] = (l4 >> (8 * u)) & 255u;
    } else {
      if (have_list) {
        const uint32_t* lp = p.idx_in + (size_t)sgi * p.in_seg_cap + t0;
        if (full) {
          const uint4 q = *reinterpret_cast<const uint4*>(lp);
          r[0] = q.x;
          r[1] = q.y;
          r[2] = q.z;
          r[3] = q.w;
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (active[u]) r[u] = lp[u];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (active[u]) {
          L0[u] = p.lens[r[u]];
          rd[u] = p.reads[r[u]];
        }
    }
    // ---- which reads this pass's FASTA would contain (RAP:543-554, 664-686), and their slots ----
    int32_t L[U];
    bool eligible[U], search[U], fallback[U];
    uint32_t home[U];
    uint4 s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      eligible[u] = active[u] && (int32_t)L0[u] >= p.min_len && (int32_t)L0[u] <= p.max_len;
      L[u] = (int32_t)L0[u];
      if (p.poly_t) {
        const int32_t tail = trailing_t(rd[u], L[u]);
        eligible[u] = eligible[u] && tail >= 3 && (L[u] - tail) >= 11;
        L[u] -= tail;
      }
      L[u] -= p.trim5 + p.trim3;
      rd[u] >>= 2 * p.trim5;
      if (eligible[u]) ++c_processed;
      search[u] = eligible[u] && L[u] > 0;
      fallback[u] = search[u] && (uint32_t)L[u] < p.key_bases;
      search[u] = search[u] && !fallback[u];
      if (KBITS && search[u]) {
        // (key_bases >= 9: a searched read has both 9-mers; the second one ends where the part of
        // the read that must match exactly ends)
        const uint32_t c0 = (uint32_t)rd[u] & ((1u << (2u * kKmerBitsK)) - 1u);
        const uint32_t c1 = (uint32_t)(rd[u] >> (2u * ((uint32_t)min(L[u], p.seed_len) - kKmerBitsK))) & ((1u << (2u * kKmerBitsK)) - 1u);
        search[u] = ((skbits[c0 >> 5] >> (c0 & 31u)) & (skbits[c1 >> 5] >> (c1 & 31u)) & 1u) != 0u;
      }
      home[u] = (((uint32_t)rd[u] & kmask) * kDictHashMul) >> hshift;
      s[u] = make_uint4(0u, 0u, 0u, 0u);
      if (search[u]) {
        s[u] = p.slots[home[u]];
        ++c_lookups;
      }
    }
    bool aligned[U];
    uint32_t o_ref[U], o_pos[U], o_mm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      aligned[u] = false;
      o_ref[u] = o_pos[u] = o_mm[u] = 0u;
      if (search[u]) {
        const uint64_t lmask = low_bits(2 * (uint32_t)L[u]);
        const uint32_t chain = (s[u].w >> kDictChainShift) & kDictChainMask;
        uint4 sl = s[u];
        if (chain == kDictChainOverflow) {
          fallback[u] = true;
        } else if (L[u] <= p.seed_len) {
          // the whole read lies in the seed: exact or nothing; the first match is the lowest (entry, offset)
          for (uint32_t j = 0;; ++j) {
            const uint64_t win = (uint64_t)sl.x | ((uint64_t)sl.y << 32);
            ++c_cands;
            if ((sl.w & kDictOccBit) && ((win ^ rd[u]) & lmask) == 0ull && (uint32_t)L[u] <= (sl.w & kDictAfterMask)) {
              aligned[u] = true;
              o_ref[u] = sl.z;
              o_pos[u] = sl.w >> kDictOffShift;
              break;
            }
            if (j >= chain) break;
            sl = p.slots[(home[u] + j + 1u) & smask];
          }
        } else {
          // `-n 0` on a read longer than the seed: exact inside it, up to max_mm_total mismatches behind
          const uint64_t seedmask = low_bits(2 * (uint32_t)p.seed_len);
          uint32_t best = ~0u;
          for (uint32_t j = 0;; ++j) {
            const uint64_t win = (uint64_t)sl.x | ((uint64_t)sl.y << 32);
            ++c_cands;
            const uint64_t m = mismatch_bits(win, rd[u]) & lmask;
            const uint32_t mmt = (uint32_t)__popcll(m);
            if ((sl.w & kDictOccBit) && (m & seedmask) == 0ull && (int32_t)mmt <= p.max_mm_total &&
                (uint32_t)L[u] <= (sl.w & kDictAfterMask) && ((mmt << 8) | j) < best) {
              best = (mmt << 8) | j;
              aligned[u] = true;
              o_ref[u] = sl.z;
              o_pos[u] = sl.w >> kDictOffShift;
              o_mm[u] = mmt;
            }
            if (j >= chain || (best >> 8) == 0u) break;
            sl = p.slots[(home[u] + j + 1u) & smask];
          }
        }
      }
    }
    bool any_fb = false;
#pragma unroll
    for (int u = 0; u < U; ++u) any_fb |= fallback[u];
    if (__builtin_expect(__any(any_fb), 0)) {
      for (int u = 0; u < U; ++u) {  // (not unrolled: rare)
        if (!fallback[u]) continue;
        uint32_t bseg = 0xFFFFu, bbefore = 255u, rows = 0;
        const uint64_t best = fm_exact_fallback(p, rd[u], L[u], bseg, bbefore, rows);
        ++c_lookups;
        c_cands += rows;
        if (best != ~0ull) {
          aligned[u] = true;
          o_mm[u] = (uint32_t)(best >> 32);
          locate_entry(segs, (uint32_t)best, bseg, bbefore, o_ref[u], o_pos[u]);
        }
      }
    }

    // ---- outputs ----
    uint32_t n_surv = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c_aligned += aligned[u] ? 1u : 0u;
      n_surv += (active[u] && !aligned[u]) ? 1u : 0u;
    }
    if (FIRST && full) {
      uint32_t pid = 0, mmv = 0;
      int4_t refs, poss;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pid |= (aligned[u] ? (uint32_t)p.pass_index : 0xFFu) << (8 * u);
        mmv |= (aligned[u] ? o_mm[u] : 0u) << (8 * u);
        refs[u] = aligned[u] ? (int32_t)o_ref[u] : -1;
        poss[u] = aligned[u] ? (int32_t)o_pos[u] : -1;
      }
      *reinterpret_cast<uint32_t*>(p.pass_id + t0) = pid;
      *reinterpret_cast<uint32_t*>(p.mm + t0) = mmv;
      *reinterpret_cast<int4_t*>(p.ref_id + t0) = refs;
      *reinterpret_cast<int4_t*>(p.pos + t0) = poss;
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (aligned[u]) {
          p.pass_id[r[u]] = (int8_t)p.pass_index;
          p.ref_id[r[u]] = (int32_t)o_ref[u];
          p.pos[r[u]] = (int32_t)o_pos[u];
          p.mm[r[u]] = (uint8_t)o_mm[u];
        } else if (active[u] && (FIRST || !p.idx_out)) {
          // first pass: every output is written; last pass: whatever is still unclaimed stays unannotated
          p.pass_id[r[u]] = (int8_t)-1;
          p.ref_id[r[u]] = -1;
          p.pos[r[u]] = -1;
          p.mm[r[u]] = 0;
        }
      }
    }
    // ---- survivors feed the next pass: the workgroup's own list segment, in read order; one LDS
    // atomic per wave ----
    if (p.idx_out) {
      const uint32_t incl = wave_incl_scan(n_surv);
      const uint32_t total = __shfl(incl, 63, 64);
      if (total) {
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&ctl[0], total);
        wbase = __shfl(wbase, 0, 64);
        uint32_t* dst = p.idx_out + (size_t)blockIdx.x * p.out_seg_cap + wbase + (incl - n_surv);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (active[u] && !aligned[u]) *dst++ = r[u];
      }
    }
  }
  // counters: wave -> workgroup (LDS) -> one global atomic per counter and workgroup
  const uint64_t t_processed = wave_sum(c_processed), t_aligned = wave_sum(c_aligned);
  const uint64_t t_cands = wave_sum(c_cands), t_lookups = wave_sum(c_lookups);
  if (lane == 0) {
    if (t_processed) atomicAdd(&wg_cnt[0], (unsigned long long)t_processed);
    if (t_aligned) atomicAdd(&wg_cnt[1], (unsigned long long)t_aligned);
    if (t_cands) atomicAdd(&wg_cnt[3], (unsigned long long)t_cands);
    if (t_lookups) atomicAdd(&wg_cnt[4], (unsigned long long)t_lookups);
  }
  __syncthreads();
  if (threadIdx.x < 5 && wg_cnt[threadIdx.x]) atomicAdd((unsigned long long*)&p.counters[threadIdx.x], wg_cnt[threadIdx.x]);
  if (p.idx_out && threadIdx.x == 0) p.out_count[blockIdx.x] = ctl[0];
}

hipError_t launch_exact_dict(const ExactParams& p, uint32_t grid, hipStream_t stream) {
  // the streaming instantiation needs the identity list and arrays it can address 16 bytes at a time
  const bool first = !p.idx_in && ((uintptr_t)p.reads % 16 == 0) && ((uintptr_t)p.lens % 4 == 0) && ((uintptr_t)p.pass_id % 4 == 0) &&
                     ((uintptr_t)p.mm % 4 == 0) && ((uintptr_t)p.ref_id % 16 == 0) && ((uintptr_t)p.pos % 16 == 0);
  const bool kb = p.kbits != nullptr && p.key_bases >= kKmerBitsK;
  if (first && kb) hipLaunchKernelGGL((exact_dict_kernel<true, true>), dim3(grid), dim3(kBlock), 0, stream, p);
  else if (first) hipLaunchKernelGGL((exact_dict_kernel<true, false>), dim3(grid), dim3(kBlock), 0, stream, p);
  else if (kb) hipLaunchKernelGGL((exact_dict_kernel<false, true>), dim3(grid), dim3(kBlock), 0, stream, p);
  else hipLaunchKernelGGL((exact_dict_kernel<false, false>), dim3(grid), dim3(kBlock), 0, stream, p);
  return hipGetLastError();
}

}  // namespace mrg
