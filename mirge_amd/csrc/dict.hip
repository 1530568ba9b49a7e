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

#include <algorithm>
#include <type_traits>

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
// of an overflowed chain.  (The seed kernels' dictionary units of LARGE libraries -- where repeats make such keys
// common -- search backwards instead: device_util.hpp, fm_exact_search; that code in this kernel or in seed_kernel
// cost their streaming loops 2-6 % on the headline run, same-box A/B.)  Returns the best (mm << 32 | text position) or ~0.
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

// STRETCH (a streaming launch whose chunks do not go round: exact_dict_stretches): every workgroup takes ONE contiguous
// stretch of the batch instead of every gridDim-th chunk; <.., .., false> is the code of rounds 4-6, bit for bit.
template <bool FIRST, bool KBITS, bool STRETCH = false>
__global__ void __launch_bounds__(kBlock, 8) exact_dict_kernel(const ExactParams p) {
  static_assert(FIRST || !STRETCH, "stretches are the streaming instantiation's");
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
  // STRETCH: one contiguous stretch per workgroup (a multiple of four reads: the vector loads and stores stay aligned).
  // 10 M reads are 4.77 chunks per workgroup: by chunks, three quarters of the workgroups run a fifth trip while the rest
  // have left; by stretches every workgroup's last trip is three quarters full (configs[1]: 49.7 -> 47.7 us).  A batch
  // whose chunks go round evenly (100 M reads: 47.7 of 48) keeps the chunks: 512 streams 1.5 MB apart cost the headline's
  // pass 0 5 % (0.695 -> 0.73 ms), a sweep of neighbouring chunks does not -- and a run-time choice between the two in
  // ONE instantiation cost it 9 % (8 B more scratch): hence two.
  const uint32_t per = STRETCH ? (((p.n_total + gridDim.x - 1u) / gridDim.x) + 3u) & ~3u : 0u;
  const uint32_t my_lo = STRETCH ? (uint32_t)min((uint64_t)p.n_total, (uint64_t)blockIdx.x * per) : 0u;
  const uint32_t my_hi = STRETCH ? (uint32_t)min((uint64_t)p.n_total, (uint64_t)my_lo + per) : 0u;
  // (STRETCH: `chunk` is the trip's first read)
  const uint32_t c_first = STRETCH ? my_lo : blockIdx.x, c_step = STRETCH ? kChunk : gridDim.x, c_end = STRETCH ? my_hi : n_chunks;
  for (uint32_t chunk = c_first; chunk < c_end; chunk += c_step) {
    const uint32_t sgi = STRETCH ? 0u : chunk % in_nseg, depth = STRETCH ? 0u : chunk / in_nseg;
    const uint32_t t0 = (STRETCH ? chunk : depth * kChunk) + threadIdx.x * U;
    const uint32_t count = have_list ? p.in_count[sgi] : (STRETCH ? my_hi : p.n_total);
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
    // (the streaming instantiation's read indices are arithmetic: no registers held, or spilled, for them)
    auto rid = [&](int u) -> uint32_t { return FIRST ? t0 + (uint32_t)u : r[u]; };
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
      // (nothing but the load here: a counter bumped between the four slot loads was a spilled register
      // whose scratch reload waited with vmcnt(0) -- for the slot load just issued, one after the other)
      if (search[u]) s[u] = p.slots[home[u]];
    }
    bool aligned[U];
    uint32_t o_ref[U], o_pos[U], o_mm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c_lookups += search[u] ? 1u : 0u;
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
    if (p.packed) {
      // the one output array: 16 bytes per quartet instead of 40
      if (FIRST && full) {
        typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
        u4_t w;
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = aligned[u] ? pack_assignment(p.pass_index, o_ref[u], o_pos[u], o_mm[u]) : 0u;
        *reinterpret_cast<u4_t*>(p.packed + t0) = w;
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (aligned[u]) p.packed[rid(u)] = pack_assignment(p.pass_index, o_ref[u], o_pos[u], o_mm[u]);
          else if (active[u] && (FIRST || !p.idx_out)) p.packed[rid(u)] = 0u;
        }
      }
    } else if (FIRST && full) {
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
          p.pass_id[rid(u)] = (int8_t)p.pass_index;
          p.ref_id[rid(u)] = (int32_t)o_ref[u];
          p.pos[rid(u)] = (int32_t)o_pos[u];
          p.mm[rid(u)] = (uint8_t)o_mm[u];
        } else if (active[u] && (FIRST || !p.idx_out)) {
          // first pass: every output is written; last pass: whatever is still unclaimed stays unannotated
          p.pass_id[rid(u)] = (int8_t)-1;
          p.ref_id[rid(u)] = -1;
          p.pos[rid(u)] = -1;
          p.mm[rid(u)] = 0;
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
          if (active[u] && !aligned[u]) *dst++ = rid(u);
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

// ---------------------------------------------------------------------------
// seed_kernel (kernels.hpp): units of seed searches / dictionary probes over tiles of reads.
// ---------------------------------------------------------------------------
namespace {

constexpr uint32_t kUnitWords = 28u;  // LDS table entry of a unit (what a row's verification needs)
enum UnitWord : uint32_t { UW_SA16 = 0, UW_TEXT = 2, UW_SEGSTART = 4, UW_SEGREF = 6, UW_SEGOFF = 8, UW_CHUNKSEG = 10, UW_FLAGS = 12,
                           UW_LIMITS = 13, UW_MEMBERS = 14, UW_BUCKETS = 22, UW_SA = 24, UW_BPAIR = 26 };
constexpr uint32_t kRowFromBucket = 1u << 23;  // tag bit of a row-queue entry: x indexes the unit's buckets, not its wide rows
constexpr uint32_t kRowFromPair = 1u << 24;    // ... x indexes the unit's pair rows (wave_seed_kernel: 8-byte rows, the text decides)
constexpr uint32_t kSeedCtlWords = 16u;
static_assert(4u + 2u * kSeedMaxUnits <= kSeedCtlWords, "seed_kernel: two sets of item counters");
static_assert(2u * kSeedMaxUnits * kSeedMaxMembers <= 64u && kSeedMaxMembers <= 4u, "seed_kernel: one lane per member counter");
constexpr uint32_t kSeedCntSlots = 16u * 2u + kSeedMaxUnits * 2u;  // per pass processed / aligned, per unit candidates / lookups

struct SeedLds {
  unsigned long long* srd;   // [tile] packed read
  unsigned long long* best;  // [tile] best key
  uint2* rows;               // [row_cap] (row index, slot | unit << 11 | offset << 13 | k' << 19)
  uint4* wide;               // [kSeedWideCap] (lo, hi, slot | unit | offset | k', -)
  uint32_t* items;           // [n_units][item_cap] slot | seed << 11
  uint32_t* utab;            // [kSeedMaxUnits][kUnitWords]
  uint32_t* ctl;             // [0] survivors, [1] longest input segment, [2] rows, [3] wide, [4 + 3 parity + u] items of unit u (by tile parity)
  unsigned long long* cnt;   // [kSeedCntSlots]
  uint8_t* sL0;              // [tile] read length, 255 = no read
  unsigned long long* srh;   // [tile] second word of the read (LONG instantiations only: behind everything else)
};

__device__ __forceinline__ SeedLds carve_seed_lds(uint32_t* smem, uint32_t tile, uint32_t row_cap, uint32_t item_cap, uint32_t n_units) {
  SeedLds l;
  l.srd = reinterpret_cast<unsigned long long*>(smem);
  l.best = l.srd + tile;
  l.rows = reinterpret_cast<uint2*>(l.best + tile);
  l.wide = reinterpret_cast<uint4*>(l.rows + row_cap);
  l.items = reinterpret_cast<uint32_t*>(l.wide + kSeedWideCap);
  l.utab = l.items + n_units * item_cap;
  l.ctl = l.utab + kSeedMaxUnits * kUnitWords;
  l.cnt = reinterpret_cast<unsigned long long*>(l.ctl + kSeedCtlWords);
  l.sL0 = reinterpret_cast<uint8_t*>(l.cnt + kSeedCntSlots);
  l.srh = reinterpret_cast<unsigned long long*>(l.sL0 + ((tile + 7u) & ~7u));
  return l;
}

// the read as one unit's bowtie run sees it: length window, poly-T rule, trims
__device__ __forceinline__ bool unit_view(int32_t min_len, int32_t max_len, int32_t poly_t, int32_t trim5, int32_t trim3, uint64_t rd,
                                          uint32_t L0, uint64_t& q, int32_t& L) {
  bool el = (int32_t)L0 >= min_len && (int32_t)L0 <= max_len;
  L = (int32_t)L0;
  if (poly_t) {
    const int32_t tail = trailing_t(rd, L);
    el = el && tail >= 3 && (L - tail) >= 11;
    L -= tail;
  }
  L -= trim5 + trim3;
  q = rd >> (2 * trim5);
  return el;
}

// Round 6, reads of 33..63 nt on the seed kernels (LONG instantiations): the read is two words (rd = bases 0..31, the second
// word = bases 32..63); a unit's view of it is q (the first 32 bases behind the 5' trim) + what lies behind.  Seeds,
// bitmaps, buckets and dictionary keys only ever look at q; the second word is compared where an alignment is verified --
// so it is not carried along but FETCHED THERE (QH: where it lies, `v()` loads it): as a value in every lane's registers
// and LDS slot it cost the kernels 30..60 bytes of scratch in their hot loops, and every read of the batch paid for it.
struct QH {
  const uint64_t* p;  // the read's second word in the batch's second plane (null: the read has none)
  uint32_t sh;        // 2 x trim5
  __device__ __forceinline__ uint64_t v() const { return p ? (*p >> sh) : 0ull; }
};
__device__ __forceinline__ bool unit_view2(int32_t min_len, int32_t max_len, int32_t poly_t, int32_t trim5, int32_t trim3, uint64_t rd,
                                           uint64_t hpv, uint32_t L0, uint64_t& q, QH& qh, int32_t& L) {
  bool el = (int32_t)L0 >= min_len && (int32_t)L0 <= max_len;
  L = (int32_t)L0;
  const uint64_t* hp = L0 > 32u ? reinterpret_cast<const uint64_t*>(hpv) : nullptr;
  if (poly_t) {
    int32_t tail;
    if (hp) {
      const int32_t th = trailing_t(*hp, L - 32);
      tail = th == L - 32 ? th + trailing_t(rd, 32) : th;
    } else {
      tail = trailing_t(rd, L);
    }
    el = el && tail >= 3 && (L - tail) >= 11;
    L -= tail;
  }
  L -= trim5 + trim3;
  q = rd;
  if (trim5) q = (rd >> (2 * trim5)) | ((hp ? *hp : 0ull) << (64 - 2 * trim5));
  qh = QH{hp, 2u * (uint32_t)trim5};
  return el;
}

template <bool LONG>
__device__ __forceinline__ bool unit_view_t(int32_t min_len, int32_t max_len, int32_t poly_t, int32_t trim5, int32_t trim3, uint64_t rd,
                                            uint64_t hpv, uint32_t L0, uint64_t& q, QH& qh, int32_t& L) {
  if (LONG) return unit_view2(min_len, max_len, poly_t, trim5, trim3, rd, hpv, L0, q, qh, L);
  qh = QH{nullptr, 0u};
  return unit_view(min_len, max_len, poly_t, trim5, trim3, rd, L0, q, L);
}

// seed length of a read of L bases in a unit: K = V + 1 seeds of floor(R / K) bases
// (LONG: the seeds lie in the first word -- V + 1 disjoint stretches of the first min(R, 32) bases still leave one clean)
template <bool LONG = false>
__device__ __forceinline__ int32_t seed_bases(int32_t L, int32_t min_seed_len, int32_t V) {
  int32_t R = min(L, min_seed_len);
  if (LONG) R = min(R, 32);
  return V ? (R >> 1) : R;
}

__device__ __forceinline__ const void* lds_pointer(const uint32_t* t, uint32_t w) {
  return reinterpret_cast<const void*>((uint64_t)t[w] | ((uint64_t)t[w + 1] << 32));
}

// Does the chain of the read's home slot hold a window that agrees with the read's first 32 bases: the seed part (the first
// min(32, seed_len) bases) letter for letter, at most max_mm_total differences in the rest of the window?  (The negative
// filter of the LONG instantiations: a read of more than 32 bases without such a window cannot align.)
template <class Unit>
__device__ __forceinline__ bool dict_window_hit(const Unit& un, uint64_t q, uint4 first) {
  const int32_t seed_len = un.m[0].seed_len, max_total = un.m[0].max_mm_total;
  const uint32_t smask = (1u << un.log2_slots) - 1u;
  const uint32_t kmask = un.key_bases >= 16u ? 0xFFFFFFFFu : ((1u << (2u * un.key_bases)) - 1u);
  const uint32_t home = (((uint32_t)q & kmask) * kDictHashMul) >> (32u - un.log2_slots);
  const uint64_t seedmask = low_bits(2u * (uint32_t)min(32, seed_len));
  uint4 sl = first;
  const uint32_t chain = (sl.w >> kDictChainShift) & kDictChainMask;
  for (uint32_t jj = 0;; ++jj) {
    const uint64_t m = mismatch_bits((uint64_t)sl.x | ((uint64_t)sl.y << 32), q);
    if ((sl.w & kDictOccBit) && (m & seedmask) == 0ull && (int32_t)__popcll(m) <= max_total && (sl.w & kDictAfterMask) > 32u) return true;
    if (jj >= chain) break;
    sl = un.slots[(home + jj + 1u) & smask];
  }
  return false;
}

// ... and in a unit with seed buckets a long read takes seeds of bucket_k bases where its seed region holds two of them
// (a 28-base region: [0, 11) and [11, 22) -- one mismatch still leaves one clean, and a seed of exactly that length is
// answered in the stream from one bucket line instead of parking the read for the jump table)
template <bool LONG, class KU>
__device__ __forceinline__ int32_t unit_seed_bases(const KU& un, int32_t L) {
  const int32_t k = seed_bases<LONG>(L, un.min_seed_len, un.max_mm_seed);
  if (LONG && L > 32 && un.kind == 0u && un.buckets && un.max_mm_seed == 1 && k > (int32_t)un.bucket_k) return (int32_t)un.bucket_k;
  return k;
}

// One candidate row of a seed whose k' first bases matched at text position wr.x, `off` read
// bases left of it: the key of a valid alignment (pass : 8 | mismatches : 8 | start : 32 | segment : 16), ~0 = none.
// (bucket rows keep their segment room in six bits each: fm_index.hpp)
template <bool LONG = false>
__device__ __forceinline__ unsigned long long seed_row_key(const uint32_t* ut, const uint4 wr, uint64_t q, int32_t L, uint32_t off,
                                                           uint32_t kprime, bool bucket_row, QH qh = QH{nullptr, 0u}) {
  constexpr unsigned long long kNone = ~0ull;
  const uint32_t before = bucket_row ? (wr.y & 63u) : (wr.y & 255u), after = bucket_row ? ((wr.y >> 6) & 63u) : ((wr.y >> 8) & 255u);
  const uint32_t seg16 = wr.y >> 16;
  if (bucket_row && wr.x == 0xFFFFFFFFu) return kNone;
  const uint32_t need_after = (uint32_t)L - off;
  if ((off > before) | (need_after > after)) return kNone;
  const uint32_t limits = ut[UW_LIMITS], flags = ut[UW_FLAGS];
  const int32_t min_seed_len = (int32_t)(limits & 0xFFFFu), max_total = (int32_t)(limits >> 16);
  const int32_t V = (int32_t)((flags >> 21) & 3u);
  // mismatches the stored context shows: the <= 16 bases left of the position, the bases 8..23 right of it
  // (a unit without wide rows -- small libraries: 8-byte rows and a text that stay in L2 -- passes
  // kprime = 0 and zero context words: the text decides)
  uint32_t mm = 0;
  if (kprime) {
    const uint32_t c = min(off, 16u);
    if (c) {
      const uint32_t want = (uint32_t)(q >> (2u * (off - c))) & (uint32_t)low_bits(2u * c);
      const uint32_t x = (wr.z >> (32u - 2u * c)) ^ want;
      mm += (uint32_t)__popc((x | (x >> 1)) & 0x55555555u & (uint32_t)low_bits(2u * c));
    }
    if (need_after > 8u) {
      const uint32_t c2 = min(need_after, 24u) - 8u;
      const uint32_t sh = 2u * (off + 8u);  // (< 64: off <= 16 with wide rows)
      const uint64_t qq = (LONG && sh && L > 32) ? ((q >> sh) | (qh.v() << (64u - sh))) : (q >> sh);
      const uint32_t want = (uint32_t)qq & (uint32_t)low_bits(2u * c2);
      const uint32_t x = (wr.w ^ want) & (uint32_t)low_bits(2u * c2);
      mm += (uint32_t)__popc((x | (x >> 1)) & 0x55555555u);
    }
  }
  if ((int32_t)mm > max_total) return kNone;  // a lower bound of the alignment's mismatches: final
  const uint32_t s = wr.x - off;
  uint64_t mbits = 0, mbits_h = 0;
  const bool covered = kprime != 0u && off <= 16u && need_after <= 24u && kprime >= 8u && L <= min_seed_len;
  if (!covered) {
    const uint32_t* text = reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_TEXT));
    mbits = mismatch_bits(text_window(text, s), q) & low_bits(2u * (uint32_t)L);
    mm = (uint32_t)__popcll(mbits);
    if ((int32_t)mm > max_total) return kNone;
    if (LONG && L > 32) {  // the second word against the text behind the first 32 bases
      mbits_h = mismatch_bits(text_window(text, s + 32u), qh.v()) & low_bits(2u * (uint32_t)(L - 32));
      mm += (uint32_t)__popcll(mbits_h);
      if ((int32_t)mm > max_total) return kNone;
    }
  }
  // which member (pass) the entry belongs to, and that pass's policy
  const uint32_t n_members = (flags >> 18) & 7u;
  uint32_t mi = 0;
  if (n_members > 1u) {
    uint32_t sg = seg16;
    if (sg == 0xFFFFu) {
      const uint32_t* chunk_seg = reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_CHUNKSEG));
      const uint32_t* seg_start = reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGSTART));
      sg = chunk_seg[s >> 5];
      while (seg_start[sg + 1] <= s) ++sg;
    }
    const uint32_t ref = ((flags >> 17) & 1u) ? sg : reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGREF))[sg];
    for (uint32_t j = 1; j < n_members; ++j) mi += ref >= ut[UW_MEMBERS + 2u * j + 1u] ? 1u : 0u;
  }
  const uint32_t mw = ut[UW_MEMBERS + 2u * mi];
  const int32_t pass_index = (int32_t)(mw & 0xFFu), seed_len = (int32_t)((mw >> 8) & 0xFFFFu), m_total = (int32_t)(mw >> 24);
  uint32_t mm_seed = mm;
  if (L > seed_len) {  // (then never `covered`)
    mm_seed = (uint32_t)__popcll(mbits & low_bits(2u * (uint32_t)seed_len));
    if (LONG && seed_len > 32) mm_seed += (uint32_t)__popcll(mbits_h & low_bits(2u * (uint32_t)(seed_len - 32)));
  }
  if ((int32_t)mm_seed > V || (int32_t)mm > m_total) return kNone;
  return ((unsigned long long)pass_index << 56) | ((unsigned long long)mm << 48) | ((unsigned long long)s << 16) | seg16;
}

template <bool LONG = false>
__device__ __forceinline__ void verify_seed_row(const uint32_t* ut, const uint4 wr, uint64_t q, int32_t L, uint32_t off, uint32_t kprime,
                                                unsigned long long* best_slot, bool bucket_row = false, QH qh = QH{nullptr, 0u}) {
  const unsigned long long key = seed_row_key<LONG>(ut, wr, q, L, off, kprime, bucket_row, qh);
  if (key != ~0ull) atomicMin(best_slot, key);
}

}  // namespace

// BUCKETS: some unit has seed buckets (the instantiation without them needs fewer registers: WAVES = 8
// workgroups per CU instead of 6)
// FAT: the input list carries its reads (SeedParams::in_stride = 4) -- an instantiation of its own: as a run-time
// branch of the walk it cost the index-list instantiation 28 more bytes of scratch and 40 % of its speed
// LONG: the batch holds reads of 33..63 nt (second word in p.reads_hi): the <.., true> instantiations carry it; <.., false> is
// the code of round 5
template <bool BUCKETS, int WAVES, bool FAT, bool LONG>
__global__ void __launch_bounds__(kSeedThreads, WAVES) seed_kernel(const SeedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t T = p.reads_per_lane, tile = kSeedThreads * T;
  const SeedLds l = carve_seed_lds(smem, tile, p.row_cap, p.item_cap, p.n_units);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  // ---- unit table, counters, control ----
  if (tid < p.n_units) {
    const SeedUnit& un = p.unit[tid];
    uint32_t* t = l.utab + tid * kUnitWords;
    auto put = [&](uint32_t w, const void* ptr) {
      t[w] = (uint32_t)(uint64_t)ptr;
      t[w + 1] = (uint32_t)((uint64_t)ptr >> 32);
    };
    put(UW_SA16, un.sa16);
    put(UW_TEXT, un.text);
    put(UW_SEGSTART, un.seg_start);
    put(UW_SEGREF, un.seg_ref);
    put(UW_SEGOFF, un.seg_off);
    put(UW_CHUNKSEG, un.chunk_seg);
    put(UW_BUCKETS, un.buckets);
    put(UW_SA, un.sa);
    t[UW_FLAGS] = (uint32_t)un.trim5 | ((uint32_t)un.trim3 << 8) | (un.poly_t ? 1u << 16 : 0u) | (un.simple_segs ? 1u << 17 : 0u) |
                  (un.n_members << 18) | ((uint32_t)un.max_mm_seed << 21) | (un.kind == 1u ? 1u << 23 : 0u);
    t[UW_LIMITS] = (uint32_t)min(un.min_seed_len, 0xFFFF) | ((uint32_t)min(un.max_total, 0xFFFF) << 16);
    for (uint32_t j = 0; j < kSeedMaxMembers; ++j) {
      t[UW_MEMBERS + 2u * j] = (uint32_t)un.m[j].pass_index | ((uint32_t)min(un.m[j].seed_len, 0xFFFF) << 8) | ((uint32_t)min(un.m[j].max_mm_total, 255) << 24);
      t[UW_MEMBERS + 2u * j + 1u] = un.m[j].entry_lo;
    }
  }
  for (uint32_t i = tid; i < kSeedCntSlots; i += kSeedThreads) l.cnt[i] = 0ull;
  if (tid < kSeedCtlWords) l.ctl[tid] = 0u;
  __syncthreads();
  if (p.idx_in) longest_segment(p.in_count, p.in_nseg, &l.ctl[1]);
  __syncthreads();

  const uint32_t in_nseg = p.idx_in ? p.in_nseg : 1u;
  const uint32_t depth_chunks = p.idx_in ? (l.ctl[1] + tile - 1) / tile : (p.n_total + tile - 1) / tile;
  const uint32_t n_chunks = in_nseg * depth_chunks;
  // ---- the walk, software-pipelined: the read of tile + grid and the list entry of tile + 2 grid are
  // in flight while a tile is worked on ----
  // (the list cursor: chunk c is chunk c / in_nseg of segment c % in_nseg; it advances by the grid size,
  // so the division is done once)
  uint32_t f_chunk = blockIdx.x, f_sgi = blockIdx.x % in_nseg, f_depth = blockIdx.x / in_nseg;
  const uint32_t f_dsgi = gridDim.x % in_nseg, f_ddepth = gridDim.x / in_nseg;
  // (an index list: the entry two tiles ahead, then the gathers of the tile ahead; a list that carries its reads --
  // in_stride = 4, SeedParams -- delivers index, length and read in one load: one tile ahead, no gather)
  constexpr bool fat_in = FAT;
  auto fetch_next = [&](uint32_t& r_out, uint32_t& L_out, uint64_t& rd_out) -> bool {
    bool act = false;
    r_out = 0;
    if (f_chunk < n_chunks) {
      // (segment and depth are the workgroup's: scalar registers, so that the segment's count is a scalar
      // load -- a vector load here would be waited for with vmcnt(0) and drain the reads just requested)
      const uint32_t sgi = __builtin_amdgcn_readfirstlane(f_sgi), t_base = __builtin_amdgcn_readfirstlane(f_depth * tile);
      const uint32_t t = t_base + tid;
      const auto* kp = kernel_args_here<SeedParams>();  // (the list's pointers: scalar loads per trip, not spilled registers)
      const uint32_t* idx_in = kp->idx_in;
      if (idx_in) {
        typedef const __attribute__((address_space(4))) uint32_t* const_u32_t;  // (written by the launch before this one)
        act = t < ((const_u32_t)(uintptr_t)kp->in_count)[sgi];
        // (entries of in_stride words: 1 = indices, 4 = a list that carries its reads; the index is word 0)
        if (fat_in) {
          // a list that carries its reads: the whole entry, one 16-byte load -- the caller's read registers directly
          if (act) {
            const uint4 e = reinterpret_cast<const uint4*>(idx_in)[(size_t)sgi * kp->in_seg_cap + t_base + tid];
            r_out = e.x;
            L_out = e.y;
            rd_out = (uint64_t)e.z | ((uint64_t)e.w << 32);
          }
        } else {
          const uint32_t* seg = idx_in + (size_t)sgi * kp->in_seg_cap + t_base;
          if (act) r_out = seg[tid];
        }
      } else {
        act = t < kp->n_total;
        r_out = act ? t : 0u;
      }
    }
    f_chunk += gridDim.x;
    f_sgi += f_dsgi;
    f_depth += f_ddepth;
    if (f_sgi >= in_nseg) {
      f_sgi -= in_nseg;
      ++f_depth;
    }
    return act;
  };
  uint32_t r_b = 0, r_c = 0, L_b = 255u;
  uint64_t rd_b = 0;
  uint32_t c_inl_lookups = 0, c_inl_cands = 0, inl_unit = 0;  // bucket lookups answered in phase 1 (diagnostics, per lane)
  uint32_t el_mask = 0;  // bit u: the lane's read is offered to unit u (set in phase 1, read in phase 3)
  // reads offered to / claimed by each member pass, summed over the wave's tiles: ONE vector register,
  // lane 2c / 2c + 1 = the two counters of member c (units in order, members in order) -- two dozen
  // scalar accumulators live across the loop would be spilled
  uint32_t acc_v = 0;
  uint32_t L_unused = 0;
  uint64_t rd_unused = 0;
  bool act_b = fetch_next(r_b, L_b, rd_b);
  bool act_c = false;
  if (!fat_in) {
    act_c = fetch_next(r_c, L_unused, rd_unused);
    if (act_b) {
      L_b = p.lens[r_b];
      rd_b = p.reads[r_b];
    }
  }
  // The item counters are double-buffered by tile parity: a tile without items has no barrier behind
  // the one that closes phase 1, so a fast wave may already be pushing the NEXT tile's items while a
  // slow one still reads this tile's counters (it cannot get two tiles ahead: the next tile's barrier
  // waits for the slow wave).
  uint32_t ctl_items = 4u;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x, ctl_items ^= (4u ^ (4u + kSeedMaxUnits))) {
    const bool active = act_b;
    const uint32_t r = r_b, L0 = active ? L_b : 255u, slot = tid;
    const uint64_t rd = rd_b;
    L_b = 255u;
    rd_b = 0;
    if (fat_in) {
      act_b = fetch_next(r_b, L_b, rd_b);
    } else {
      act_b = act_c;
      r_b = r_c;
      if (act_b) {
        L_b = p.lens[r_b];
        rd_b = p.reads[r_b];
      }
      act_c = fetch_next(r_c, L_unused, rd_unused);
    }
    // ================= phase 1: the tile's reads into LDS, and their items =================
    // A seed of exactly bucket_k bases in a unit with seed buckets is answered right here: the first four
    // rows of both seeds' buckets are requested together (one 128-byte line per seed, eight loads in
    // flight per lane) and verified into a lane-local best; a fuller or overflowing bucket, any other
    // seed length and every unit without buckets go through the item queue.
    {
      unsigned long long my_best = ~0ull;
      el_mask = 0u;
      // (a long read's second word: gathered here -- the lists carry the first word only)
      const uint64_t rdh = (LONG && active && L0 > 32u) ? (uint64_t)(uintptr_t)(p.reads_hi + r) : 0ull;  // (where it lies: QH)
      l.srd[slot] = rd;
      if (LONG) l.srh[slot] = rdh;
      l.sL0[slot] = (uint8_t)L0;
      for (uint32_t ui = 0; ui < p.n_units; ++ui) {
        const SeedUnit& un = p.unit[ui];
        uint64_t q = 0;
        QH qh{nullptr, 0u};
        int32_t L = 0;
        const bool el = active && unit_view_t<LONG>(un.min_len, un.max_len, un.poly_t, un.trim5, un.trim3, rd, rdh, L0, q, qh, L);
        el_mask |= el ? 1u << ui : 0u;
        const int32_t V = un.max_mm_seed;
        const bool search = el && L > V;
        const uint32_t n_seeds = un.kind == 1u ? 1u : (uint32_t)V + 1u;
        const int32_t k = seed_bases<LONG>(L, un.min_seed_len, V);
        uint32_t queued = search ? (1u << n_seeds) - 1u : 0u;  // seeds that need the item queue
        if (BUCKETS && un.kind == 0u && un.buckets) {
          const bool inl = search && (uint32_t)k == un.bucket_k;
          if (__any(inl)) {
            const uint32_t* ut = l.utab + ui * kUnitWords;
            const uint32_t cmask = (1u << (2u * un.bucket_k)) - 1u;
            const uint32_t base0 = ((uint32_t)q & cmask) * kSeedBucketRows;
            const uint32_t base1 = ((uint32_t)(q >> (2u * un.bucket_k)) & cmask) * kSeedBucketRows;
            const bool two = V >= 1;
            uint4 ra[4], rb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              ra[i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
              rb[i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
            }
            if (inl) {
#pragma unroll
              for (int i = 0; i < 4; ++i) ra[i] = un.buckets[base0 + i];
              if (two) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[i] = un.buckets[base1 + i];
              }
            }
            if (inl) {
              const uint32_t cnt0 = (ra[0].y >> 12) & 15u, cnt1 = two ? ((rb[0].y >> 12) & 15u) : 0u;
              c_inl_lookups += two ? 2u : 1u;
              queued = 0u;
              if (cnt0 <= 4u) {
                c_inl_cands += cnt0;
#pragma unroll
                for (int i = 0; i < 4; ++i) my_best = min(my_best, seed_row_key(ut, ra[i], q, L, 0u, un.bucket_k, true));
              } else {
                queued |= 1u;
              }
              if (two) {
                if (cnt1 <= 4u) {
                  c_inl_cands += cnt1;
#pragma unroll
                  for (int i = 0; i < 4; ++i) my_best = min(my_best, seed_row_key(ut, rb[i], q, L, un.bucket_k, un.bucket_k, true));
                } else {
                  queued |= 2u;
                }
              }
            }
          }
        }
        // the unit's presence bitmaps: both seeds' words are requested before either is looked at
        if (un.kind == 0u && un.kbits && k >= 8) {
          const uint32_t kb = (uint32_t)min(k, 11), woff = seed_kbits_word_off(kb), cmask = (1u << (2u * kb)) - 1u;
          const uint32_t code0 = (uint32_t)q & cmask, code1 = (uint32_t)(q >> (2u * (uint32_t)k)) & cmask;
          uint32_t w0 = ~0u, w1 = ~0u;
          if (queued & 1u) w0 = un.kbits[woff + (code0 >> 5)];
          if (queued & 2u) w1 = un.kbits[woff + (code1 >> 5)];
          if (!((w0 >> (code0 & 31u)) & 1u)) queued &= ~1u;
          if (!((w1 >> (code1 & 31u)) & 1u)) queued &= ~2u;
        }
        for (uint32_t j = 0; j < n_seeds; ++j) {
          const bool need = ((queued >> j) & 1u) != 0u;
          const uint64_t mask = __ballot(need);
          if (mask) {
            uint32_t base = 0;
            if (lane == (uint32_t)__ffsll((long long)mask) - 1u) base = atomicAdd(&l.ctl[ctl_items + ui], (uint32_t)__popcll(mask));
            base = __shfl(base, __ffsll((long long)mask) - 1, 64);
            if (need) l.items[ui * p.item_cap + base + mbcnt(mask)] = slot | (j << 11);
          }
        }
        if (un.kind == 0u && un.buckets) inl_unit = ui;
      }
      l.best[slot] = my_best;
    }
    __syncthreads();
    uint32_t any_items = 0;
    for (uint32_t ui = 0; ui < p.n_units; ++ui) any_items |= l.ctl[ctl_items + ui];
    if (any_items) {  // (workgroup-uniform)
    // ================= phase 2a: items -> dictionary answers / suffix-array rows =================
    for (uint32_t ui = 0; ui < p.n_units; ++ui) {
      const SeedUnit& un = p.unit[ui];
      const uint32_t n_items = l.ctl[ctl_items + ui];
      uint32_t c_lookups = 0, c_cands = 0;
      for (uint32_t it0 = 0; it0 < n_items; it0 += kSeedThreads) {
        const uint32_t it = it0 + tid;
        const bool has = it < n_items;
        uint32_t slot = 0, j = 0;
        uint64_t q = 0;
        QH qh{nullptr, 0u};
        int32_t L = 0;
        if (has) {
          const uint32_t e = l.items[ui * p.item_cap + it];
          slot = e & 2047u;
          j = e >> 11;
          unit_view_t<LONG>(un.min_len, un.max_len, un.poly_t, un.trim5, un.trim3, l.srd[slot], LONG ? l.srh[slot] : 0ull, l.sL0[slot], q, qh, L);
        }
        if (un.kind == 1u) {
          // ---- exact-match dictionary (see exact_dict_kernel) ----
          if (has) {
            const int32_t pass_index = un.m[0].pass_index, seed_len = un.m[0].seed_len, max_total = un.m[0].max_mm_total;
            const uint32_t smask = (1u << un.log2_slots) - 1u;
            const uint32_t kmask = un.key_bases >= 16u ? 0xFFFFFFFFu : ((1u << (2u * un.key_bases)) - 1u);
            // (a read of more than 32 bases always takes the FM rows: dict_unit_probe says why -- unless the table says no)
            bool fallback = (uint32_t)L < un.key_bases || (LONG && L > 32);
            unsigned long long key = ~0ull;
            bool hopeless = false;
            if (LONG && L > 32 && (uint32_t)L >= un.key_bases) {
              const uint4 sl0 = un.slots[(((uint32_t)q & kmask) * kDictHashMul) >> (32u - un.log2_slots)];
              ++c_lookups;
              hopeless = ((sl0.w >> kDictChainShift) & kDictChainMask) != kDictChainOverflow && !dict_window_hit(un, q, sl0);
              fallback = !hopeless;
            }
            if (!fallback && !hopeless) {
              const uint64_t lmask = low_bits(2u * (uint32_t)L), seedmask = low_bits(2u * (uint32_t)min(L, seed_len));
              const uint32_t home = (((uint32_t)q & kmask) * kDictHashMul) >> (32u - un.log2_slots);
              uint4 sl = un.slots[home];
              ++c_lookups;
              const uint32_t chain = (sl.w >> kDictChainShift) & kDictChainMask;
              if (chain == kDictChainOverflow) {
                fallback = true;
              } else {
                uint32_t bestj = ~0u;
                for (uint32_t jj = 0;; ++jj) {
                  const uint64_t win = (uint64_t)sl.x | ((uint64_t)sl.y << 32);
                  ++c_cands;
                  const uint64_t m = mismatch_bits(win, q) & lmask;
                  const uint32_t mmt = (uint32_t)__popcll(m);
                  if ((sl.w & kDictOccBit) && (m & seedmask) == 0ull && (int32_t)mmt <= max_total && (uint32_t)L <= (sl.w & kDictAfterMask) &&
                      ((mmt << 8) | jj) < bestj) {
                    bestj = (mmt << 8) | jj;
                    key = ((unsigned long long)pass_index << 56) | ((unsigned long long)mmt << 48) | ((unsigned long long)sl.z << 21) |
                          (sl.w >> kDictOffShift);
                  }
                  if (jj >= chain || (bestj >> 8) == 0u) break;
                  sl = un.slots[(home + jj + 1u) & smask];
                }
              }
            }
            if (fallback) {
              // the FM index: rows of the longest prefix a jump table knows, each against the text
              const int32_t R = min(L, seed_len);
              uint32_t tab_off = 0;
              const uint32_t kp = un.tabs.k[0] ? pick_table(un.tabs, R, tab_off) : 0u;
              uint32_t lo = 0, hi = un.n + 1u;
              if (kp) {
                const uint32_t* tab = un.ftab + tab_off + lex_code(q & low_bits(2u * kp), kp);
                lo = tab[0];
                hi = tab[1];
              }
              ++c_lookups;
              const uint64_t lmask = low_bits(2u * (uint32_t)L), seedmask = low_bits(2u * (uint32_t)R);
              uint64_t bestk = ~0ull;
              uint32_t bseg = 0xFFFFu, bbefore = 255u;
              for (uint32_t i = lo; i < hi; ++i) {
                const uint64_t row = un.sa[i];
                ++c_cands;
                if ((uint32_t)L > ((uint32_t)(row >> 40) & 255u)) continue;
                const uint64_t m = mismatch_bits(text_window(un.text, (uint32_t)row), q) & lmask;
                uint32_t mmt = (uint32_t)__popcll(m);
                if ((m & seedmask) != 0ull || (int32_t)mmt > max_total) continue;
                if (LONG && L > 32) {  // the second word against the text behind the first 32 bases
                  const uint64_t mh = mismatch_bits(text_window(un.text, (uint32_t)row + 32u), qh.v()) & low_bits(2u * (uint32_t)(L - 32));
                  mmt += (uint32_t)__popcll(mh);
                  if ((R > 32 && (mh & low_bits(2u * (uint32_t)(R - 32))) != 0ull) || (int32_t)mmt > max_total) continue;
                }
                const uint64_t kk = ((uint64_t)mmt << 32) | (uint32_t)row;
                if (kk < bestk) {
                  bestk = kk;
                  bseg = (uint32_t)(row >> 48);
                  bbefore = (uint32_t)(row >> 32) & 255u;
                }
              }
              if (bestk != ~0ull) {
                uint32_t ref, pos;
                SegTables segs{un.seg_start, un.seg_ref, un.seg_off, un.chunk_seg, un.simple_segs};
                locate_entry(segs, (uint32_t)bestk, bseg, bbefore, ref, pos);
                key = ((unsigned long long)pass_index << 56) | ((unsigned long long)(bestk >> 32) << 48) | ((unsigned long long)ref << 21) | pos;
              }
            }
            if (key != ~0ull) atomicMin(&l.best[slot], key);
          }
          continue;
        }
        // ---- a seed: jump-table load, then its rows into the row queue ----
        uint32_t lo = 0, n_rows = 0, tag = 0;
        if (has) {
          const int32_t k = seed_bases<LONG>(L, un.min_seed_len, un.max_mm_seed);
          const uint32_t off = j * (uint32_t)k;
          uint32_t kp = 0, cnt = kSeedBucketOverflow;
          if (BUCKETS && un.buckets && (uint32_t)k == un.bucket_k) {
            // the seed's rows sit in one 128-byte line addressed by the seed itself
            kp = un.bucket_k;
            lo = ((uint32_t)(q >> (2u * off)) & ((1u << (2u * kp)) - 1u)) * kSeedBucketRows;
            cnt = (un.buckets[lo].y >> 12) & 15u;
            ++c_lookups;
          }
          if (cnt != kSeedBucketOverflow) {
            n_rows = cnt;
            tag = slot | (ui << 11) | (off << 13) | (kp << 19) | kRowFromBucket;
          } else {
            uint32_t tab_off = 0, shift = 0;
            const uint32_t K = un.tabs.k[0] ? pick_seed_table(un.tabs, k, tab_off, shift) : 0u;
            uint32_t hi = un.n + 1u;
            lo = 0;
            kp = 0;
            if (K) {
              // (a seed shorter than the table's K: the rows of every K-mer it starts)
              kp = min(K, (uint32_t)k);
              const uint32_t* tab = un.ftab + tab_off + (lex_code((q >> (2u * off)) & low_bits(2u * kp), kp) << shift);
              lo = tab[0];
              hi = tab[1u << shift];
            }
            ++c_lookups;
            n_rows = hi > lo ? hi - lo : 0u;
            tag = slot | (ui << 11) | (off << 13) | (kp << 19);
          }
          c_cands += n_rows;
        }
        const bool is_wide = n_rows > kSeedRowsPerItem;
        const uint32_t mine = is_wide ? 0u : n_rows;
        const uint32_t incl = wave_incl_scan(mine);
        const uint32_t total = __shfl(incl, 63, 64);
        uint32_t base = 0;
        if (total) {
          if (lane == 0) base = atomicAdd(&l.ctl[2], total);
          base = __shfl(base, 0, 64);
        }
        if (mine) {
          // (entries past the queue's capacity -- other tiles' share of a crowded tile -- are verified here)
          const uint32_t first = base + (incl - mine);
          const uint32_t* ut = l.utab + ui * kUnitWords;
          for (uint32_t i = 0; i < mine; ++i) {
            if (first + i < p.row_cap) l.rows[first + i] = make_uint2(lo + i, tag);
            else if ((tag & kRowFromBucket) || un.sa16)
              verify_seed_row<LONG>(ut, (tag & kRowFromBucket) ? un.buckets[lo + i] : un.sa16[lo + i], q, L, (tag >> 13) & 63u, (tag >> 19) & 15u,
                                    &l.best[slot], (tag & kRowFromBucket) != 0u, qh);
            else {
              const uint64_t row = un.sa[lo + i];
              verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), q, L, (tag >> 13) & 63u, 0u, &l.best[slot], false, qh);
            }
          }
        }
        if (is_wide) {
          const uint32_t w = atomicAdd(&l.ctl[3], 1u);
          if (w < kSeedWideCap) {
            l.wide[w] = make_uint4(lo, lo + n_rows, tag, 0u);
          } else {
            const uint32_t* ut = l.utab + ui * kUnitWords;
            for (uint32_t i = 0; i < n_rows; ++i) {
              if (un.sa16) {
                verify_seed_row<LONG>(ut, un.sa16[lo + i], q, L, (tag >> 13) & 63u, (tag >> 19) & 15u, &l.best[slot], false, qh);
              } else {
                const uint64_t row = un.sa[lo + i];
                verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), q, L, (tag >> 13) & 63u, 0u, &l.best[slot], false, qh);
              }
            }
          }
        }
      }
      // diagnostics of the unit: one LDS atomic per wave
      const uint64_t t_l = wave_sum(c_lookups), t_c = wave_sum(c_cands);
      if (lane == 0) {
        if (t_c) atomicAdd(&l.cnt[32u + 2u * ui], (unsigned long long)t_c);
        if (t_l) atomicAdd(&l.cnt[32u + 2u * ui + 1u], (unsigned long long)t_l);
      }
    }
    __syncthreads();
    if (tid < p.n_units) l.ctl[ctl_items + tid] = 0u;  // the item queues are read: empty for the tile after next
    // ================= phase 2b: one row per lane =================
    {
      const uint32_t n_rows = min(l.ctl[2], p.row_cap);
      for (uint32_t x = tid; x < n_rows; x += kSeedThreads) {
        const uint2 e = l.rows[x];
        const uint32_t slot = e.y & 2047u, ui = (e.y >> 11) & 3u, off = (e.y >> 13) & 63u, kp = (e.y >> 19) & 15u;
        const bool from_bucket = (e.y & kRowFromBucket) != 0u;
        const uint32_t* ut = l.utab + ui * kUnitWords;
        const uint32_t flags = ut[UW_FLAGS];
        uint64_t q;
        QH qh{nullptr, 0u};
        int32_t L;
        unit_view_t<LONG>(0, 255, (int32_t)((flags >> 16) & 1u), (int32_t)(flags & 255u), (int32_t)((flags >> 8) & 255u), l.srd[slot],
                          LONG ? l.srh[slot] : 0ull, l.sL0[slot], q, qh, L);
        const uint4* wide = reinterpret_cast<const uint4*>(lds_pointer(ut, from_bucket ? UW_BUCKETS : UW_SA16));
        if (wide) {
          verify_seed_row<LONG>(ut, wide[e.x], q, L, off, kp, &l.best[slot], from_bucket, qh);
        } else {
          const uint64_t row = reinterpret_cast<const uint64_t*>(lds_pointer(ut, UW_SA))[e.x];
          verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), q, L, off, 0u, &l.best[slot], false, qh);
        }
      }
      const uint32_t n_wide = min(l.ctl[3], kSeedWideCap);
      for (uint32_t w = 0; w < n_wide; ++w) {
        const uint4 e = l.wide[w];
        const uint32_t slot = e.z & 2047u, ui = (e.z >> 11) & 3u, off = (e.z >> 13) & 63u, kp = (e.z >> 19) & 15u;
        const uint32_t* ut = l.utab + ui * kUnitWords;
        const uint32_t flags = ut[UW_FLAGS];
        uint64_t q;
        QH qh{nullptr, 0u};
        int32_t L;
        unit_view_t<LONG>(0, 255, (int32_t)((flags >> 16) & 1u), (int32_t)(flags & 255u), (int32_t)((flags >> 8) & 255u), l.srd[slot],
                          LONG ? l.srh[slot] : 0ull, l.sL0[slot], q, qh, L);
        const uint4* sa16 = reinterpret_cast<const uint4*>(lds_pointer(ut, UW_SA16));
        const uint64_t* sa8 = reinterpret_cast<const uint64_t*>(lds_pointer(ut, UW_SA));
        for (uint32_t i = e.x + tid; i < e.y; i += kSeedThreads) {
          if (sa16) {
            verify_seed_row<LONG>(ut, sa16[i], q, L, off, kp, &l.best[slot], false, qh);
          } else {
            const uint64_t row = sa8[i];
            verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), q, L, off, 0u, &l.best[slot], false, qh);
          }
        }
      }
    }
    __syncthreads();
    }
    // ================= phase 3: the claim, outputs, survivors, counters =================
    // (no barrier between this and the next tile's phase 1: both touch the lane's own slot only; the
    // row queue, read before the barrier above, is emptied here, ahead of the next tile's barrier)
    if (tid == 0) {
      l.ctl[2] = 0u;
      l.ctl[3] = 0u;
    }
    {
      const unsigned long long key = l.best[slot];
      const bool claimed = key != ~0ull;
      const int32_t cp = claimed ? (int32_t)(key >> 56) : 255;
      uint32_t o_ref = 0, o_pos = 0;
      uint32_t tile_cnt = 0, c = 0, sel = 0;  // sel: unit << 2 | member of the claiming pass
      for (uint32_t ui = 0; ui < p.n_units; ++ui) {
        const SeedUnit& un = p.unit[ui];
        const bool el = ((el_mask >> ui) & 1u) != 0u;
        for (uint32_t mi = 0; mi < un.n_members; ++mi, ++c) {
          const int32_t pi = un.m[mi].pass_index;
          const uint32_t n_off = (uint32_t)__popcll(__ballot(el && cp >= pi)), n_al = (uint32_t)__popcll(__ballot(cp == pi));
          tile_cnt = lane == 2u * c ? n_off : tile_cnt;
          tile_cnt = lane == 2u * c + 1u ? n_al : tile_cnt;
          sel = cp == pi ? (ui << 2) | mi : sel;
        }
      }
      acc_v += tile_cnt;
      if (claimed) {
        // one decode, with the claiming unit's tables from the LDS unit table
        const uint32_t* ut = l.utab + (sel >> 2) * kUnitWords;
        const uint32_t flags = ut[UW_FLAGS];
        if (flags & (1u << 23)) {
          o_ref = (uint32_t)(key >> 21) & 0x7FFFFFFu;
          o_pos = (uint32_t)key & 0x1FFFFFu;
        } else {
          SegTables segs{reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGSTART)), reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGREF)),
                         reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGOFF)), reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_CHUNKSEG)),
                         (flags >> 17) & 1u};
          uint32_t ref;
          locate_entry(segs, (uint32_t)(key >> 16), (uint32_t)key & 0xFFFFu, 255u, ref, o_pos);
          o_ref = ref - ut[UW_MEMBERS + 2u * (sel & 3u) + 1u];
        }
      }
      if (p.packed) {
        if (claimed) p.packed[r] = pack_assignment(cp, o_ref, o_pos, (uint32_t)((key >> 48) & 255u));
        else if (active && !p.idx_out && !p.out_init) p.packed[r] = 0u;
      } else if (claimed) {
        p.pass_id[r] = (int8_t)cp;
        p.ref_id[r] = (int32_t)o_ref;
        p.pos[r] = (int32_t)o_pos;
        p.mm[r] = (uint8_t)((key >> 48) & 255u);
      } else if (active && !p.idx_out && !p.out_init) {
        p.pass_id[r] = (int8_t)-1;
        p.ref_id[r] = -1;
        p.pos[r] = -1;
        p.mm[r] = 0;
      }
      if (p.idx_out) {
        const bool survive = active && !claimed;
        const uint64_t mask = __ballot(survive);
        if (mask) {
          uint32_t wbase = 0;
          if (lane == 0) wbase = atomicAdd(&l.ctl[0], (uint32_t)__popcll(mask));
          wbase = __shfl(wbase, 0, 64);
          // (the survivor takes its read along: the next launch streams 16-byte entries instead of gathering; from the
          // lane's own LDS slot -- kept in registers across phase 2 they were spilled, and the launch took 1.87 ms
          // instead of 1.32)
          if (survive) {
            const unsigned long long own = l.srd[slot];
            reinterpret_cast<uint4*>(p.idx_out)[(size_t)blockIdx.x * p.out_seg_cap + wbase + mbcnt(mask)] =
                make_uint4(r, (uint32_t)l.sL0[slot], (uint32_t)own, (uint32_t)(own >> 32));
          }
        }
      }
    }
  }
  // ---- counters: one global atomic per non-zero counter and workgroup ----
  {
    const uint64_t t_l = wave_sum(c_inl_lookups), t_c = wave_sum(c_inl_cands);
    if (lane == 0) {
      if (t_c) atomicAdd(&l.cnt[32u + 2u * inl_unit], (unsigned long long)t_c);
      if (t_l) atomicAdd(&l.cnt[32u + 2u * inl_unit + 1u], (unsigned long long)t_l);
    }
    // lane 2c / 2c + 1 holds the wave's counts of member c
    uint32_t c = 0, cslot = 0;
    for (uint32_t ui = 0; ui < p.n_units; ++ui)
      for (uint32_t mi = 0; mi < p.unit[ui].n_members; ++mi, ++c) {
        const uint32_t pi = (uint32_t)p.unit[ui].m[mi].pass_index;
        cslot = (lane >> 1) == c ? 2u * pi + (lane & 1u) : cslot;
      }
    if (lane < 2u * c && acc_v) atomicAdd(&l.cnt[cslot], (unsigned long long)acc_v);
  }
  __syncthreads();
  if (tid < 32u) {
    const unsigned long long v = l.cnt[tid];
    if (v) atomicAdd((unsigned long long*)&p.stats[(tid >> 1) * 5u + (tid & 1u)], v);
  } else if (tid < 32u + 2u * p.n_units) {
    const uint32_t ui = (tid - 32u) >> 1, what = (tid - 32u) & 1u;  // 0 candidates, 1 lookups
    const unsigned long long v = l.cnt[tid];
    if (v) atomicAdd((unsigned long long*)&p.stats[(uint32_t)p.unit[ui].m[0].pass_index * 5u + 3u + what], v);
  }
  if (p.idx_out && tid == 0) p.out_count[blockIdx.x] = l.ctl[0];
}

// ---------------------------------------------------------------------------
// wave_seed_kernel: the same units as seed_kernel, but every WAVE works on its own -- no barrier between
// the setup and the final counter flush.  What the tile kernel's counters said: three
// quarters of the reads a seed launch walks have no item in any unit (their seeds fail the presence
// bitmaps, their buckets are answered in one trip) and still paid the tile protocol: LDS staging, three
// barriers, the claim over a tile with holes.  Here a lane
//   streams its read: eligibility per unit, bitmap probes, the inline answers (a bucket seed's rows, a
//   dictionary's home slot) -- and FINISHES it on the spot when nothing is left to look up: claim,
//   output, survivor list, counters;
//   parks it as a CANDIDATE in the wave's own LDS region otherwise.  As soon as 64 candidates are parked
//   the wave works them off with dense lanes: their items unit by unit (jump-table / bucket-count /
//   slot loads), the rows of those items through a row queue (64 rows verified per trip, whoever's
//   they are, 64-bit atomic min into the candidate's slot), then the claim of the 64 candidates.
// Waves of a workgroup share the unit table, the counter slots and the cursor of the workgroup's
// survivor segment, nothing else; a slow wave (a poly-A seed with 10^5 rows) delays nobody.
// ---------------------------------------------------------------------------
namespace {

constexpr uint32_t kWaveCand = 128u;   // candidate slots of a wave (fewer than 64 parked + the 64 of one trip)
constexpr uint32_t kWaveItems = 128u;  // item list of a wave (fewer than 64 pending + one plane of 64)
constexpr uint32_t kWaveRowsMin = 128u;  // smallest row queue: fewer than 64 pending + one item's kSeedRowsPerItem rows must fit

struct WaveLds {
  unsigned long long* rd;    // [kWaveCand] packed read
  unsigned long long* rdh;   // [kWaveCand] its second word (LONG instantiations; else = rd)
  unsigned long long* best;  // [kWaveCand] best key
  uint32_t* r;               // [kWaveCand] read index
  uint32_t* meta;            // [kWaveCand] length | eligibility mask << 8 | queued seeds << 16
  uint2* rows;               // [row_cap] (row index, candidate | offset << 13 | k' << 19 | bucket flag)
  uint16_t* items;           // [kWaveItems] candidate lane | seed << 6
};

__host__ __device__ constexpr uint32_t wave_lds_bytes(uint32_t row_cap, bool long_reads) {
  return kWaveCand * (8u + 8u + 4u + 4u) + (long_reads ? kWaveCand * 8u : 0u) + row_cap * 8u + kWaveItems * 2u;
}
__host__ __device__ constexpr uint32_t wave_shared_words() { return kSeedMaxUnits * kUnitWords + kSeedCtlWords + 2u * kSeedCntSlots; }

// LDS written by some lanes of the wave, read by others: program order is enough in hardware (one wave's
// LDS operations execute in order); this keeps the compiler from moving them across
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A dictionary unit's answer for one read (exact_dict_kernel's probe): the best key, ~0 = none.
// `first`: the home slot when the caller has loaded it already (first.w's occupied bit may be clear).
// `wide` (may be null): set when the FM fallback's interval holds more than kDictFallbackRows rows -- the rows are then
// NOT compared here (one lane, one row after the other: 50 000 rows for a poly-A read whose home's first positions sit
// in a tail too short for it = a millisecond of ONE lane, and the launch waits for it); the caller leaves the read to
// the wave (dict_fallback_wave, behind the stream).
constexpr uint32_t kDictFallbackRows = 64u;
// LONG: a read of more than 32 bases (qh = bases 32..) always takes the FM search: a slot's window shows 32 bases, and a
// position whose window and room equal an earlier one's was left out of the table -- for a longer read the text behind the
// window decides, which the table does not know.
template <bool LONG = false, class Unit>
__device__ __forceinline__ unsigned long long dict_unit_probe(const Unit& un, uint64_t q, int32_t L, bool have_first, uint4 first,
                                                              uint32_t& c_lookups, uint32_t& c_cands, bool* wide = nullptr, QH qh = QH{nullptr, 0u}) {
  const int32_t pass_index = un.m[0].pass_index, seed_len = un.m[0].seed_len, max_total = un.m[0].max_mm_total;
  const uint32_t smask = (1u << un.log2_slots) - 1u;
  const uint32_t kmask = un.key_bases >= 16u ? 0xFFFFFFFFu : ((1u << (2u * un.key_bases)) - 1u);
  bool fallback = (uint32_t)L < un.key_bases || (LONG && L > 32);
  unsigned long long key = ~0ull;
  if (LONG && L > 32 && (uint32_t)L >= un.key_bases) {
    // (the table as a negative filter: no window like the read's first 32 bases in the home's chain -> no alignment)
    uint4 sl = first;
    if (!have_first) {
      sl = un.slots[(((uint32_t)q & kmask) * kDictHashMul) >> (32u - un.log2_slots)];
      ++c_lookups;
    }
    if (((sl.w >> kDictChainShift) & kDictChainMask) != kDictChainOverflow && !dict_window_hit(un, q, sl)) return ~0ull;
  }
  if (!fallback) {
    const uint64_t lmask = low_bits(2u * (uint32_t)L), seedmask = low_bits(2u * (uint32_t)min(L, seed_len));
    const uint32_t home = (((uint32_t)q & kmask) * kDictHashMul) >> (32u - un.log2_slots);
    uint4 sl = first;
    if (!have_first) {
      sl = un.slots[home];
      ++c_lookups;
    }
    uint32_t chain = (sl.w >> kDictChainShift) & kDictChainMask;
    // (an overflowed home keeps its first -- lowest -- positions in the chain: an exact match among them is the answer,
    // anything else asks the FM index: exact_dict_kernel)
    const bool over = chain == kDictChainOverflow;
    if (over) chain = kDictChainOverflow - 1u;
    uint32_t bestj = ~0u;
    for (uint32_t jj = 0;; ++jj) {
      const uint64_t win = (uint64_t)sl.x | ((uint64_t)sl.y << 32);
      ++c_cands;
      const uint64_t m = mismatch_bits(win, q) & lmask;
      const uint32_t mmt = (uint32_t)__popcll(m);
      if ((sl.w & kDictOccBit) && (m & seedmask) == 0ull && (int32_t)mmt <= max_total && (uint32_t)L <= (sl.w & kDictAfterMask) &&
          ((mmt << 8) | jj) < bestj) {
        bestj = (mmt << 8) | jj;
        key = ((unsigned long long)pass_index << 56) | ((unsigned long long)mmt << 48) | ((unsigned long long)sl.z << 21) |
              (sl.w >> kDictOffShift);
      }
      if (jj >= chain || (bestj >> 8) == 0u) break;
      sl = un.slots[(home + jj + 1u) & smask];
    }
    if (over && (bestj >> 8) != 0u) {
      fallback = true;
      key = ~0ull;
    }
  }
  if (fallback) {
    // the FM index (device_util.hpp: fm_exact_search)
    uint32_t bseg = 0xFFFFu, bbefore = 255u, rows = 0, steps = 0;
    const uint64_t bestk = fm_exact_search<LONG>(un.blocks, un.super, un.primary, un.ftab, un.tabs, un.sa, un.text, un.n, q, L, min(L, seed_len),
                                                 max_total, bseg, bbefore, rows, steps, wide ? kDictFallbackRows : 0xFFFFFFFFu, (LONG && L > 32) ? qh.v() : 0ull);
    if (wide && rows > kDictFallbackRows) {
      *wide = true;
      c_lookups += 1u + steps;
      return ~0ull;
    }
    c_lookups += 1u + steps;
    c_cands += rows;
    if (bestk != ~0ull) {
      uint32_t ref, pos;
      SegTables segs{un.seg_start, un.seg_ref, un.seg_off, un.chunk_seg, un.simple_segs};
      locate_entry(segs, (uint32_t)bestk, bseg, bbefore, ref, pos);
      key = ((unsigned long long)pass_index << 56) | ((unsigned long long)(bestk >> 32) << 48) | ((unsigned long long)ref << 21) | pos;
    }
  }
  return key;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// The FM fallback of a dictionary unit by the whole wave (wave-uniform arguments): the backward search as in
// fm_exact_search (every lane the same addresses), the interval's rows 64 a trip.  Returns the unit's key for the read.
template <bool LONG = false, class Unit>
__device__ __forceinline__ unsigned long long dict_fallback_wave(const Unit& un, uint64_t q, int32_t L, uint32_t* cands, QH qhp = QH{nullptr, 0u}) {
  const uint64_t qh = LONG ? qhp.v() : 0ull;  // (one read, the whole wave: loaded once)
  const uint32_t lane = threadIdx.x & 63u;
  const int32_t pass_index = un.m[0].pass_index, seed_len = un.m[0].seed_len, max_total = un.m[0].max_mm_total;
  const int32_t R = min(L, seed_len);
  uint32_t tab_off = 0;
  const uint32_t k = un.tabs.k[0] ? pick_table(un.tabs, R, tab_off) : 0u;
  uint32_t lo = 0, hi = un.n + 1u;
  int32_t j = R;
  auto from = [&](int32_t at) -> uint64_t {  // the bases from read offset `at` on (fm_exact_search)
    if (!LONG) return q >> (2u * (uint32_t)at);
    if (at >= 32) return qh >> (2u * (uint32_t)(at - 32));
    return at ? ((q >> (2u * (uint32_t)at)) | (qh << (64u - 2u * (uint32_t)at))) : q;
  };
  if (k) {
    j = R - (int32_t)k;
    const uint32_t* tab = un.ftab + tab_off + lex_code(from(j) & low_bits(2u * k), k);
    lo = tab[0];
    hi = tab[1];
  }
  while (j > 0 && hi > lo) {
    --j;
    const uint32_t c = (uint32_t)from(j) & 3u;
    lo = lf_step(un.blocks, un.super, un.primary, c, lo);
    hi = lf_step(un.blocks, un.super, un.primary, c, hi);
  }
  const uint64_t tailmask = LONG ? (low_bits(2u * (uint32_t)min(L, 32)) & ~low_bits(2u * (uint32_t)min(R, 32)))
                                 : (low_bits(2u * (uint32_t)L) & ~low_bits(2u * (uint32_t)R));
  const uint64_t tailmask_h = (LONG && L > 32) ? (low_bits(2u * (uint32_t)(L - 32)) & ~low_bits(2u * (uint32_t)max(R - 32, 0))) : 0ull;
  unsigned long long best = ~0ull;  // mismatches : 8 | position : 32 | before : 8 | segment : 16
  for (uint32_t i = lo + lane; i < hi; i += 64u) {
    const uint64_t row = un.sa[i];
    ++*cands;
    if ((uint32_t)L > ((uint32_t)(row >> 40) & 255u)) continue;
    const uint32_t s = (uint32_t)row;
    uint32_t mmt = 0;
    if (L > R) {
      if (!LONG || tailmask) mmt = (uint32_t)__popcll(mismatch_bits(text_window(un.text, s), q) & tailmask);
      if ((int32_t)mmt > max_total) continue;
      if (LONG && tailmask_h) {
        mmt += (uint32_t)__popcll(mismatch_bits(text_window(un.text, s + 32u), qh) & tailmask_h);
        if ((int32_t)mmt > max_total) continue;
      }
    }
    const unsigned long long key = ((unsigned long long)mmt << 56) | ((unsigned long long)s << 24) | (((row >> 32) & 255ull) << 16) | (row >> 48);
    best = key < best ? key : best;
  }
  best = wave_min_u64(best);
  if (best == ~0ull) return ~0ull;
  uint32_t ref, pos;
  SegTables segs{un.seg_start, un.seg_ref, un.seg_off, un.chunk_seg, un.simple_segs};
  locate_entry(segs, (uint32_t)(best >> 24), (uint32_t)best & 0xFFFFu, (uint32_t)(best >> 16) & 255u, ref, pos);
  return ((unsigned long long)pass_index << 56) | ((best >> 56) << 48) | ((unsigned long long)ref << 21) | pos;
}

constexpr uint32_t kMetaWalkLater = 1u << 31;
constexpr uint32_t kMetaDictLaterShift = 28u;  // ... bits 28-30: the dictionary probe of unit 0 / 1 / 2 is left to the end too  // candidate meta word: a seed of the read is left to its position list

// A seed whose bucket row 0 is `hdr`: its k-mer overflows, has a position list, and the unit is one where a walk in text
// order may stop early -- ONE library, one seed mismatch, the whole read inside the seed region.
template <class KU>
__device__ __forceinline__ bool seed_is_listed(const KU& un, const uint4 hdr, int32_t L) {
  return ((hdr.y >> 12) & 15u) == kSeedBucketOverflow && un.pos_rows != nullptr && hdr.w > kSeedRowsPerItem && un.n_members == 1u &&
         un.max_mm_seed == 1 && L <= un.min_seed_len;
}


// A read of a one-mismatch unit (ONE library, the whole read is the seed region, two seeds of k = bucket_k bases) with a
// seed whose k-mer has 10^2..10^5 rows (an interspersed element, poly-A, a tandem motif): the best alignment through
// such seeds, by the whole wave, WITHOUT verifying every row of their suffix intervals (round 5: 64 rows a trip; one
// poly-A seed is 350 trips of ONE wave, and the launch waits for that wave).  s0 / s1: (first row, rows) of the POSITION
// LIST (fm_index.hpp: seed_pos_lists -- the k-mer's rows in text order) of seed 0 (read offset 0) / seed 1 (offset k);
// .y == 0: that seed is not wide -- the caller has verified EVERY row of it, and found no exact alignment.
//   * ONE wide seed: an alignment through it has the other half of the read with at most one substitution.  Without a
//     substitution it lies in the other seed's rows (verified).  With one, the other half is one of 3 k variants, and
//     each variant's rows sit in its own seed bucket: 33 bucket lines, one per lane, every row verified -- complete, and
//     the wide seed's rows are never touched.  (A variant whose own bucket overflows: the list is walked after all.)
//   * BOTH seeds wide: a list is in TEXT ORDER and the answer within a stratum is the lowest position, so a walk stops
//     at the first 64-row stretch with a valid alignment; an exact alignment beats every other wherever it lies, and
//     lies in BOTH lists: when a list's first valid stretch holds an exact row, its lowest is the lowest exact
//     alignment.  Both first hits with a mismatch: whether an exact alignment exists LATER is decided in the
//     suffix-sorted rows of seed 0, ordered by the text behind the seed -- a 64-ary search for seed 1's bases (two or
//     three trips, from the rows' own context words) names the rows where BOTH seeds match; up to 64 are verified where
//     they are, more are looked for along the list.  A walk that finds nothing in its first stretches (a long list, few
//     alignments) is cut short and that side completed through the other half's variants, as above.
// Returns the best key (~0: none); *cands += rows looked at.  Wave-uniform arguments.
template <class KU>
__device__ __forceinline__ unsigned long long pos_list_answer(const KU* un, const uint32_t* ut, uint64_t q, int32_t L, uint2 s0, uint2 s1,
                                                              uint32_t* cands) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t k = un->bucket_k;
  const uint4* pos_rows = un->pos_rows;
  uint32_t looked = 0;
  constexpr uint32_t kAll = 0xFFFFFFFFu;
  // the first 64-row stretch of a list with a valid alignment: its best key (exact_only: rows with a mismatch do not
  // count); gives up after max_chunks stretches (resolved = false: the list holds more rows, none valid so far)
  auto walk = [&](uint2 s, uint32_t off, bool exact_only, uint32_t max_chunks, bool& resolved, uint32_t from_chunk = 0u) -> unsigned long long {
    resolved = true;
    uint32_t chunks = from_chunk;
    for (uint32_t base = 64u * from_chunk; base < s.y; base += 64u, ++chunks) {
      if (chunks == max_chunks) {
        resolved = false;
        return ~0ull;
      }
      unsigned long long key = ~0ull;
      if (base + lane < s.y) {
        key = seed_row_key(ut, pos_rows[s.x + base + lane], q, L, off, k, false);
        ++looked;
        if (exact_only && ((key >> 48) & 255ull) != 0ull) key = ~0ull;
      }
      const unsigned long long m = wave_min_u64(key);
      if (m != ~0ull) return m;
    }
    return ~0ull;
  };
  // every alignment with exactly one substitution in the k bases at read offset `off`, through the 3 k variants' buckets
  // (a lane a variant); overflow: some variant's bucket does not hold all of its rows
  const uint32_t cmask = (1u << (2u * k)) - 1u;
  auto variants = [&](uint32_t off, bool& overflow) -> unsigned long long {
    unsigned long long key = ~0ull;
    bool over = false;
    if (lane < 3u * k) {
      // (seed_row_key takes the k bases at `off` as matched -- they are what the bucket was addressed with -- so the row
      // is judged against the read WITH the substitution: nothing else may differ, and the substitution is the one mismatch)
      const uint64_t flip = (uint64_t)(lane % 3u + 1u) << (2u * (off + lane / 3u));
      const uint64_t qv = q ^ flip;
      const uint4* line = un->buckets + (size_t)((uint32_t)(qv >> (2u * off)) & cmask) * kSeedBucketRows;
      const uint4 r0 = line[0];
      const uint32_t cnt = (r0.y >> 12) & 15u;
      ++looked;
      if (cnt == kSeedBucketOverflow) {
        over = true;
      } else {
        for (uint32_t i = 0; i < cnt; ++i) {
          unsigned long long kk = seed_row_key(ut, i ? line[i] : r0, qv, L, off, k, true);
          kk = (kk != ~0ull && ((kk >> 48) & 255ull) == 0ull) ? kk + (1ull << 48) : ~0ull;
          key = kk < key ? kk : key;
        }
      }
    }
    overflow = __any(over) != 0;
    return wave_min_u64(key);
  };
  unsigned long long best = ~0ull;
  bool res = true, over = false;
  if (!s0.y || !s1.y) {
    // ---- one wide seed: the other half's variants ----
    const bool wide0 = s0.y != 0u;
    best = variants(wide0 ? k : 0u, over);
    if (over) {
      const unsigned long long w = walk(wide0 ? s0 : s1, wide0 ? 0u : k, false, kAll, res);
      best = w < best ? w : best;
    }
    *cands += looked;
    return best;
  }
  // ---- both seeds wide ----
  // (one trip for what does not depend on anything: the first stretch of both lists and seed 0's interval bounds)
  uint32_t tab_off = 0, shift = 0;
  pick_seed_table(un->tabs, (int32_t)k, tab_off, shift);
  const uint32_t* tab = un->ftab + tab_off + (lex_code(q & low_bits(2u * k), k) << shift);
  uint4 f0 = make_uint4(0u, 0u, 0u, 0u), f1 = f0;
  if (lane < s0.y) f0 = pos_rows[s0.x + lane];
  if (lane < s1.y) f1 = pos_rows[s1.x + lane];
  uint32_t lo = tab[0];
  const uint32_t hi = tab[1u << shift];
  bool res0 = true, res1 = true;
  unsigned long long a = lane < s0.y ? seed_row_key(ut, f0, q, L, 0u, k, false) : ~0ull;
  unsigned long long b = lane < s1.y ? seed_row_key(ut, f1, q, L, k, k, false) : ~0ull;
  looked += (lane < s0.y ? 1u : 0u) + (lane < s1.y ? 1u : 0u);
  a = wave_min_u64(a);
  b = wave_min_u64(b);
  if (a == ~0ull && s0.y > 64u) a = walk(s0, 0u, false, 4u, res0, 1u);
  if (b == ~0ull && s1.y > 64u) b = walk(s1, k, false, 4u, res1, 1u);
  if (!res0) {  // (alignments with seed 0 clean and a substitution in the other half; without one: the exact search below)
    a = variants(k, over);
    if (over) a = walk(s0, 0u, false, kAll, res);
  }
  if (!res1) {
    b = variants(0u, over);
    if (over) b = walk(s1, k, false, kAll, res);
  }
  best = a < b ? a : b;
  // an exact alignment further along?  not when a side was walked to its end without any hit (an exact alignment lies
  // in both lists), not when a walk's first valid stretch already held one
  const bool none0 = res0 && a == ~0ull, none1 = res1 && b == ~0ull;
  if (!none0 && !none1 && ((best >> 48) & 255ull) != 0ull) {
    // ---- seed 0's rows, sorted by what follows the seed ----
    // (the k bases of seed 1, not the whole rest: a 23-nt read's last base lies outside both seeds, and an alignment with
    // both seeds clean and THAT base different is in neither side's variants -- the rows found here are verified in full)
    const uint32_t rest = k;  // inside the row's right context (bases 8..23 behind its position)
    const uint32_t want = lex_code((q >> (2u * k)) & low_bits(2u * rest), rest);
    auto row_rest = [&](uint32_t i) -> uint32_t {
      const uint4 r = un->sa16[i];
      return lex_code((uint64_t)(r.w >> (2u * (k - kWideRowRightSkip))) & low_bits(2u * rest), rest);
    };
    uint32_t n = hi > lo ? hi - lo : 0u;
    while (n > 64u) {  // lower bound of `want`: 64 probes a trip
      const uint32_t step = (n + 63u) / 64u;
      const uint32_t i = lo + lane * step;
      const bool in = lane * step < n;
      const bool less = in && row_rest(i) < want;
      looked += in ? 1u : 0u;
      const uint32_t n_less = (uint32_t)__popcll(__ballot(less));  // (a prefix of the lanes: the rows are sorted)
      if (n_less == 0u) {
        n = 0u;  // the first row is not smaller: the lower bound is `lo` itself
        break;
      }
      const uint32_t nlo = lo + (n_less - 1u) * step + 1u;
      const uint32_t nend = min(lo + n, lo + n_less * step);
      lo = nlo;
      n = nend > nlo ? nend - nlo : 0u;
    }
    if (n) {
      const bool less = lane < n && row_rest(lo + lane) < want;
      looked += lane < n ? 1u : 0u;
      lo += (uint32_t)__popcll(__ballot(less));
    }
    // rows [lo, ...) whose rest equals `want`: the exact occurrences (contiguous)
    unsigned long long e = ~0ull;
    bool eq = false;
    if (lo + lane < hi) {
      const uint4 r = un->sa16[lo + lane];
      ++looked;
      eq = lex_code((uint64_t)(r.w >> (2u * (k - kWideRowRightSkip))) & low_bits(2u * rest), rest) == want;
      if (eq) e = seed_row_key(ut, r, q, L, 0u, k, false);
    }
    const uint64_t eqm = __ballot(eq);
    e = wave_min_u64(e);
    if (eqm == ~0ull) {  // more than 64 of them: the list in text order -- its first valid row, and its first exact one
      const unsigned long long e2 = walk(s0, 0u, false, kAll, res), e3 = walk(s0, 0u, true, kAll, res);
      e = e2 < e ? e2 : e;
      e = e3 < e ? e3 : e;
    }
    best = e < best ? e : best;
  }
  *cands += looked;
  return best;
}

}  // namespace

// LONG: the batch holds reads of 33..63 nt (second word in p.reads_hi; unit_view2, seed_bases<LONG>, seed_row_key<LONG>):
// the instantiations <.., .., true>; <.., .., false> is the code of round 5, nothing of that in it.
template <bool BUCKETS, int WAVES, bool LONG>
__global__ void __launch_bounds__(kSeedThreads, WAVES) wave_seed_kernel(const SeedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  // (the units are read where they sit, in the kernel-argument segment: scalar loads; a by-value copy whose
  // address reaches the nested lambdas is moved to scratch as a whole -- 1 KB per lane)
  typedef const __attribute__((address_space(4))) SeedUnit KUnit;
  const auto* const kargs = kernel_args_here<SeedParams>();
  uint32_t* const utab = smem;
  uint32_t* const ctl = utab + kSeedMaxUnits * kUnitWords;  // [0] survivors of the workgroup, [1] longest input segment
  unsigned long long* const cnt = reinterpret_cast<unsigned long long*>(ctl + kSeedCtlWords);
  WaveLds w;
  {
    uint8_t* base = reinterpret_cast<uint8_t*>(smem + wave_shared_words()) + (size_t)wv * wave_lds_bytes(p.row_cap, LONG);
    w.rd = reinterpret_cast<unsigned long long*>(base);
    w.rdh = LONG ? w.rd + kWaveCand : w.rd;   // (second words of the candidates; no array of its own without long reads)
    w.best = w.rd + (LONG ? 2u : 1u) * kWaveCand;
    w.rows = reinterpret_cast<uint2*>(w.best + kWaveCand);
    w.r = reinterpret_cast<uint32_t*>(w.rows + p.row_cap);
    w.meta = w.r + kWaveCand;
    w.items = reinterpret_cast<uint16_t*>(w.meta + kWaveCand);
  }
  const uint32_t row_cap = p.row_cap;
  // ---- unit table, counters, control (as seed_kernel) ----
  if (tid < p.n_units) {
    const KUnit& un = kargs->unit[tid];
    uint32_t* t = utab + tid * kUnitWords;
    auto put = [&](uint32_t wd, const void* ptr) __attribute__((always_inline)) {
      t[wd] = (uint32_t)(uint64_t)ptr;
      t[wd + 1] = (uint32_t)((uint64_t)ptr >> 32);
    };
    put(UW_SA16, un.sa16);
    put(UW_TEXT, un.text);
    put(UW_SEGSTART, un.seg_start);
    put(UW_SEGREF, un.seg_ref);
    put(UW_SEGOFF, un.seg_off);
    put(UW_CHUNKSEG, un.chunk_seg);
    put(UW_BUCKETS, un.buckets);
    put(UW_SA, un.sa);
    put(UW_BPAIR, un.bpair_rows);
    t[UW_FLAGS] = (uint32_t)un.trim5 | ((uint32_t)un.trim3 << 8) | (un.poly_t ? 1u << 16 : 0u) | (un.simple_segs ? 1u << 17 : 0u) |
                  (un.n_members << 18) | ((uint32_t)un.max_mm_seed << 21) | (un.kind == 1u ? 1u << 23 : 0u);
    t[UW_LIMITS] = (uint32_t)min(un.min_seed_len, 0xFFFF) | ((uint32_t)min(un.max_total, 0xFFFF) << 16);
    for (uint32_t j = 0; j < kSeedMaxMembers; ++j) {
      t[UW_MEMBERS + 2u * j] = (uint32_t)un.m[j].pass_index | ((uint32_t)min(un.m[j].seed_len, 0xFFFF) << 8) | ((uint32_t)min(un.m[j].max_mm_total, 255) << 24);
      t[UW_MEMBERS + 2u * j + 1u] = un.m[j].entry_lo;
    }
  }
  for (uint32_t i = tid; i < kSeedCntSlots; i += kSeedThreads) cnt[i] = 0ull;
  if (tid < kSeedCtlWords) ctl[tid] = 0u;
  __syncthreads();
  if (p.idx_in) longest_segment(p.in_count, p.in_nseg, &ctl[1]);
  __syncthreads();

  const uint32_t tile = kSeedThreads;
  const uint32_t in_nseg = p.idx_in ? p.in_nseg : 1u;
  const uint32_t depth_chunks = p.idx_in ? (ctl[1] + tile - 1) / tile : (p.n_total + tile - 1) / tile;
  const uint32_t n_chunks = in_nseg * depth_chunks;
  // ---- the walk (seed_kernel's): a workgroup's chunk of 256 list entries, 64 per wave; the read of chunk
  // + grid and the list entry of chunk + 2 grid are in flight while a chunk is worked on ----
  uint32_t f_chunk = blockIdx.x, f_sgi = blockIdx.x % in_nseg, f_depth = blockIdx.x / in_nseg;
  const uint32_t f_dsgi = gridDim.x % in_nseg, f_ddepth = gridDim.x / in_nseg;
  const bool fat_in = p.idx_in && p.in_stride == 4u;  // (seed_kernel's walk: index list two chunks deep, or entries with their reads)
  auto fetch_next = [&](uint32_t& r_out, uint32_t& L_out, uint64_t& rd_out) __attribute__((always_inline)) -> bool {
    bool act = false;
    r_out = 0;
    if (f_chunk < n_chunks) {
      const uint32_t sgi = __builtin_amdgcn_readfirstlane(f_sgi), t_base = __builtin_amdgcn_readfirstlane(f_depth * tile);
      const uint32_t t = t_base + tid;
      const auto* kp = kernel_args_here<SeedParams>();
      const uint32_t* idx_in = kp->idx_in;
      if (idx_in) {
        typedef const __attribute__((address_space(4))) uint32_t* const_u32_t;
        act = t < ((const_u32_t)(uintptr_t)kp->in_count)[sgi];
        if (fat_in) {
          if (act) {  // a list that carries its reads: the whole entry in one 16-byte load
            const uint4 e = reinterpret_cast<const uint4*>(idx_in)[(size_t)sgi * kp->in_seg_cap + t_base + tid];
            r_out = e.x;
            L_out = e.y;
            rd_out = (uint64_t)e.z | ((uint64_t)e.w << 32);
          }
        } else {
          const uint32_t* seg = idx_in + (size_t)sgi * kp->in_seg_cap + t_base;
          if (act) r_out = seg[tid];
        }
      } else {
        act = t < kp->n_total;
        r_out = act ? t : 0u;
      }
    }
    f_chunk += gridDim.x;
    f_sgi += f_dsgi;
    f_depth += f_ddepth;
    if (f_sgi >= in_nseg) {
      f_sgi -= in_nseg;
      ++f_depth;
    }
    return act;
  };

  uint32_t acc_v = 0;  // lane 2c / 2c + 1: reads offered to / claimed by member pass c (units in order, members in order)
  // diagnostics, per lane: of the unit being worked off (flushed per unit), and of the answers given inline in
  // the stream by bucket units / dictionary units (flushed once, to the first unit of that kind)
  uint32_t c_lookups = 0, c_cands = 0, c_bl = 0, c_bc = 0, c_dl = 0, c_dc = 0;
  // ---- a read is finished: the claim in cascade order, ONE entry decode, output, survivor list, counters ----
  auto finalize = [&](bool valid, uint32_t r, uint32_t L0, uint64_t rd, uint32_t el_mask, unsigned long long key) __attribute__((always_inline)) {
    const bool claimed = valid && key != ~0ull;
    const int32_t cp = claimed ? (int32_t)(key >> 56) : 255;
    uint32_t o_ref = 0, o_pos = 0;
    uint32_t tile_cnt = 0, c = 0, sel = 0;  // sel: unit << 2 | member of the claiming pass
    for (uint32_t ui = 0; ui < p.n_units; ++ui) {
      const KUnit& un = kargs->unit[ui];
      const bool el = valid && ((el_mask >> ui) & 1u) != 0u;
      for (uint32_t mi = 0; mi < un.n_members; ++mi, ++c) {
        const int32_t pi = un.m[mi].pass_index;
        const uint32_t n_off = (uint32_t)__popcll(__ballot(el && cp >= pi)), n_al = (uint32_t)__popcll(__ballot(cp == pi));
        tile_cnt = lane == 2u * c ? n_off : tile_cnt;
        tile_cnt = lane == 2u * c + 1u ? n_al : tile_cnt;
        sel = cp == pi ? (ui << 2) | mi : sel;
      }
    }
    acc_v += tile_cnt;
    if (claimed) {
      const uint32_t* ut = utab + (sel >> 2) * kUnitWords;
      const uint32_t flags = ut[UW_FLAGS];
      if (flags & (1u << 23)) {
        o_ref = (uint32_t)(key >> 21) & 0x7FFFFFFu;
        o_pos = (uint32_t)key & 0x1FFFFFu;
      } else {
        SegTables segs{reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGSTART)), reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGREF)),
                       reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_SEGOFF)), reinterpret_cast<const uint32_t*>(lds_pointer(ut, UW_CHUNKSEG)),
                       (flags >> 17) & 1u};
        uint32_t ref;
        locate_entry(segs, (uint32_t)(key >> 16), (uint32_t)key & 0xFFFFu, 255u, ref, o_pos);
        o_ref = ref - ut[UW_MEMBERS + 2u * (sel & 3u) + 1u];
      }
    }
    if (p.packed) {
      if (claimed) p.packed[r] = pack_assignment(cp, o_ref, o_pos, (uint32_t)((key >> 48) & 255u));
      else if (valid && !p.idx_out && !p.out_init) p.packed[r] = 0u;
    } else if (claimed) {
      p.pass_id[r] = (int8_t)cp;
      p.ref_id[r] = (int32_t)o_ref;
      p.pos[r] = (int32_t)o_pos;
      p.mm[r] = (uint8_t)((key >> 48) & 255u);
    } else if (valid && !p.idx_out && !p.out_init) {
      p.pass_id[r] = (int8_t)-1;
      p.ref_id[r] = -1;
      p.pos[r] = -1;
      p.mm[r] = 0;
    }
    if (p.idx_out) {
      const bool survive = valid && !claimed;
      const uint64_t mask = __ballot(survive);
      if (mask) {
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&ctl[0], (uint32_t)__popcll(mask));
        wbase = __shfl(wbase, 0, 64);
        if (survive)
          reinterpret_cast<uint4*>(p.idx_out)[(size_t)blockIdx.x * p.out_seg_cap + wbase + mbcnt(mask)] =
              make_uint4(r, L0, (uint32_t)rd, (uint32_t)(rd >> 32));
      }
    }
  };
  auto flush_diag = [&](uint32_t ui) __attribute__((always_inline)) {
    const uint64_t t_l = wave_sum(c_lookups), t_c = wave_sum(c_cands);
    if (lane == 0) {
      if (t_c) atomicAdd(&cnt[32u + 2u * ui], (unsigned long long)t_c);
      if (t_l) atomicAdd(&cnt[32u + 2u * ui + 1u], (unsigned long long)t_l);
    }
    c_lookups = 0;
    c_cands = 0;
  };

  // ---- the wave's row queue: rows of items, verified 64 at a time, whoever's they are ----
  uint32_t rpend = 0;  // rows waiting (wave-uniform, < 64 between pushes)
  auto verify_rows = [&](uint32_t from, uint32_t n, uint32_t cbase) __attribute__((always_inline)) {  // rows [from, from + n), n <= 64
    wave_lds_sync();
    if (lane < n) {
      const uint2 e = w.rows[from + lane];
      const uint32_t c = cbase + (e.y & 63u), ui = (e.y >> 11) & 3u, off = (e.y >> 13) & 63u, kp = (e.y >> 19) & 15u;
      const bool from_bucket = (e.y & kRowFromBucket) != 0u;
      const uint32_t* ut = utab + ui * kUnitWords;
      const uint32_t flags = ut[UW_FLAGS];
      uint64_t q;
      int32_t L;
      QH qh{nullptr, 0u};
      unit_view_t<LONG>(0, 255, (int32_t)((flags >> 16) & 1u), (int32_t)(flags & 255u), (int32_t)((flags >> 8) & 255u), w.rd[c], LONG ? w.rdh[c] : 0ull,
                 w.meta[c] & 255u, q, qh, L);
      const uint4* wide = reinterpret_cast<const uint4*>(lds_pointer(ut, from_bucket ? UW_BUCKETS : UW_SA16));
      if (e.y & kRowFromPair) {
        const uint64_t row = reinterpret_cast<const uint64_t*>(lds_pointer(ut, UW_BPAIR))[e.x];
        verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), q, L, off, 0u, &w.best[c], false, qh);
      } else if (wide) {
        verify_seed_row<LONG>(ut, wide[e.x], q, L, off, kp, &w.best[c], from_bucket, qh);
      } else {
        const uint64_t row = reinterpret_cast<const uint64_t*>(lds_pointer(ut, UW_SA))[e.x];
        verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), q, L, off, 0u, &w.best[c], false, qh);
      }
    }
  };
  // (tag: candidate lane | unit << 11 | offset << 13 | k' << 19 | bucket flag; n_rows <= kSeedRowsPerItem)
  auto push_rows = [&](uint32_t lo, uint32_t n_rows, uint32_t tag, uint32_t cbase) __attribute__((always_inline)) {
    uint32_t left = n_rows;
    while (__any(left != 0u)) {
      const uint32_t incl = wave_incl_scan(left);
      const uint32_t room = row_cap - rpend;
      const bool fits = left != 0u && incl <= room;
      const uint64_t fm = __ballot(fits);
      // (the lanes that fit are a prefix of the lanes with rows; after a drain rpend < 64, so the first one always fits)
      const uint32_t total = fm ? __shfl(incl, 63 - __clzll((long long)fm), 64) : 0u;
      if (fits) {
        const uint32_t first = rpend + incl - left;
        for (uint32_t i = 0; i < left; ++i) w.rows[first + i] = make_uint2(lo + i, tag);
        left = 0u;
      }
      rpend += total;
      while (rpend >= 64u) {
        rpend -= 64u;
        verify_rows(rpend, 64u, cbase);
      }
    }
  };

  // the wave's walk buffer: records (32 bytes: index, length | eligibility, read, best key so far) of the reads with a seed
  // left to its position list; full -> the seed is verified row by row after all
  uint4* const walk_buf = BUCKETS && p.walk_buf ? p.walk_buf + 2u * (size_t)(blockIdx.x * (kSeedThreads / 64u) + wv) * p.walk_cap : nullptr;
  uint32_t n_walk = 0;
  bool walk_room = BUCKETS && p.walk_buf != nullptr && p.walk_cap >= 64u;
  // ---- 64 parked candidates (or what is left at the end): items unit by unit, rows, the claim ----
  auto work_off = [&](uint32_t cbase, uint32_t n_cand) __attribute__((always_inline)) {
    wave_lds_sync();
    const bool valid = lane < n_cand;
    const uint32_t my_meta = valid ? w.meta[cbase + lane] : 0u;
    for (uint32_t ui = 0; ui < p.n_units; ++ui) {
      const KUnit& un = kargs->unit[ui];
      // (a unit with pair tables: three items per read in pair mode -- bit 3 of the read's four item bits)
      const uint32_t n_seeds = un.kind == 1u ? 1u : ((BUCKETS && un.bpair_anchor) ? 3u : (uint32_t)un.max_mm_seed + 1u);
      uint32_t n_it = 0;  // items of this unit waiting in the list (wave-uniform)
      // lane < m: item (candidate lane, seed) from the list
      auto run_items = [&](uint32_t from, uint32_t m) __attribute__((always_inline)) {
        wave_lds_sync();
        const bool has = lane < m;
        uint32_t cl = 0, j = 0;
        uint64_t q = 0;
        QH qh{nullptr, 0u};
        int32_t L = 0;
        bool pair_mode = false;
        if (has) {
          const uint32_t e = w.items[from + lane];
          cl = e & 63u;
          j = e >> 6;
          const uint32_t cm = w.meta[cbase + cl];
          pair_mode = BUCKETS && ((cm >> (16u + 4u * ui + 3u)) & 1u) != 0u;
          unit_view_t<LONG>(un.min_len, un.max_len, un.poly_t, un.trim5, un.trim3, w.rd[cbase + cl], LONG ? w.rdh[cbase + cl] : 0ull, cm & 255u, q, qh, L);
        }
        if (un.kind == 1u) {
          if (has) {
            // a read already left to the end of the kernel (a seed with a position list in a unit in front): its probe goes
            // there too -- the pass in front may claim it; so does a probe whose FM fallback meets a wide interval
            if (BUCKETS && walk_room && (w.meta[cbase + cl] & kMetaWalkLater)) {
              atomicOr(&w.meta[cbase + cl], 1u << (kMetaDictLaterShift + ui));
            } else {
              bool wide = false;
              const unsigned long long key = dict_unit_probe<LONG>(un, q, L, false, make_uint4(0u, 0u, 0u, 0u), c_lookups, c_cands,
                                                                   (BUCKETS && walk_room) ? &wide : nullptr, qh);
              if (wide) atomicOr(&w.meta[cbase + cl], kMetaWalkLater | (1u << (kMetaDictLaterShift + ui)));
              else if (key != ~0ull) atomicMin(&w.best[cbase + cl], key);
            }
          }
          return;
        }
        // ---- a seed: bucket count or jump-table load, then its rows into the row queue ----
        uint32_t lo = 0, n_rows = 0, tag = 0;
        bool listed = false;  // the seed's k-mer has a position list: answered behind the unit's other items
        if (BUCKETS && has && pair_mode) {
          // item j = anchor pair (0,1), (1,2), (0,2): one mismatch in the seed region leaves one of them clean
          const uint32_t A = un.bpair_anchor, kb = 2u * A, amask = (1u << kb) - 1u, n_codes1 = (1u << (2u * kb)) + 1u;
          const uint32_t i0 = j == 1u ? 1u : 0u, i1 = j == 0u ? 1u : 2u;
          const uint32_t a0 = (uint32_t)(q >> (i0 * kb)) & amask, a1 = (uint32_t)(q >> (i1 * kb)) & amask;
          const uint32_t t = i1 - i0 - 1u;
          const TableEntry2 te = *reinterpret_cast<const TableEntry2*>(un.bpair_jump + t * n_codes1 + (a0 | (a1 << kb)));
          lo = te.lo + (t ? un.bpair_row_off[1] : un.bpair_row_off[0]);
          n_rows = te.hi - te.lo;
          ++c_lookups;
          c_cands += n_rows;
          tag = cl | (ui << 11) | ((i0 * A) << 13) | kRowFromPair;
        } else if (has) {
          const int32_t k = unit_seed_bases<LONG>(un, L);
          const uint32_t off = j * (uint32_t)k;
          uint32_t kp = 0, bcnt = kSeedBucketOverflow;
          if (BUCKETS && un.buckets && (uint32_t)k == un.bucket_k) {
            kp = un.bucket_k;
            lo = ((uint32_t)(q >> (2u * off)) & ((1u << (2u * kp)) - 1u)) * kSeedBucketRows;
            const uint4 hdr = un.buckets[lo];
            bcnt = (hdr.y >> 12) & 15u;
            ++c_lookups;
            // an overflowing k-mer with a position list, in a unit where a walk in text order may stop early (ONE
            // library, one seed mismatch, the whole read is the seed region): answered behind the unit's other items
            if (walk_room && seed_is_listed(un, hdr, L)) {
              listed = true;
              atomicOr(&w.meta[cbase + cl], kMetaWalkLater);
            }
          }
          if (listed) {
            n_rows = 0u;
          } else if (bcnt != kSeedBucketOverflow) {
            n_rows = bcnt;
            tag = cl | (ui << 11) | (off << 13) | (kp << 19) | kRowFromBucket;
          } else {
            uint32_t tab_off = 0, shift = 0;
            const uint32_t K = un.tabs.k[0] ? pick_seed_table(un.tabs, k, tab_off, shift) : 0u;
            uint32_t hi = un.n + 1u;
            lo = 0;
            kp = 0;
            if (K) {
              kp = min(K, (uint32_t)k);
              const uint32_t* tab = un.ftab + tab_off + (lex_code((q >> (2u * off)) & low_bits(2u * kp), kp) << shift);
              lo = tab[0];
              hi = tab[1u << shift];
            }
            ++c_lookups;
            n_rows = hi > lo ? hi - lo : 0u;
            tag = cl | (ui << 11) | (off << 13) | (kp << 19);
          }
          c_cands += n_rows;
        }
        // wide intervals (a poly-A seed matches 10^5 rows): the whole wave verifies one after the other
        uint64_t wide_m = __ballot(n_rows > kSeedRowsPerItem);
        while (wide_m) {
          const int src = __ffsll((long long)wide_m) - 1;
          wide_m &= wide_m - 1ull;
          const uint32_t w_lo = __shfl(lo, src, 64), w_n = __shfl(n_rows, src, 64), w_tag = __shfl(tag, src, 64);
          const uint64_t w_q = __shfl((unsigned long long)q, src, 64);
          const QH w_qh{LONG ? reinterpret_cast<const uint64_t*>((uintptr_t)__shfl((unsigned long long)(uintptr_t)qh.p, src, 64)) : nullptr, qh.sh};
          const int32_t w_L = __shfl(L, src, 64);
          const uint32_t* ut = utab + ui * kUnitWords;
          unsigned long long* slot = &w.best[cbase + (w_tag & 63u)];
          for (uint32_t i = lane; i < w_n; i += 64u) {
            if (BUCKETS && (w_tag & kRowFromPair)) {
              const uint64_t row = un.bpair_rows[w_lo + i];
              verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), w_q, w_L, (w_tag >> 13) & 63u, 0u, slot, false, w_qh);
            } else if (un.sa16) {
              verify_seed_row<LONG>(ut, un.sa16[w_lo + i], w_q, w_L, (w_tag >> 13) & 63u, (w_tag >> 19) & 15u, slot, false, w_qh);
            } else {
              const uint64_t row = un.sa[w_lo + i];
              verify_seed_row<LONG>(ut, make_uint4((uint32_t)row, (uint32_t)(row >> 32), 0u, 0u), w_q, w_L, (w_tag >> 13) & 63u, 0u, slot, false, w_qh);
            }
          }
        }
        push_rows(lo, n_rows > kSeedRowsPerItem ? 0u : n_rows, tag, cbase);
      };
      for (uint32_t j = 0; j <= n_seeds; ++j) {  // (the last trip only works off what is left in the list)
        if (j < n_seeds) {
          const bool need = ((my_meta >> (16u + 4u * ui + j)) & 1u) != 0u;
          const uint64_t mask = __ballot(need);
          if (need) w.items[n_it + mbcnt(mask)] = (uint16_t)(lane | (j << 6));
          n_it += (uint32_t)__popcll(mask);
        }
        if (n_it >= 64u || (j == n_seeds && n_it)) {
          const uint32_t m = min(n_it, 64u);
          n_it -= m;
          run_items(n_it, m);
        }
      }
      flush_diag(ui);
    }
    if (rpend) {
      verify_rows(0u, rpend, cbase);
      rpend = 0u;
    }
    wave_lds_sync();
    // a candidate with a seed left to its position list (run_items): everything else about it is known -- its record goes
    // to the wave's walk buffer, the walk and its claim happen behind the stream (the end of this kernel)
    const bool later = BUCKETS && valid && (w.meta[cbase + lane] & kMetaWalkLater) != 0u;
    if (BUCKETS) {
      const uint64_t lm = __ballot(later);
      if (lm) {
        if (later) {
          uint4* rec = walk_buf + 2u * (size_t)(n_walk + mbcnt(lm));
          const unsigned long long bk = w.best[cbase + lane], rdv = w.rd[cbase + lane];
          rec[0] = make_uint4(w.r[cbase + lane], (my_meta & 0xFFFFu) | (w.meta[cbase + lane] & (7u << kMetaDictLaterShift)), (uint32_t)rdv,
                              (uint32_t)(rdv >> 32));
          const unsigned long long rh = LONG ? w.rdh[cbase + lane] : 0ull;
          rec[1] = make_uint4((uint32_t)bk, (uint32_t)(bk >> 32), (uint32_t)rh, (uint32_t)(rh >> 32));
        }
        n_walk += (uint32_t)__popcll(lm);
        walk_room = n_walk + 64u <= p.walk_cap;
      }
    }
    finalize(valid && !later, valid ? w.r[cbase + lane] : 0u, my_meta & 255u, valid ? w.rd[cbase + lane] : 0ull, (my_meta >> 8) & 255u,
             valid ? w.best[cbase + lane] : ~0ull);
  };

  uint32_t r_b = 0, r_c = 0, L_b = 255u;
  uint64_t rd_b = 0;
  uint32_t pend = 0;  // candidates parked (wave-uniform, < 64 between chunks)
  uint32_t pre_ui = kSeedMaxUnits;  // the first dictionary unit (none: kSeedMaxUnits)
  for (uint32_t ui = p.n_units; ui-- > 0u;)
    if (kargs->unit[ui].kind == 1u) pre_ui = ui;
  uint32_t L_unused = 0;
  uint64_t rd_unused = 0;
  bool act_b = fetch_next(r_b, L_b, rd_b);
  bool act_c = false;
  if (!fat_in) {
    act_c = fetch_next(r_c, L_unused, rd_unused);
    if (act_b) {
      L_b = p.lens[r_b];
      rd_b = p.reads[r_b];
    }
  }
  // (one trip more than there are chunks: the last one has no reads and works off what is still parked)
  for (uint32_t chunk = blockIdx.x;; chunk += gridDim.x) {
    const bool last_trip = chunk >= n_chunks;
    const bool active = act_b;
    const uint32_t r = r_b, L0 = active ? L_b : 255u;
    const uint64_t rd = rd_b;
    // (a long read's second word: gathered here, where it is needed -- the lists carry the first word only)
    const uint64_t rdh = (LONG && active && L0 > 32u) ? (uint64_t)(uintptr_t)(p.reads_hi + r) : 0ull;  // (where it lies: QH)
    L_b = 255u;
    rd_b = 0;
    if (fat_in) {
      act_b = fetch_next(r_b, L_b, rd_b);
    } else {
      act_b = act_c;
      r_b = r_c;
      if (act_b) {
        L_b = p.lens[r_b];
        rd_b = p.reads[r_b];
      }
      act_c = fetch_next(r_c, L_unused, rd_unused);
    }
    // ================= the stream: eligibility, filters, inline answers =================
    unsigned long long my_best = ~0ull;
    uint32_t el_mask = 0u, queued_all = 0u;
    // the home slot of the first dictionary unit is requested before anything else, so that its trip runs
    // beside the bucket / bitmap trips of the units in front of it (when enough lanes ask: see below)
    uint4 pre_sl = make_uint4(0u, 0u, 0u, 0u);
    bool pre_direct = false;
    if (pre_ui < p.n_units) {
      const KUnit& un = kargs->unit[pre_ui];
      uint64_t q = 0;
        QH qh{nullptr, 0u};
      int32_t L = 0;
      const bool el = active && unit_view_t<LONG>(un.min_len, un.max_len, un.poly_t, un.trim5, un.trim3, rd, rdh, L0, q, qh, L);
      const bool search = el && L > 0;
      if ((uint32_t)__popcll(__ballot(search)) >= 16u) {
        const uint32_t kmask = un.key_bases >= 16u ? 0xFFFFFFFFu : ((1u << (2u * un.key_bases)) - 1u);
        pre_direct = search && (uint32_t)L >= un.key_bases;
        if (pre_direct) {
          // (one 16-byte load: left to itself the compiler loads the meta half, tests the chain, then loads the window)
          const uint4* sp = un.slots + ((((uint32_t)q & kmask) * kDictHashMul) >> (32u - un.log2_slots));
          typedef uint32_t u4v __attribute__((ext_vector_type(4)));
          const u4v v = *reinterpret_cast<const volatile u4v*>(sp);
          pre_sl = make_uint4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    for (uint32_t ui = 0; ui < p.n_units; ++ui) {
      const KUnit& un = kargs->unit[ui];
      uint64_t q = 0;
        QH qh{nullptr, 0u};
      int32_t L = 0;
      const bool el = active && unit_view_t<LONG>(un.min_len, un.max_len, un.poly_t, un.trim5, un.trim3, rd, rdh, L0, q, qh, L);
      el_mask |= el ? 1u << ui : 0u;
      const int32_t V = un.max_mm_seed;
      const bool search = el && L > V;
      uint32_t queued = 0u;
      if (un.kind == 1u) {
        // a dictionary unit: when enough lanes ask, the home slot was requested at the top of the trip (one
        // load for the wave) and a chain of one slot is answered here; longer chains, overflowed homes and
        // reads shorter than the key are parked.  A unit few reads are offered to (pre-tRNA: the poly-T
        // rule) parks them all: their slot loads then run with dense lanes.  (Only the first dictionary
        // unit of a launch is answered inline.)
        queued = search ? 1u : 0u;
        if (ui == pre_ui && pre_direct) {
          ++c_dl;
          const uint32_t chain = (pre_sl.w >> kDictChainShift) & kDictChainMask;
          if (LONG && L > 32) {
            // a long read: the table cannot answer it (dict_unit_probe), but it can say NO -- an alignment needs a position
            // whose 32-base window agrees with the read's first 32 bases under the pass's rules, and every such window sits in
            // a slot of this chain; only a read with such a slot (or an overflowed home) is parked for the FM search
            if (chain != kDictChainOverflow && !dict_window_hit(un, q, pre_sl)) queued = 0u;
          } else if (chain != kDictChainOverflow) {
            // (the home slot is in hand: the probe neither loads nor counts it again; the few lanes whose key
            // chains further walk their chain here -- parked, they would leave the stream's order)
            uint32_t dl = 0;
            my_best = min(my_best, dict_unit_probe<LONG>(un, q, L, true, pre_sl, dl, c_dc, nullptr, qh));
            c_dl += dl;
            queued = 0u;
          }
        }
      } else {
        const uint32_t n_seeds = (uint32_t)V + 1u;
        const int32_t k = unit_seed_bases<LONG>(un, L);
        queued = search ? (1u << n_seeds) - 1u : 0u;  // seeds that need the index
        // a seed region of 3 A .. 4 A - 1 bases in a large library: the three anchor pairs instead of two seeds of
        // 8..9 bases (40..170 rows each in 11 Mbp); parked, the pair lookups run with dense lanes
        if (BUCKETS && un.bpair_anchor && search) {
          const int32_t R = min(L, un.min_seed_len);
          if (R >= (int32_t)(3u * un.bpair_anchor) && R < (int32_t)(4u * un.bpair_anchor)) queued = 15u;
        }
        if (BUCKETS && un.buckets && !(queued & 8u)) {
          // a seed of exactly bucket_k bases: the first four rows of both seeds' buckets are requested together
          // (one 64-byte half line per seed) and verified here; a fuller or overflowing bucket is parked
          const bool inl = search && (uint32_t)k == un.bucket_k;
          if (__any(inl)) {
            const uint32_t* ut = utab + ui * kUnitWords;
            const uint32_t cmask = (1u << (2u * un.bucket_k)) - 1u;
            const uint32_t base0 = ((uint32_t)q & cmask) * kSeedBucketRows;
            const uint32_t base1 = ((uint32_t)(q >> (2u * un.bucket_k)) & cmask) * kSeedBucketRows;
            const bool two = V >= 1;
            uint4 ra[4], rb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              ra[i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
              rb[i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
            }
            if (inl) {
#pragma unroll
              for (int i = 0; i < 4; ++i) ra[i] = un.buckets[base0 + i];
              if (two) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[i] = un.buckets[base1 + i];
              }
            }
            // the first four rows of each bucket are verified here; a fuller bucket (5..8 rows: one seed in
            // eight) gets its second half line in a second trip of the lanes that need one -- parked
            // instead, those reads would leave the stream's order (the next launch's gathers then fetch
            // their lines twice: 14 M more L2 misses in the 2-mismatch pass behind this launch); only an
            // overflowing bucket (the jump table serves that k-mer) is parked
            bool more0 = false, more1 = false;
            if (inl) {
              const uint32_t cnt0 = (ra[0].y >> 12) & 15u, cnt1 = two ? ((rb[0].y >> 12) & 15u) : 0u;
              c_bl += two ? 2u : 1u;
              queued = 0u;
              if (cnt0 != kSeedBucketOverflow) {
                c_bc += cnt0;
#pragma unroll
                for (int i = 0; i < 4; ++i) my_best = min(my_best, seed_row_key<LONG>(ut, ra[i], q, L, 0u, un.bucket_k, true, qh));
                more0 = cnt0 > 4u;
              } else {
                queued |= 1u;
              }
              if (two) {
                if (cnt1 != kSeedBucketOverflow) {
                  c_bc += cnt1;
#pragma unroll
                  for (int i = 0; i < 4; ++i) my_best = min(my_best, seed_row_key<LONG>(ut, rb[i], q, L, un.bucket_k, un.bucket_k, true, qh));
                  more1 = cnt1 > 4u;
                } else {
                  queued |= 2u;
                }
              }
            }
            if (__any(more0 || more1)) {
              if (more0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i] = un.buckets[base0 + 4u + i];
              }
              if (more1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[i] = un.buckets[base1 + 4u + i];
              }
              if (more0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) my_best = min(my_best, seed_row_key<LONG>(ut, ra[i], q, L, 0u, un.bucket_k, true, qh));
              }
              if (more1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) my_best = min(my_best, seed_row_key<LONG>(ut, rb[i], q, L, un.bucket_k, un.bucket_k, true, qh));
              }
            }
          }
        }
        // the unit's presence bitmaps: both seeds' words are requested before either is looked at
        if (un.kbits && k >= 8 && !(queued & 8u)) {
          const uint32_t kb = (uint32_t)min(k, 11), woff = seed_kbits_word_off(kb), cmask = (1u << (2u * kb)) - 1u;
          const uint32_t code0 = (uint32_t)q & cmask, code1 = (uint32_t)(q >> (2u * (uint32_t)k)) & cmask;
          uint32_t w0 = ~0u, w1 = ~0u;
          if (queued & 1u) w0 = un.kbits[woff + (code0 >> 5)];
          if (queued & 2u) w1 = un.kbits[woff + (code1 >> 5)];
          if (!((w0 >> (code0 & 31u)) & 1u)) queued &= ~1u;
          if (!((w1 >> (code1 & 31u)) & 1u)) queued &= ~2u;
        }
      }
      queued_all |= queued << (4u * ui);  // (four bits per unit: three items, pair mode)
    }
    // ================= finished on the spot, or parked =================
    const bool park = queued_all != 0u;
    finalize(active && !park, r, L0, rd, el_mask, my_best);
    const uint64_t pm = __ballot(park);
    if (pm) {
      const uint32_t c = pend + mbcnt(pm);
      if (park) {
        w.rd[c] = rd;
        if (LONG) w.rdh[c] = rdh;
        w.best[c] = my_best;
        w.r[c] = r;
        w.meta[c] = L0 | (el_mask << 8) | (queued_all << 16);
      }
      pend += (uint32_t)__popcll(pm);
    }
    if (pend >= 64u || (last_trip && pend)) {
      const uint32_t m = min(pend, 64u);
      pend -= m;
      work_off(pend, m);
    }
    if (last_trip) break;
  }
  // ---- the reads left to their position lists, 64 records a round: the walks one read after the other by the whole wave
  // (pos_list_answer), then the claim of the 64 ----
  if (BUCKETS && n_walk) {
    __threadfence();
    if (lane == 0u && p.walk_diag) atomicAdd((unsigned long long*)&p.stats[(uint32_t)kargs->unit[0].m[0].pass_index * 5u + 2u], (unsigned long long)n_walk);
    for (uint32_t base = 0; base < n_walk; base += 64u) {
      const uint32_t m = min(64u, n_walk - base);
      wave_lds_sync();
      if (lane < m) {
        const uint4 a = walk_buf[2u * (size_t)(base + lane)], b = walk_buf[2u * (size_t)(base + lane) + 1u];
        w.r[lane] = a.x;
        w.meta[lane] = a.y;
        w.rd[lane] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
        if (LONG) w.rdh[lane] = (unsigned long long)b.z | ((unsigned long long)b.w << 32);
        w.best[lane] = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
      }
      wave_lds_sync();
      for (uint32_t i = 0; i < m; ++i) {
        const uint32_t cm = w.meta[i];
        const unsigned long long rdv = w.rd[i], rdhv = LONG ? w.rdh[i] : 0ull;
        for (uint32_t ui = 0; ui < p.n_units; ++ui) {
          const KUnit& un = kargs->unit[ui];
          if (!((cm >> (8u + ui)) & 1u)) continue;
          uint64_t q = 0;
        QH qh{nullptr, 0u};
          int32_t L = 0;
          unit_view_t<LONG>(un.min_len, un.max_len, un.poly_t, un.trim5, un.trim3, rdv, rdhv, cm & 255u, q, qh, L);
          if (un.kind == 1u) {
            // a dictionary probe left to the end: not needed once a pass in front has the read; else the slots as ever
            // (every lane the same loads), and a wide FM interval by the whole wave
            if (!((cm >> (kMetaDictLaterShift + ui)) & 1u) || (w.best[i] >> 56) < (unsigned long long)un.m[0].pass_index) continue;
            bool wide = false;
            uint32_t dl = 0, dc = 0;
            unsigned long long key = dict_unit_probe<LONG>(un, q, L, false, make_uint4(0u, 0u, 0u, 0u), dl, dc, &wide, qh);
            if (wide) key = dict_fallback_wave<LONG>(un, q, L, &dc, qh);
            if (lane == 0u) {
              c_lookups += dl;
              c_cands += dc;
              if (key < w.best[i]) w.best[i] = key;
            }
            wave_lds_sync();
            flush_diag(ui);
            continue;
          }
          if (!un.pos_rows) continue;
          if (L <= un.max_mm_seed || (uint32_t)unit_seed_bases<LONG>(un, L) != un.bucket_k) continue;
          const unsigned long long cur = w.best[i];
          // an earlier pass has the read, or this one has it without a mismatch (found by a seed verified completely: every
          // exact alignment lies in its rows): nothing here can be better
          if ((cur >> 48) <= ((unsigned long long)un.m[0].pass_index << 8)) continue;
          const uint32_t cmask = (1u << (2u * un.bucket_k)) - 1u;
          const uint4 h0 = un.buckets[((uint32_t)q & cmask) * kSeedBucketRows];
          const uint4 h1 = un.buckets[((uint32_t)(q >> (2u * un.bucket_k)) & cmask) * kSeedBucketRows];
          const uint2 none = make_uint2(0u, 0u);
          const uint2 s0 = seed_is_listed(un, h0, L) ? make_uint2(h0.z, h0.w) : none, s1 = seed_is_listed(un, h1, L) ? make_uint2(h1.z, h1.w) : none;
          if (!(s0.y | s1.y)) continue;
          const unsigned long long key = p.walk_diag >= 2u ? ~0ull : pos_list_answer(&un, utab + ui * kUnitWords, q, L, s0, s1, &c_cands);
          if (lane == 0u && key < cur) w.best[i] = key;
          wave_lds_sync();
          flush_diag(ui);
        }
      }
      wave_lds_sync();
      const bool valid = lane < m;
      const uint32_t mt = valid ? w.meta[lane] : 0u;
      finalize(valid, valid ? w.r[lane] : 0u, mt & 255u, valid ? w.rd[lane] : 0ull, (mt >> 8) & 255u, valid ? w.best[lane] : ~0ull);
    }
  }
  // ---- counters: one global atomic per non-zero counter and workgroup ----
  {
    // (the inline answers' lookups / candidates: to the first bucket unit / the first dictionary unit)
    uint32_t b_unit = 0, d_unit = 0;
    for (uint32_t ui = p.n_units; ui-- > 0u;) {
      if (kargs->unit[ui].kind == 0u && kargs->unit[ui].buckets) b_unit = ui;
      if (kargs->unit[ui].kind == 1u) d_unit = ui;
    }
    c_lookups = c_bl;
    c_cands = c_bc;
    flush_diag(b_unit);
    c_lookups = c_dl;
    c_cands = c_dc;
    flush_diag(d_unit);
    uint32_t c = 0, cslot = 0;
    for (uint32_t ui = 0; ui < p.n_units; ++ui)
      for (uint32_t mi = 0; mi < kargs->unit[ui].n_members; ++mi, ++c) {
        const uint32_t pi = (uint32_t)kargs->unit[ui].m[mi].pass_index;
        cslot = (lane >> 1) == c ? 2u * pi + (lane & 1u) : cslot;
      }
    if (lane < 2u * c && acc_v) atomicAdd(&cnt[cslot], (unsigned long long)acc_v);
  }
  __syncthreads();
  if (tid < 32u) {
    const unsigned long long v = cnt[tid];
    if (v) atomicAdd((unsigned long long*)&p.stats[(tid >> 1) * 5u + (tid & 1u)], v);
  } else if (tid < 32u + 2u * p.n_units) {
    const uint32_t ui = (tid - 32u) >> 1, what = (tid - 32u) & 1u;  // 0 candidates, 1 lookups
    const unsigned long long v = cnt[tid];
    if (v) atomicAdd((unsigned long long*)&p.stats[(uint32_t)kargs->unit[ui].m[0].pass_index * 5u + 3u + what], v);
  }
  if (p.idx_out && tid == 0) p.out_count[blockIdx.x] = ctl[0];
}

// ---------------------------------------------------------------------------
// pair_wave_kernel: the 2-mismatch pass (`-5 1 -3 2 -v 2 --best`, RAP:585/599) through the anchor pairs
// of stratum_kernel (kernels.hip, fm_index.hpp: PairTables), for one-word reads without N, every wave on
// its own.  stratum_kernel holds 64 reads per wave and walks the six pairs with all of them: a pair's
// jump-table load, row compaction, verification and stratum test are issued for the wave as long as ONE
// lane is still open, and after the first pair 45 % of the lanes are closed (1 400 VALU lane-instructions
// per read for 3.4 lookups and 3.9 rows).  Here a wave takes 256 reads per trip into its LDS region
// (four per lane) and works in three rounds -- pair (0,1) for every read; (2,3) for the reads whose best
// is worse than exact; the other four for the reads still worse than one mismatch (bowtie's --best is
// stratum first: a hit below the bound is final) -- with the ITEMS (read, pair) of a round compacted
// over the wave, 64 lookups per trip, and their ROWS through a row queue, 64 verifications per trip.
// Same tables, same candidates, same assignment (fewest mismatches, lowest text position).
// ---------------------------------------------------------------------------
namespace {

constexpr uint32_t kPairReads = 256u;   // reads of a wave per trip
constexpr uint32_t kPairItems = 192u;   // item list (fewer than 128 pending + the 64 of one sweep over the lanes)
constexpr uint32_t kPairRowCap = 160u;  // row queue
constexpr uint32_t kPairSharedWords = 16u;  // ctl[4], counters [5] x 8 bytes

struct PairLds {
  unsigned long long* rd;    // [kPairReads] trimmed read
  unsigned long long* best;  // [kPairReads] mm : 8 | text position : 32 | segment : 16 | before : 8, ~0 = none
  uint8_t* len;              // [kPairReads] trimmed length, 0 = not searched
  uint8_t* st;               // [kPairReads] anchor length (bits 0-2), bit 6 = no anchors fit (scan), bit 7 = a read sits here
  uint2* rows;               // [kPairRowCap] (row index, slot | need_before << 8)
  uint16_t* items;           // [kPairItems] slot | pair << 8
};
__host__ __device__ constexpr uint32_t pair_wave_lds_bytes() { return kPairReads * (8u + 8u + 1u + 1u) + kPairRowCap * 8u + kPairItems * 2u; }

}  // namespace

// LONG: the batch holds reads of 33..63 nt (p.reads_hi): a read whose trimmed length is still beyond 32 bases is counted as
// offered and not searched -- the planner lets such a read in only when the library's longest entry is shorter
template <bool LONG>
__global__ void __launch_bounds__(kSeedThreads, 5) pair_wave_kernel(const MatchParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  uint32_t* const ctl = smem;  // [0] survivors of the workgroup, [1] longest input segment
  unsigned long long* const wg_cnt = reinterpret_cast<unsigned long long*>(smem + 4);
  PairLds w;
  {
    uint8_t* base = reinterpret_cast<uint8_t*>(smem + kPairSharedWords) + (size_t)wv * pair_wave_lds_bytes();
    w.rd = reinterpret_cast<unsigned long long*>(base);
    w.best = w.rd + kPairReads;
    w.rows = reinterpret_cast<uint2*>(w.best + kPairReads);
    w.items = reinterpret_cast<uint16_t*>(w.rows + kPairRowCap);
    w.len = reinterpret_cast<uint8_t*>(w.items + kPairItems);
    w.st = w.len + kPairReads;
  }
  if (tid < 4u) ctl[tid] = 0u;
  if (tid < 5u) wg_cnt[tid] = 0ull;
  __syncthreads();
  if (p.idx_in) longest_segment(p.in_count, p.in_nseg, &ctl[1]);
  __syncthreads();

  const uint32_t PA = p.pair_anchor;
  const uint32_t in_nseg = p.idx_in ? p.in_nseg : 1u;
  const uint32_t depth_chunks = p.idx_in ? (ctl[1] + kPairReads - 1u) / kPairReads : (p.n_total + kPairReads - 1u) / kPairReads;
  const uint32_t n_chunks = in_nseg * depth_chunks;
  SegTables segs{p.seg_start, p.seg_ref, p.seg_off, p.chunk_seg, p.simple_segs};
  uint32_t c_processed = 0, c_aligned = 0, c_cands = 0, c_lookups = 0;
  uint32_t rpend = 0;  // rows waiting in the queue (wave-uniform)

  // A row of a pair list (or of the suffix array) as a candidate of the read in `slot`, in two steps so that a lane can
  // have the loads of TWO candidates in flight (a wave works through dependent L2 trips -- row, then text window --
  // and was waiting for them half of its time): row_start checks the room and says where the alignment starts,
  // row_finish compares the text window.
  auto row_start = [&](uint64_t row, uint32_t slot, uint32_t need_before, uint32_t& s) __attribute__((always_inline)) -> bool {
    const uint32_t before = (uint32_t)(row >> 32) & 255u, after = (uint32_t)(row >> 40) & 255u;
    const uint32_t need_after = (uint32_t)w.len[slot] - need_before;
    s = (uint32_t)row - need_before;
    return !((need_before > before) | (need_after > after));
  };
  auto row_finish = [&](uint64_t row, uint32_t slot, uint32_t need_before, uint32_t s, uint64_t win) __attribute__((always_inline)) {
    const uint64_t q = w.rd[slot];
    const int32_t L = (int32_t)w.len[slot];
    const uint64_t m = mismatch_bits(win, q) & low_bits(2u * (uint32_t)L);
    const uint32_t mm_total = (uint32_t)__popcll(m);
    uint32_t mm_seed = mm_total;
    if (L > p.seed_len) mm_seed = (uint32_t)__popcll(m & low_bits(2u * (uint32_t)p.seed_len));
    if (((int32_t)mm_seed > p.max_mm_seed) | ((int32_t)mm_total > p.max_mm_total)) return;
    const uint32_t before = (uint32_t)(row >> 32) & 255u;
    const uint32_t rbefore = before < 255u ? before - need_before : 255u;
    const unsigned long long key = ((unsigned long long)mm_total << 56) | ((unsigned long long)s << 24) | ((unsigned long long)(row >> 48) << 8) |
                                   (unsigned long long)rbefore;
    atomicMin(&w.best[slot], key);
  };
  auto verify_one = [&](uint64_t row, uint32_t slot, uint32_t need_before) __attribute__((always_inline)) {
    uint32_t s;
    if (row_start(row, slot, need_before, s)) row_finish(row, slot, need_before, s, text_window(p.text, s));
  };
  auto verify_rows = [&](uint32_t from, uint32_t n) __attribute__((always_inline)) {  // rows [from, from + n), n <= 128: two per lane
    wave_lds_sync();
    const bool h0 = lane < n, h1 = lane + 64u < n;
    uint2 e0 = make_uint2(0u, 0u), e1 = make_uint2(0u, 0u);
    if (h0) e0 = w.rows[from + lane];
    if (h1) e1 = w.rows[from + 64u + lane];
    uint64_t r0 = 0, r1 = 0;
    if (h0) r0 = p.pair_rows[e0.x];
    if (h1) r1 = p.pair_rows[e1.x];
    uint32_t s0 = 0, s1 = 0;
    const bool ok0 = h0 && row_start(r0, e0.y & 255u, e0.y >> 8, s0);
    const bool ok1 = h1 && row_start(r1, e1.y & 255u, e1.y >> 8, s1);
    uint64_t w0 = 0, w1 = 0;
    if (ok0) w0 = text_window(p.text, s0);
    if (ok1) w1 = text_window(p.text, s1);
    if (ok0) row_finish(r0, e0.y & 255u, e0.y >> 8, s0, w0);
    if (ok1) row_finish(r1, e1.y & 255u, e1.y >> 8, s1, w1);
  };
  auto push_rows = [&](uint32_t lo, uint32_t n_rows, uint32_t tag) __attribute__((always_inline)) {
    uint32_t left = n_rows;
    while (__any(left != 0u)) {
      const uint32_t incl = wave_incl_scan(left);
      const uint32_t room = kPairRowCap - rpend;
      const bool fits = left != 0u && incl <= room;
      const uint64_t fm = __ballot(fits);
      const uint32_t total = fm ? __shfl(incl, 63 - __clzll((long long)fm), 64) : 0u;
      if (fits) {
        const uint32_t first = rpend + incl - left;
        for (uint32_t i = 0; i < left; ++i) w.rows[first + i] = make_uint2(lo + i, tag);
        left = 0u;
      }
      rpend += total;
      while (rpend >= 128u) {
        rpend -= 128u;
        verify_rows(rpend, 128u);
      }
    }
  };
  // items [from, from + m), m <= 128, two per lane: (slot, pair) -> its jump-table entry (both loads in flight) -> its
  // rows into the queue
  auto run_items = [&](uint32_t from, uint32_t m) __attribute__((always_inline)) {
    wave_lds_sync();
    uint32_t lo[2] = {0u, 0u}, n_rows[2] = {0u, 0u}, tag[2] = {0u, 0u};
    const TableEntry2* tp[2] = {nullptr, nullptr};
    uint32_t roff[2] = {0u, 0u};
#pragma unroll
    for (uint32_t h = 0; h < 2u; ++h) {
      if (lane + 64u * h < m) {
        const uint32_t e = w.items[from + 64u * h + lane];
        const uint32_t slot = e & 255u, pr = e >> 8;
        // (i, j) = (0,1) (2,3) (1,2) (0,2) (1,3) (0,3): table j - i - 1
        const uint32_t i = (0x010120u >> (4u * pr)) & 15u, j = (0x332231u >> (4u * pr)) & 15u;
        const uint64_t q = w.rd[slot];
        const uint32_t A = w.st[slot] & 7u;
        const bool second = A != PA;
        const uint32_t kb = 2u * A, amask = (1u << kb) - 1u, n_codes1 = (1u << (2u * kb)) + 1u;
        const uint32_t ai = (uint32_t)(q >> (i * kb)) & amask, aj = (uint32_t)(q >> (j * kb)) & amask;
        const uint32_t t = j - i - 1u;
        tp[h] = reinterpret_cast<const TableEntry2*>((second ? p.pair_jump_s : p.pair_jump) + t * n_codes1 + (ai | (aj << kb)));
        const uint32_t r0 = second ? p.pair_row_off_s[0] : p.pair_row_off[0], r1 = second ? p.pair_row_off_s[1] : p.pair_row_off[1],
                       r2 = second ? p.pair_row_off_s[2] : p.pair_row_off[2];
        roff[h] = t == 0u ? r0 : (t == 1u ? r1 : r2);
        tag[h] = slot | ((i * A) << 8);
      }
    }
    TableEntry2 te[2] = {{0u, 0u}, {0u, 0u}};
    if (tp[0]) te[0] = *tp[0];
    if (tp[1]) te[1] = *tp[1];
#pragma unroll
    for (uint32_t h = 0; h < 2u; ++h) {
      if (tp[h]) {
        lo[h] = te[h].lo + roff[h];
        n_rows[h] = te[h].hi - te[h].lo;
        ++c_lookups;
        c_cands += n_rows[h];
      }
    }
    // a key of a repeat family can have hundreds of rows: the whole wave verifies those, one item after the other
    for (uint32_t h = 0; h < 2u; ++h) {
      uint64_t wide_m = __ballot(n_rows[h] > kSeedRowsPerItem);
      while (wide_m) {
        const int src = __ffsll((long long)wide_m) - 1;
        wide_m &= wide_m - 1ull;
        const uint32_t w_lo = __shfl(lo[h], src, 64), w_n = __shfl(n_rows[h], src, 64), w_tag = __shfl(tag[h], src, 64);
        for (uint32_t x = lane; x < w_n; x += 64u) verify_one(p.pair_rows[w_lo + x], w_tag & 255u, w_tag >> 8);
      }
      push_rows(lo[h], n_rows[h] > kSeedRowsPerItem ? 0u : n_rows[h], tag[h]);
    }
  };

  // ---- the walk, software-pipelined (as seed_kernel's): while a trip's 256 reads are searched, the reads of the
  // next trip and the list entries of the one after it are in flight (loaded where they were needed, they cost
  // 0.37 of this launch's 0.98 ms: two dependent memory trips per 256 reads with nothing else to do) ----
  const uint32_t wc_stride = gridDim.x * 4u;
  auto load_entries = [&](uint32_t wc, uint32_t (&r)[4]) __attribute__((always_inline)) -> uint32_t {
    uint32_t m = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) r[u] = 0u;
    if (wc < n_chunks) {
      const uint32_t sgi = wc % in_nseg, depth = wc / in_nseg;
      const uint32_t count = p.idx_in ? p.in_count[sgi] : p.n_total;
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) {
        const uint32_t t = depth * kPairReads + lane + 64u * u;
        if (t < count) {
          r[u] = p.idx_in ? p.idx_in[((size_t)sgi * p.in_seg_cap + t) * p.in_stride] : t;
          m |= 1u << u;
        }
      }
    }
    return m;
  };
  auto load_reads = [&](const uint32_t (&r)[4], uint32_t m, uint32_t (&L)[4], uint64_t (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      L[u] = 0u;
      d[u] = 0ull;
      if ((m >> u) & 1u) {
        L[u] = p.lens[r[u]];
        d[u] = p.reads[r[u]];
      }
    }
  };
  // a list that carries its reads (in_stride = 4: the seed launches write such lists): entry = index, length, read --
  // streamed, 16 bytes per read, where the gathers through an index list touch 4.5 lines per line's worth of reads
  // (this pass walks 22 % of the batch): 0.36 -> 0.23 ms of this launch
  auto load_fat = [&](uint32_t wc, uint32_t (&r)[4], uint32_t (&L)[4], uint64_t (&d)[4]) __attribute__((always_inline)) -> uint32_t {
    uint32_t m = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      r[u] = 0u;
      L[u] = 0u;
      d[u] = 0ull;
    }
    if (wc < n_chunks) {
      const uint32_t sgi = wc % in_nseg, depth = wc / in_nseg;
      const uint32_t count = p.in_count[sgi];
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) {
        const uint32_t t = depth * kPairReads + lane + 64u * u;
        if (t < count) {
          const uint4 e = reinterpret_cast<const uint4*>(p.idx_in)[(size_t)sgi * p.in_seg_cap + t];
          r[u] = e.x;
          L[u] = e.y;
          d[u] = (uint64_t)e.z | ((uint64_t)e.w << 32);
          m |= 1u << u;
        }
      }
    }
    return m;
  };
  const bool fat = p.idx_in && p.in_stride == 4u;
  uint32_t r_b[4], L_b[4], r_c[4];
  uint64_t d_b[4];
  uint32_t m_b, m_c = 0;
  if (fat) {
    m_b = load_fat(blockIdx.x * 4u + wv, r_b, L_b, d_b);
  } else {
    m_b = load_entries(blockIdx.x * 4u + wv, r_b);
    load_reads(r_b, m_b, L_b, d_b);
    m_c = load_entries(blockIdx.x * 4u + wv + wc_stride, r_c);
  }
  for (uint32_t wc = blockIdx.x * 4u + wv; wc < n_chunks; wc += wc_stride) {
    // ---- the wave's 256 reads into its LDS region (RAP:543-554: which of them this pass's FASTA holds) ----
    uint32_t scan_m = 0;  // bit u: my read u has no anchors that fit (scan)
    uint32_t cur_r[4];
    const uint32_t cur_m = m_b;
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t slot = lane + 64u * u;
      const bool active = ((m_b >> u) & 1u) != 0u;
      const uint32_t L0 = L_b[u];
      const uint64_t rd = d_b[u];
      cur_r[u] = r_b[u];
      const bool eligible = active && (int32_t)L0 >= p.min_len && (int32_t)L0 <= p.max_len;
      const int32_t L = (int32_t)L0 - p.trim5 - p.trim3;
      if (eligible && p.count_processed) ++c_processed;
      const bool searching = eligible && L > p.max_mm_seed && !(LONG && L > 32);
      const int32_t R = min(L, p.seed_len);
      uint32_t A = 0u;
      if (searching) {
        if (R >= (int32_t)(4u * PA)) A = PA;
        else if (p.pair_jump_s && R >= (int32_t)(4u * (PA - 1u))) A = PA - 1u;
      }
      const bool scan = searching && A == 0u;
      scan_m |= scan ? 1u << u : 0u;
      uint64_t rdv = rd >> (2 * p.trim5);
      if (LONG && searching && L0 > 32u && p.trim5) rdv |= p.reads_hi[r_b[u]] << (64 - 2 * p.trim5);  // (bases 32.. that the 5' trim lets in)
      w.rd[slot] = rdv;
      w.best[slot] = ~0ull;
      w.len[slot] = (uint8_t)(searching ? L : 0);
      w.st[slot] = (uint8_t)(A | (scan ? 0x40u : 0u) | (active ? 0x80u : 0u));
    }
    // the pipeline moves on: the next trip's reads, the entries of the trip after it
    if (fat) {
      m_b = load_fat(wc + wc_stride, r_b, L_b, d_b);
    } else {
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) r_b[u] = r_c[u];
      m_b = m_c;
      load_reads(r_b, m_b, L_b, d_b);
      m_c = load_entries(wc + 2u * wc_stride, r_c);
    }
    wave_lds_sync();
    // ---- reads too short for anchors (fewer than 4 x 3 seed bases: under 15 nt): every text position, by the wave ----
    for (uint32_t u = 0; u < 4u; ++u) {
      uint64_t sm = __ballot((scan_m >> u) & 1u);
      while (sm) {
        const int src = __ffsll((long long)sm) - 1;
        sm &= sm - 1ull;
        const uint32_t slot = (uint32_t)src + 64u * u;
        for (uint32_t x = lane; x <= p.n; x += 64u) verify_one(p.sa[x], slot, 0u);
        c_cands += lane == 0u ? p.n + 1u : 0u;
      }
    }
    // ---- three rounds: pair (0,1); (2,3) for best > 0 mismatches; the other four for best > 1 ----
    for (uint32_t round = 0; round < 3u; ++round) {
      const uint32_t pr_lo = round < 2u ? round : 2u, pr_hi = round < 2u ? round + 1u : 6u;
      uint32_t n_it = 0;
      wave_lds_sync();
      for (uint32_t pr = pr_lo; pr < pr_hi; ++pr) {
        for (uint32_t u = 0; u <= 4u; ++u) {  // (the last trip of the last pair only works off what is left)
          if (u < 4u) {
            const uint32_t slot = lane + 64u * u;
            const bool open = w.len[slot] != 0u && (w.st[slot] & 7u) != 0u && (uint32_t)(w.best[slot] >> 56) >= round;
            const uint64_t mask = __ballot(open);
            if (open) w.items[n_it + mbcnt(mask)] = (uint16_t)(slot | (pr << 8));
            n_it += (uint32_t)__popcll(mask);
          }
          if (n_it >= 128u || (u == 4u && pr + 1u == pr_hi && n_it)) {
            const uint32_t m = min(n_it, 128u);
            n_it -= m;
            run_items(n_it, m);
          }
        }
      }
      if (rpend) {
        verify_rows(0u, rpend);
        rpend = 0u;
      }
    }
    wave_lds_sync();
    // ---- the claim, outputs, survivors (RAP:341-345) ----
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t slot = lane + 64u * u;
      const bool active = ((cur_m >> u) & 1u) != 0u;
      const unsigned long long key = w.best[slot];
      const bool aligned = active && key != ~0ull;
      const uint32_t r = cur_r[u];
      if (aligned) {
        ++c_aligned;
        uint32_t ref, pos;
        locate_entry(segs, (uint32_t)(key >> 24), (uint32_t)(key >> 8) & 0xFFFFu, (uint32_t)key & 0xFFu, ref, pos);
        const uint32_t mm = (uint32_t)(key >> 56);
        if (p.packed) {
          p.packed[r] = pack_assignment(p.pass_index, ref, pos, mm);
        } else {
          p.pass_id[r] = (int8_t)p.pass_index;
          p.ref_id[r] = (int32_t)ref;
          p.pos[r] = (int32_t)pos;
          p.mm[r] = (uint8_t)mm;
        }
      } else if (active && !p.idx_out && !p.out_init) {
        if (p.packed) {
          p.packed[r] = 0u;
        } else {
          p.pass_id[r] = (int8_t)-1;
          p.ref_id[r] = -1;
          p.pos[r] = -1;
          p.mm[r] = 0;
        }
      }
      if (p.idx_out) {
        const bool survive = active && !aligned;
        const uint64_t mask = __ballot(survive);
        if (mask) {
          uint32_t wbase = 0;
          if (lane == 0) wbase = atomicAdd(&ctl[0], (uint32_t)__popcll(mask));
          wbase = __shfl(wbase, 0, 64);
          if (survive) {  // (only with a pass behind this one -- spike-in: the entry is put together again)
            const uint64_t whole = p.reads[r];
            reinterpret_cast<uint4*>(p.idx_out)[(size_t)blockIdx.x * p.out_seg_cap + wbase + mbcnt(mask)] =
                make_uint4(r, (uint32_t)p.lens[r], (uint32_t)whole, (uint32_t)(whole >> 32));
          }
        }
      }
    }
    wave_lds_sync();  // (the next trip overwrites the region)
  }
  const uint64_t t_processed = wave_sum(c_processed), t_aligned = wave_sum(c_aligned);
  const uint64_t t_cands = wave_sum(c_cands), t_lookups = wave_sum(c_lookups);
  if (lane == 0) {
    if (t_processed) atomicAdd(&wg_cnt[0], (unsigned long long)t_processed);
    if (t_aligned) atomicAdd(&wg_cnt[1], (unsigned long long)t_aligned);
    if (t_cands) atomicAdd(&wg_cnt[3], (unsigned long long)t_cands);
    if (t_lookups) atomicAdd(&wg_cnt[4], (unsigned long long)t_lookups);
  }
  __syncthreads();
  if (tid < 5u && wg_cnt[tid]) atomicAdd((unsigned long long*)&p.counters[tid], wg_cnt[tid]);
  if (p.idx_out && tid == 0) p.out_count[blockIdx.x] = ctl[0];
}

uint32_t pair_wave_lds_total() { return kPairSharedWords * 4u + (kSeedThreads / 64u) * pair_wave_lds_bytes(); }

hipError_t launch_pair_wave(const MatchParams& p, uint32_t grid, hipStream_t stream) {
  if (p.reads_hi) hipLaunchKernelGGL(pair_wave_kernel<true>, dim3(grid), dim3(kSeedThreads), pair_wave_lds_total(), stream, p);
  else hipLaunchKernelGGL(pair_wave_kernel<false>, dim3(grid), dim3(kSeedThreads), pair_wave_lds_total(), stream, p);
  return hipGetLastError();
}

// mrg_pack_assignments: the four output arrays as one word per read (include/mirge_amd.h); four reads per lane
__global__ void __launch_bounds__(256) pack_assignments_kernel(const int8_t* __restrict__ pass_id, const int32_t* __restrict__ ref_id,
                                                               const int32_t* __restrict__ pos, const uint8_t* __restrict__ mm, uint64_t n,
                                                               uint32_t* __restrict__ packed, uint32_t vec) {
  auto pack = [](int32_t pi, int32_t ref, int32_t ps, uint32_t m) -> uint32_t {
    return pi < 0 ? 0u : pack_assignment(pi, (uint32_t)ref, (uint32_t)ps, m);
  };
  const uint64_t quads = n / 4;
  for (uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x; vec && q < quads; q += (uint64_t)gridDim.x * 256u) {
    const uint32_t p4 = reinterpret_cast<const uint32_t*>(pass_id)[q], m4 = reinterpret_cast<const uint32_t*>(mm)[q];
    const int4_t r4 = reinterpret_cast<const int4_t*>(ref_id)[q], o4 = reinterpret_cast<const int4_t*>(pos)[q];
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    u4_t out;
#pragma unroll
    for (int u = 0; u < 4; ++u) out[u] = pack((int32_t)(int8_t)((p4 >> (8 * u)) & 255u), r4[u], o4[u], (m4 >> (8 * u)) & 255u);
    reinterpret_cast<u4_t*>(packed)[q] = out;
  }
  for (uint64_t i = (vec ? quads * 4 : 0) + (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u)
    packed[i] = pack((int32_t)pass_id[i], ref_id[i], pos[i], mm[i]);
}

hipError_t launch_pack_assignments(const int8_t* pass_id, const int32_t* ref_id, const int32_t* pos, const uint8_t* mm, uint64_t n,
                                   uint32_t* packed, hipStream_t stream) {
  if (!n) return hipSuccess;
  const bool vec = ((uintptr_t)pass_id % 4 == 0) && ((uintptr_t)mm % 4 == 0) && ((uintptr_t)ref_id % 16 == 0) && ((uintptr_t)pos % 16 == 0) &&
                   ((uintptr_t)packed % 16 == 0);
  const uint64_t want = (n / 4 + 255) / 256;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want, 1), 256ull * 16);
  hipLaunchKernelGGL(pack_assignments_kernel, dim3(grid), dim3(256), 0, stream, pass_id, ref_id, pos, mm, n, packed, vec ? 1u : 0u);
  return hipGetLastError();
}

uint32_t seed_lds_bytes(const SeedParams& p) {
  if (p.impl == 1u) return wave_shared_words() * 4u + (kSeedThreads / 64u) * wave_lds_bytes(p.row_cap, p.reads_hi != nullptr);
  const uint32_t tile = kSeedThreads * p.reads_per_lane;
  return tile * 8u + tile * 8u + p.row_cap * 8u + kSeedWideCap * 16u + p.n_units * p.item_cap * 4u +
         kSeedMaxUnits * kUnitWords * 4u + kSeedCtlWords * 4u + kSeedCntSlots * 8u + ((tile + 7u) & ~7u) + (p.reads_hi ? tile * 8u : 0u);
}

uint32_t seed_wgs_per_cu(const SeedParams& p) {
  bool buckets = false;
  for (uint32_t u = 0; u < p.n_units; ++u) buckets |= p.unit[u].kind == 0u && p.unit[u].buckets != nullptr;
  uint32_t by_regs = (p.impl == 1u && (p.wave_regs & 1u)) ? (buckets ? 5u : 6u) : (buckets ? 6u : 8u);
  // (the instantiations that carry long reads get more registers: one or two workgroups per CU fewer)
  if (p.reads_hi) by_regs = p.impl == 1u ? (buckets ? 4u : 5u) : (buckets ? 5u : 6u);
  const uint32_t by_lds = (160u * 1024u) / seed_lds_bytes(p);
  return by_regs < by_lds ? by_regs : (by_lds ? by_lds : 1u);
}

hipError_t launch_seed(const SeedParams& p, uint32_t grid, hipStream_t stream) {
  if (p.reads_per_lane != 1u) return hipErrorInvalidValue;  // (a lane works on slot = tid of a tile of kSeedThreads reads)
  const uint32_t lds = seed_lds_bytes(p);
  bool buckets = false;
  for (uint32_t u = 0; u < p.n_units; ++u) buckets |= p.unit[u].kind == 0u && p.unit[u].buckets != nullptr;
  if (p.impl == 1u) {
    if (p.row_cap < kWaveRowsMin) return hipErrorInvalidValue;
    // (p.wave_regs: the instantiation with more registers and fewer resident workgroups -- an A/B knob)
    const bool more_regs = (p.wave_regs & 1u) != 0u;
    const void* wk = buckets ? (more_regs ? reinterpret_cast<const void*>(wave_seed_kernel<true, 5, false>) : reinterpret_cast<const void*>(wave_seed_kernel<true, 6, false>))
                             : (more_regs ? reinterpret_cast<const void*>(wave_seed_kernel<false, 6, false>) : reinterpret_cast<const void*>(wave_seed_kernel<false, 8, false>));
    if (lds > 48u * 1024u) {
      hipError_t e = hipFuncSetAttribute(wk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    if (p.reads_hi) {  // a batch with reads of 33..63 nt: the instantiations that carry the second word
      const void* lk = buckets ? (more_regs ? reinterpret_cast<const void*>(wave_seed_kernel<true, 4, true>) : reinterpret_cast<const void*>(wave_seed_kernel<true, 4, true>))
                               : (more_regs ? reinterpret_cast<const void*>(wave_seed_kernel<false, 5, true>) : reinterpret_cast<const void*>(wave_seed_kernel<false, 5, true>));
      if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(lk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
      }
      if (buckets && more_regs) hipLaunchKernelGGL((wave_seed_kernel<true, 4, true>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
      else if (buckets) hipLaunchKernelGGL((wave_seed_kernel<true, 4, true>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
      else if (more_regs) hipLaunchKernelGGL((wave_seed_kernel<false, 5, true>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
      else hipLaunchKernelGGL((wave_seed_kernel<false, 5, true>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
      return hipGetLastError();
    }
    if (buckets && more_regs) hipLaunchKernelGGL((wave_seed_kernel<true, 5, false>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    else if (buckets) hipLaunchKernelGGL((wave_seed_kernel<true, 6, false>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    else if (more_regs) hipLaunchKernelGGL((wave_seed_kernel<false, 6, false>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    else hipLaunchKernelGGL((wave_seed_kernel<false, 8, false>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    return hipGetLastError();
  }
  const bool fat = p.idx_in && p.in_stride == 4u;
  // (the same four instantiations with and without the second word of long reads)
  auto go = [&](auto long_tag) -> hipError_t {
    constexpr bool LONG = decltype(long_tag)::value;
    const void* kern = buckets ? (fat ? reinterpret_cast<const void*>(seed_kernel<true, (LONG ? 5 : 6), true, LONG>) : reinterpret_cast<const void*>(seed_kernel<true, (LONG ? 5 : 6), false, LONG>))
                               : (fat ? reinterpret_cast<const void*>(seed_kernel<false, (LONG ? 6 : 8), true, LONG>) : reinterpret_cast<const void*>(seed_kernel<false, (LONG ? 6 : 8), false, LONG>));
    if (lds > 48u * 1024u) {
      hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    if (buckets && fat) hipLaunchKernelGGL((seed_kernel<true, (LONG ? 5 : 6), true, LONG>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    else if (buckets) hipLaunchKernelGGL((seed_kernel<true, (LONG ? 5 : 6), false, LONG>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    else if (fat) hipLaunchKernelGGL((seed_kernel<false, (LONG ? 6 : 8), true, LONG>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    else hipLaunchKernelGGL((seed_kernel<false, (LONG ? 6 : 8), false, LONG>), dim3(grid), dim3(kSeedThreads), lds, stream, p);
    return hipGetLastError();
  };
  return p.reads_hi ? go(std::true_type{}) : go(std::false_type{});
}

// the streaming instantiation needs the identity list and arrays it can address 16 bytes at a time; it writes EVERY output
// of the batch (the "unannotated" values for what it does not claim)
bool exact_dict_streams(const ExactParams& p) {
  const bool out_ok = p.packed ? ((uintptr_t)p.packed % 16 == 0)
                               : (((uintptr_t)p.pass_id % 4 == 0) && ((uintptr_t)p.mm % 4 == 0) && ((uintptr_t)p.ref_id % 16 == 0) &&
                                  ((uintptr_t)p.pos % 16 == 0));
  return !p.idx_in && ((uintptr_t)p.reads % 16 == 0) && ((uintptr_t)p.lens % 4 == 0) && out_ok;
}

// ... and takes it by stretches when more than 2 % of the (workgroup, trip) slots of a chunked run would stay empty
bool exact_dict_stretches(const ExactParams& p, uint32_t grid) {
  if (!exact_dict_streams(p) || !grid) return false;
  const uint64_t per_round = (uint64_t)grid * kExactChunk;
  const uint64_t rounds = (p.n_total + per_round - 1) / per_round;
  return rounds * per_round * 50ull > (uint64_t)p.n_total * 51ull;
}

hipError_t launch_exact_dict(const ExactParams& p, uint32_t grid, hipStream_t stream) {
  const bool first = exact_dict_streams(p);
  const bool kb = p.kbits != nullptr && p.key_bases >= kKmerBitsK;
  if (first && exact_dict_stretches(p, grid)) {
    if (kb) hipLaunchKernelGGL((exact_dict_kernel<true, true, true>), dim3(grid), dim3(kBlock), 0, stream, p);
    else hipLaunchKernelGGL((exact_dict_kernel<true, false, true>), dim3(grid), dim3(kBlock), 0, stream, p);
    return hipGetLastError();
  }
  if (first && kb) hipLaunchKernelGGL((exact_dict_kernel<true, true>), dim3(grid), dim3(kBlock), 0, stream, p);
  else if (first) hipLaunchKernelGGL((exact_dict_kernel<true, false>), dim3(grid), dim3(kBlock), 0, stream, p);
  else if (kb) hipLaunchKernelGGL((exact_dict_kernel<false, true>), dim3(grid), dim3(kBlock), 0, stream, p);
  else hipLaunchKernelGGL((exact_dict_kernel<false, false>), dim3(grid), dim3(kBlock), 0, stream, p);
  return hipGetLastError();
}

}  // namespace mrg
