// gfx950 (CDNA4, wave64) kernels of the annotation cascade.
//
// What they replace in the reference: the nine external bowtie runs of
// runAnnotationPipeline (RAP:636-705), the Python survivor scan between them
// (writeSeqToAnnot RAP:543-554, updateAnnotDic RAP:341-352) and the Python
// histogram loop of summarize (SUM:34-66).
//
// match_kernel: one read per lane.  A pass's mismatch policy is turned into
// K = max_mm_seed+1 pigeonhole pieces of the seed region; each piece is an
// exact FM backward search (the rank/occ blocks and, when they fit, the packed
// text are staged in LDS), every occurrence is located through the full suffix
// array and verified against the 2-bit text with XOR+popcount, and the best
// (mismatches, text position) wins.  Unclaimed reads are appended to the next
// pass's survivor list with one wave-aggregated atomic per wave.
//
// This is integer/index work: no MFMA.  The budget that matters is LDS
// accesses + VALU per LF step for staged libraries and L2/MALL/HBM gathers for
// the large ones.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mrg {

namespace {

constexpr uint64_t kOdd = 0x5555555555555555ull;
// suffix-array rows loaded together in the candidate loop: 4 when the text windows come from LDS,
// 8 when they are HBM/L2 loads themselves (more of them in flight; measured: pass 6 -0.1 ms, the
// LDS-text passes +5 % with 8)
template <bool LDST>
struct RowsPerTrip {
  static constexpr uint32_t value = LDST ? 4u : 8u;
};

__device__ __forceinline__ uint64_t low_bits(uint32_t nbits) {
  // nbits in [0,64]
  return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Access to one library: occ blocks / text either in LDS or in global memory; the
// superblock table is always in LDS.
template <bool LDSI, bool LDST>
struct Lib {
  const uint32_t* __restrict__ gblocks;
  const uint32_t* __restrict__ gtext;
  const uint32_t* sblocks;  // LDS
  const uint32_t* stext;    // LDS
  const uint32_t* ssuper;   // LDS
  uint32_t primary;

  __device__ __forceinline__ uint4 block(uint32_t b) const {
    if (LDSI) return *reinterpret_cast<const uint4*>(sblocks + b * 4);
    return *reinterpret_cast<const uint4*>(gblocks + (size_t)b * 4);
  }

  // first BWT row of the c-suffixes + rank of c before row i, from the block `v`
  // holding row i (one 16-byte load per query; shared by lo and hi when they sit
  // in the same block)
  __device__ __forceinline__ uint32_t lf(uint32_t c, uint32_t i, const uint4& v) const {
    const uint32_t r = i & 31;
    const uint32_t pair = (c & 2) ? v.y : v.x;
    const uint32_t cnt = (c & 1) ? (pair >> 16) : (pair & 0xffffu);
    uint32_t e = ((c & 1) ? v.z : ~v.z) & ((c & 2) ? v.w : ~v.w);
    e &= (1u << r) - 1u;
    uint32_t o = ssuper[(i >> 16) * 4 + c] + cnt + (uint32_t)__popc(e);
    // the sentinel row is stored as symbol 0 inside its own block only
    o -= (uint32_t)((c == 0) & (i > primary) & ((i >> 5) == (primary >> 5)));
    return o;
  }

  __device__ __forceinline__ uint64_t window(uint32_t p) const {
    const uint32_t i = p >> 4, sh = (p & 15) * 2;
    uint32_t w0, w1, w2;
    if (LDST) {
      w0 = stext[i];
      w1 = stext[i + 1];
      w2 = stext[i + 2];
    } else {
      w0 = gtext[i];
      w1 = gtext[i + 1];
      w2 = gtext[i + 2];
    }
    const uint64_t lo64 = (uint64_t)w0 | ((uint64_t)w1 << 32);
    return (lo64 >> sh) | ((((uint64_t)w2) << 1) << (63 - sh));
  }
};

// k bases packed first-base-lowest -> their lexicographic number (first base most significant):
// reverse the order of the 2-bit groups.
__device__ __forceinline__ uint32_t lex_code(uint64_t code, uint32_t k) {
  const uint64_t r = __brevll(code) >> (64u - 2u * k);  // groups reversed, bits inside a group swapped
  return (uint32_t)(((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1));
}

// The largest jump table a seed piece of `plen` bases is long enough for: its k (0 = none) and
// its word offset.
__device__ __forceinline__ uint32_t pick_table(const JumpTables& t, int32_t plen, uint32_t& word_off) {
  uint32_t k = plen >= (int32_t)t.k[0] ? t.k[0] : 0u;
  word_off = t.off[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    const bool take = plen >= (int32_t)t.k[i];
    k = take ? t.k[i] : k;
    word_off = take ? t.off[i] : word_off;
  }
  return k;
}

// x / K for the piece counts a pass can have (1..3), x < 65536: no integer-divide sequence
__device__ __forceinline__ int32_t div_pieces(int32_t x, int32_t K) {
  return K == 1 ? x : (K == 2 ? (x >> 1) : (int32_t)(((uint32_t)x * 43691u) >> 17));
}

template <int W>
__device__ __forceinline__ uint64_t pick_word(const uint64_t (&rd)[W], uint32_t w) {
  uint64_t v = rd[0];
#pragma unroll
  for (int k = 1; k < W; ++k) v = (w == (uint32_t)k) ? rd[k] : v;
  return v;
}

// Drop `t` (<32) bases from the 5' end: base i becomes base i - t.
template <int W>
__device__ __forceinline__ void shift_out_5p(uint64_t (&rd)[W], uint32_t t) {
  if (t == 0) return;
  const uint32_t sh = 2 * t;
#pragma unroll
  for (int k = 0; k + 1 < W; ++k) rd[k] = (rd[k] >> sh) | (rd[k + 1] << (64 - sh));
  rd[W - 1] >>= sh;
}

// 2c bits of the read starting at base `at` (c <= 8).
template <int W>
__device__ __forceinline__ uint32_t read_bits(const uint64_t (&rd)[W], uint32_t at, uint32_t c) {
  uint64_t v = pick_word<W>(rd, at >> 5) >> ((at & 31u) * 2u);
  if (W > 1 && (at & 31u) + c > 32u) v |= pick_word<W>(rd, (at >> 5) + 1u) << (64u - (at & 31u) * 2u);
  return (uint32_t)v & ((1u << (2u * c)) - 1u);
}

// Row context (FmIndex::ctx) of the candidates of one seed interval: all of them have `need_before`
// read bases left of the row's position and `need_after` from it on, so what the read expects in
// the context word -- the <= 8 bases before the position (bits 0-15, nearest in the top two) and
// the bases 8..15 after it (bits 16-31) -- is computed once: (want, mask).  The mismatches a row's
// word then shows are a lower bound of the alignment's (a read N counts as a match, bases across
// a segment end only occur in alignments that are invalid anyway): "more than the pass allows"
// is final.
template <int W>
__device__ __forceinline__ uint2 context_probe(const uint64_t (&rd)[W], uint32_t need_before, uint32_t need_after) {
  uint32_t want = 0, mask = 0;
  const uint32_t c = min(need_before, 8u);
  if (c) {
    want = read_bits<W>(rd, need_before - c, c) << (16u - 2u * c);
    mask = ((1u << (2u * c)) - 1u) << (16u - 2u * c);
  }
  if (need_after > 8u) {
    const uint32_t c2 = min(need_after, 16u) - 8u;
    want |= read_bits<W>(rd, need_before + 8u, c2) << 16;
    mask |= ((1u << (2u * c2)) - 1u) << 16;
  }
  return make_uint2(want, mask);
}

__device__ __forceinline__ uint32_t context_mismatches(uint32_t ctx, uint2 probe) {
  const uint32_t x = (ctx ^ probe.x) & probe.y;
  return (uint32_t)__popc((x | (x >> 1)) & 0x55555555u);
}

// count_kernel pre-filters intervals at least this wide with the row context (narrower ones are
// cheaper to verify directly: the context word would be one more load per row).  match_kernel
// uses the context in its wave-cooperative path, which large libraries enter from
// MatchParams::wide_rows = 32 rows on.
constexpr uint32_t kCtxMinRows = 8u;

// One suffix-array row as a candidate alignment of a read whose seed search stopped with
// `need_before` read bases left of the row's text position and `need_after` from it on.
// Updates (best, best_seg, best_before) when the alignment is valid and better.
template <int W, class LibT>
__device__ __forceinline__ void verify_row(const LibT& lib, const MatchParams& p, const uint64_t row,
                                           const uint64_t (&rd)[W], const uint64_t (&nm)[W], int32_t L,
                                           uint32_t need_before, uint32_t need_after, uint64_t& best,
                                           uint32_t& best_seg, uint32_t& best_before) {
  // the alignment [pos - j, pos - j + L) must stay inside the N-free segment
  const uint32_t before = (uint32_t)(row >> 32) & 255u, after = (uint32_t)(row >> 40) & 255u;
  if ((need_before > before) | (need_after > after)) return;
  const uint32_t s = (uint32_t)row - need_before;
  uint32_t mm_total = 0, mm_seed = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const int32_t nb = min(32, L - 32 * w);
    if (nb > 0) {
      const uint64_t x = lib.window(s + 32u * w) ^ rd[w];
      uint64_t m = (x | (x >> 1)) & kOdd;
      if (p.nmask) m |= nm[w];
      m &= low_bits(2 * nb);
      mm_total += (uint32_t)__popcll(m);
      // seed mismatches only differ from the total for reads longer than the seed
      if (L > p.seed_len) {
        const int32_t ns = min(nb, max(0, p.seed_len - 32 * w));
        mm_seed += (uint32_t)__popcll(m & low_bits(2 * ns));
      }
    }
  }
  if (L <= p.seed_len) mm_seed = mm_total;
  if (((int32_t)mm_seed > p.max_mm_seed) | ((int32_t)mm_total > p.max_mm_total)) return;
  const uint64_t key = ((uint64_t)mm_total << 32) | s;
  if (key < best) {
    best = key;
    best_seg = (uint32_t)(row >> 48);
    best_before = before < 255u ? before - need_before : 255u;
  }
}

}  // namespace

// STRATA: the 2-mismatch policy (three seed pieces) with its stratum-first search; a separate
// instantiation so that the other passes keep the plain piece loop.
// CTX: the library has a row-context array (>= 2^20 bases; never together with LDST).
template <int W, bool LDSI, bool LDST, bool STRATA, bool CTX, bool KBITS>
__global__ void __launch_bounds__(MatchBlock<LDSI>::kThreads, (W == 1 ? 8 : 4))
match_kernel(const MatchParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr uint32_t BLOCK = MatchBlock<LDSI>::kThreads;

  // ---- LDS carve: [superblocks][occ blocks][text][9-mer bitmap][control] ----
  const uint32_t sup_words = p.nsup * 4;
  const uint32_t blk_words = LDSI ? p.nblk * 4 : 0u;
  const uint32_t txt_words = LDST ? p.text_words : 0u;
  uint32_t* ssuper = smem;
  uint32_t* sblocks = ssuper + sup_words;
  uint32_t* stext = sblocks + blk_words;
  uint32_t* skbits = stext + txt_words;
  uint32_t* ctl = skbits + (KBITS ? kKmerBitsWords : 0u);  // [0] survivors appended here, [1] longest input segment
  uint32_t* wave_slots = ctl + 4;           // 16 B per wave: minimum of a wave-wide verification
  {
    // 16 B per lane per trip
    const uint4* src = reinterpret_cast<const uint4*>(p.super);
    uint4* dst = reinterpret_cast<uint4*>(ssuper);
    for (uint32_t i = threadIdx.x; i < sup_words / 4; i += BLOCK) dst[i] = src[i];
  }
  if (LDSI) {
    const uint4* src = reinterpret_cast<const uint4*>(p.blocks);
    uint4* dst = reinterpret_cast<uint4*>(sblocks);
    for (uint32_t i = threadIdx.x; i < blk_words / 4; i += BLOCK) dst[i] = src[i];
  }
  if (LDST) {
    const uint4* src = reinterpret_cast<const uint4*>(p.text);
    uint4* dst = reinterpret_cast<uint4*>(stext);
    for (uint32_t i = threadIdx.x; i < txt_words / 4; i += BLOCK) dst[i] = src[i];
  }
  if (KBITS) {
    const uint4* src = reinterpret_cast<const uint4*>(p.kbits);
    uint4* dst = reinterpret_cast<uint4*>(skbits);
    for (uint32_t i = threadIdx.x; i < kKmerBitsWords / 4; i += BLOCK) dst[i] = src[i];
  }
  // The input list is the producer pass's per-workgroup segments.  Consumers walk
  // them round-robin (chunk c -> segment c % nseg, depth c / nseg): the workgroups
  // running at any moment then touch reads that are neighbours in HBM, as pass 0 does;
  // walking one segment after the other makes them 4 MB-stride accesses that pile
  // onto the same HBM channels.
  if (threadIdx.x == 0) {
    ctl[0] = 0u;
    ctl[1] = 0u;
  }
  __syncthreads();
  if (p.idx_in) {
    uint32_t mx = 0;
    for (uint32_t sgi = threadIdx.x; sgi < p.in_nseg; sgi += BLOCK) {
      const uint32_t cnt = p.in_count[sgi];
      mx = max(mx, cnt);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, off, 64));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&ctl[1], mx);
  }
  __syncthreads();

  Lib<LDSI, LDST> lib;
  lib.gblocks = p.blocks;
  lib.gtext = p.text;
  lib.sblocks = sblocks;
  lib.stext = stext;
  lib.ssuper = ssuper;
  lib.primary = p.primary;

  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-lane tallies (a lane handles a few hundred reads per launch: 32 bits suffice)
  uint32_t c_processed = 0, c_aligned = 0, c_steps = 0, c_cands = 0, c_lookups = 0;

  const uint32_t in_nseg = p.idx_in ? p.in_nseg : 1u;
  const uint32_t depth_chunks = p.idx_in ? (ctl[1] + BLOCK - 1) / BLOCK : (p.n_total + BLOCK - 1) / BLOCK;
  const uint32_t n_chunks = in_nseg * depth_chunks;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t sgi = chunk % in_nseg, depth = chunk / in_nseg;
    const uint32_t t = depth * BLOCK + threadIdx.x;
    // (the segment length is a workgroup-uniform load of a 2 KB array: not worth 2 KB of LDS)
    const bool active = t < (p.idx_in ? p.in_count[sgi] : p.n_total);
    uint32_t r = 0;
    uint64_t rd[W], nm[W];
    uint32_t L0 = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) rd[k] = nm[k] = 0ull;
    if (active) {
      r = p.idx_in ? p.idx_in[(size_t)sgi * p.in_seg_cap + t] : t;
      L0 = p.lens[r];
    }
    // ---- which reads this pass's FASTA would contain (RAP:543-554, 664-686) ----
    bool eligible = active && (int32_t)L0 >= p.min_len && (int32_t)L0 <= p.max_len;
    if (eligible) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        rd[k] = p.reads[(size_t)k * p.n_total + r];
        nm[k] = p.nmask ? p.nmask[(size_t)k * p.n_total + r] : 0ull;
      }
    }
    int32_t L = (int32_t)L0;
    if (p.poly_t) {
      // number of trailing T (code 3); an N base is never a T
      int32_t hb = -1;  // highest base that is not T
#pragma unroll
      for (int k = W - 1; k >= 0; --k) {
        const int32_t nb = min(32, max(0, L - 32 * k));
        uint64_t x = (~rd[k] | nm[k] | (nm[k] << 1)) & low_bits(2 * nb);
        if (hb < 0 && x != 0ull) hb = 32 * k + ((63 - __clzll((long long)x)) >> 1);
      }
      const int32_t tail = L - 1 - hb;
      eligible = eligible && tail >= 3 && (L - tail) >= 11;
      L = L - tail;
    }
    L -= p.trim5 + p.trim3;
    shift_out_5p<W>(rd, (uint32_t)p.trim5);
    if (p.nmask) shift_out_5p<W>(nm, (uint32_t)p.trim5);
    if (eligible) ++c_processed;

    uint64_t best = ~0ull;  // (mm << 32) | text position
    uint32_t best_seg = 0xFFFFu, best_before = 255u;
    if (eligible && L > p.max_mm_seed) {
      const int32_t R = min(L, p.seed_len);
      const int32_t Kfull = p.max_mm_seed + 1;
      // Stratum first (2-mismatch policies only): a search with K pieces finds EVERY alignment with
      // fewer than K seed mismatches, so when the 1-piece (exact) or the 2-piece search already
      // yields a best hit below its own bound, that hit is final and the 3-piece search -- ~15
      // candidates per 6-base piece -- is not run.  Most isomiRs are exact or 1-mismatch matches
      // of the flanked library entry.
      for (int32_t K = (STRATA ? 1 : Kfull); K <= Kfull; ++K) {
      for (int32_t k = 0; k < K; ++k) {
        const int32_t a = div_pieces(R * k, K), b = div_pieces(R * (k + 1), K);
        if (p.nmask) {
          // a piece holding an N can never be the exact one
          bool has_n = false;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const int32_t lo_b = max(a - 32 * w, 0), hi_b = min(b - 32 * w, 32);
            if (hi_b > lo_b) has_n |= (nm[w] & low_bits(2 * hi_b) & ~low_bits(2 * lo_b)) != 0ull;
          }
          if (has_n) continue;
        }
        if (KBITS && b - a >= (int32_t)kKmerBitsK) {
          // the piece's last 9 bases do not occur in the library: the piece cannot match, and
          // an LDS bit test has answered it instead of a jump-table load (a random L2 request)
          const uint32_t at = (uint32_t)b - kKmerBitsK;
          uint64_t v = pick_word<W>(rd, at >> 5) >> ((at & 31u) * 2u);
          if (W > 1 && (at & 31u) + kKmerBitsK > 32u) v |= pick_word<W>(rd, (at >> 5) + 1u) << (64u - (at & 31u) * 2u);
          const uint32_t c9 = (uint32_t)v & ((1u << (2u * kKmerBitsK)) - 1u);
          if (((skbits[c9 >> 5] >> (c9 & 31u)) & 1u) == 0u) continue;
          if (b - a > (int32_t)kKmerBitsK) {
            // ... and neither do its first 9 (a second, nearly independent test for longer pieces)
            const uint32_t at0 = (uint32_t)a;
            uint64_t v0 = pick_word<W>(rd, at0 >> 5) >> ((at0 & 31u) * 2u);
            if (W > 1 && (at0 & 31u) + kKmerBitsK > 32u)
              v0 |= pick_word<W>(rd, (at0 >> 5) + 1u) << (64u - (at0 & 31u) * 2u);
            const uint32_t c0 = (uint32_t)v0 & ((1u << (2u * kKmerBitsK)) - 1u);
            if (((skbits[c0 >> 5] >> (c0 & 31u)) & 1u) == 0u) continue;
          }
        }
        // ---- exact backward search of read[a,b) ----
        uint32_t lo = 0, hi = p.n + 1;
        int32_t j = b;
        uint32_t tab_off = 0;
        const uint32_t tk = p.tabs.k[0] ? pick_table(p.tabs, b - a, tab_off) : 0u;
        if (tk) {
          // the piece's last k bases in one load: BWT interval of that k-mer (largest
          // table the piece is long enough for)
          j = b - (int32_t)tk;
          uint64_t code = pick_word<W>(rd, (uint32_t)j >> 5) >> ((j & 31) * 2);
          if (W > 1 && (j & 31) + (int32_t)tk > 32)
            code |= pick_word<W>(rd, ((uint32_t)j >> 5) + 1) << (64 - (j & 31) * 2);
          code &= (1ull << (2 * tk)) - 1ull;
          const uint32_t* tab = p.ftab + tab_off + lex_code(code, tk);
          lo = tab[0];
          hi = tab[1];
          ++c_lookups;
        }
        while (j > a && hi > lo && (hi - lo) > p.wstop) {
          --j;
          const uint32_t c = (uint32_t)(pick_word<W>(rd, (uint32_t)j >> 5) >> ((j & 31) * 2)) & 3u;
          const uint4 vl = lib.block(lo >> 5);
          uint4 vh = vl;
          if ((hi >> 5) != (lo >> 5)) vh = lib.block(hi >> 5);
          lo = lib.lf(c, lo, vl);
          hi = lib.lf(c, hi, vh);
          ++c_steps;
        }
        // ---- locate + verify every occurrence ----
        const uint32_t need_before = (uint32_t)j, need_after = (uint32_t)(L - j);
        const uint32_t width = hi > lo ? hi - lo : 0u;
        c_cands += width;
        const bool wide = width > p.wide_rows;
        if (!wide) {
          // four suffix-array rows per trip: they mostly share a cache line and their loads overlap
          constexpr uint32_t kRowsPerTrip = RowsPerTrip<LDST>::value;
          for (uint32_t i = lo; i < hi; i += kRowsPerTrip) {
            uint64_t rows[kRowsPerTrip];
#pragma unroll
            for (uint32_t u = 0; u < kRowsPerTrip; ++u) rows[u] = (i + u < hi) ? p.sa[i + u] : 0ull;
#pragma unroll
            for (uint32_t u = 0; u < kRowsPerTrip; ++u)
              if (i + u < hi)
                verify_row<W>(lib, p, rows[u], rd, nm, L, need_before, need_after, best, best_seg, best_before);
          }
        }
        // A low-complexity seed (poly-A, repeats) can match 10^3..10^6 rows; one lane walking
        // them would stall its wave for that long.  Such intervals are verified by the whole
        // wave instead: the owner's read is broadcast, every lane takes rows lo+lane, +64, ...
        // (coalesced suffix-array reads) and the minimum is combined through LDS.
        uint64_t wide_mask = __ballot(wide);
        while (wide_mask) {
          const int src = __ffsll((long long)wide_mask) - 1;
          wide_mask &= wide_mask - 1;
          uint64_t o_rd[W], o_nm[W];
#pragma unroll
          for (int w = 0; w < W; ++w) {
            o_rd[w] = __shfl(rd[w], src, 64);
            o_nm[w] = __shfl(nm[w], src, 64);
          }
          const uint32_t o_lo = __shfl(lo, src, 64), o_hi = __shfl(hi, src, 64);
          const int32_t o_L = __shfl(L, src, 64);
          const uint32_t o_nb = __shfl(need_before, src, 64), o_na = __shfl(need_after, src, 64);
          uint64_t w_best = ~0ull;
          uint32_t w_seg = 0xFFFFu, w_before = 255u;
          // only the lanes still in this loop can help (others left it with a result or
          // were never offered a read): deal the rows over exactly those
          const uint64_t helpers = __ballot(true);
          const uint32_t n_help = (uint32_t)__popcll(helpers);
          const uint32_t my_rank = (uint32_t)__popcll(helpers & ((1ull << lane) - 1ull));
          const uint2 probe = CTX ? context_probe<W>(o_rd, o_nb, o_na) : make_uint2(0u, 0u);
          for (uint32_t i = o_lo + my_rank; i < o_hi; i += n_help) {
            if (CTX && (int32_t)context_mismatches(p.ctx[i], probe) > p.max_mm_total) continue;
            verify_row<W>(lib, p, p.sa[i], o_rd, o_nm, o_L, o_nb, o_na, w_best, w_seg, w_before);
          }
          unsigned long long* slot = reinterpret_cast<unsigned long long*>(wave_slots) + 2 * wave;
          if ((int)lane == src) {
            slot[0] = ~0ull;
            slot[1] = 0ull;
          }
          if (w_best != ~0ull) atomicMin(&slot[0], (unsigned long long)w_best);
          if (w_best != ~0ull && slot[0] == w_best) slot[1] = ((unsigned long long)w_seg << 32) | w_before;
          if ((int)lane == src) {
            const uint64_t got = slot[0];
            if (got < best) {
              best = got;
              best_seg = (uint32_t)(slot[1] >> 32);
              best_before = (uint32_t)slot[1];
            }
          }
        }
        if ((best >> 32) == 0ull) break;  // an exact hit is always seen by piece 0
      }
      if ((uint32_t)(best >> 32) < (uint32_t)K) break;  // complete below K mismatches (unaligned = 2^32 - 1)
      }
    }

    const bool aligned = best != ~0ull;
    if (aligned) {
      ++c_aligned;
      const uint32_t s = (uint32_t)best;
      uint32_t sg = best_seg;
      if (sg == 0xFFFFu) {  // more than 65535 segments: walk the chunk map
        sg = p.chunk_seg[s >> 5];
        while (p.seg_start[sg + 1] <= s) ++sg;
      }
      uint32_t ref = sg, pos;
      if (p.simple_segs && best_before < 255u) {
        pos = best_before;  // offset of the alignment start inside its entry, from the SA row
      } else {
        uint32_t off = 0;
        if (!p.simple_segs) {
          ref = p.seg_ref[sg];
          off = p.seg_off[sg];
        }
        pos = s - p.seg_start[sg] + off;
      }
      p.pass_id[r] = (int8_t)p.pass_index;
      p.ref_id[r] = (int32_t)ref;
      p.pos[r] = (int32_t)pos;
      p.mm[r] = (uint8_t)(best >> 32);
    } else if (active && !p.idx_out) {
      // last pass: whatever is still unclaimed stays unannotated (no memset needed)
      p.pass_id[r] = (int8_t)-1;
      p.ref_id[r] = -1;
      p.pos[r] = -1;
      p.mm[r] = 0;
    }

    // ---- survivors of this pass feed the next one ----
    // Each workgroup owns a private segment of the output list: a wave reserves its
    // slots with ONE LDS atomic and writes them straight to HBM.  No global atomics
    // (a single address sustains only ~11 ns per atomic) and no barrier in the loop.
    if (p.idx_out) {
      const bool survive = active && !aligned;
      const uint64_t mask = __ballot(survive);
      if (mask) {
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&ctl[0], (uint32_t)__popcll(mask));
        wbase = __shfl(wbase, 0, 64);
        if (survive)
          p.idx_out[(size_t)blockIdx.x * p.out_seg_cap + wbase +
                    (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = r;
      }
    }
  }
  if (p.idx_out) {
    __syncthreads();
    if (threadIdx.x == 0) p.out_count[blockIdx.x] = ctl[0];
  }

  const uint64_t t_processed = wave_sum(c_processed), t_aligned = wave_sum(c_aligned);
  const uint64_t t_steps = wave_sum(c_steps), t_cands = wave_sum(c_cands);
  const uint64_t t_lookups = wave_sum(c_lookups);
  if (lane == 0) {
    if (t_processed) atomicAdd((unsigned long long*)&p.counters[0], (unsigned long long)t_processed);
    if (t_aligned) atomicAdd((unsigned long long*)&p.counters[1], (unsigned long long)t_aligned);
    if (t_steps) atomicAdd((unsigned long long*)&p.counters[2], (unsigned long long)t_steps);
    if (t_cands) atomicAdd((unsigned long long*)&p.counters[3], (unsigned long long)t_cands);
    if (t_lookups) atomicAdd((unsigned long long*)&p.counters[4], (unsigned long long)t_lookups);
  }
}

// ---------------------------------------------------------------------------
// count_kernel: best stratum of every read against one library -- fewest mismatches of a
// valid alignment and how many alignments reach it.  Replaces the two genome bowtie runs
// of the -ai path (writeDataToCSV.py:1263 `-n 1 -a -3 2`, :1488 `-n 0 -a -3 2`), whose
// only use is "is the best hit unique" (:1277-1287) / "does it align at all" (:1491-1496).
// Same seed-and-verify as match_kernel, library served from HBM/L2 (a chromosome), no
// survivor lists.  An alignment can sit in the candidate rows of several pieces; it is
// counted at the first piece whose searched bases it matches exactly.
// ---------------------------------------------------------------------------
// LIST = second sweep of mrg_list_best: the best stratum of each read is known, every alignment
// in it is written at offsets[r] + k (what `-a --best --strata` prints, RAP:577-599, consumed by
// parseAlignment3 RAP:41-52 for the tRF tables).
template <int W, bool LIST>
__global__ void __launch_bounds__(kCountThreads) count_kernel(const CountParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.super);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (uint32_t i = threadIdx.x; i < p.nsup; i += kCountThreads) dst[i] = src[i];
  }
  __syncthreads();
  Lib<false, false> lib;
  lib.gblocks = p.blocks;
  lib.gtext = p.text;
  lib.sblocks = nullptr;
  lib.stext = nullptr;
  lib.ssuper = smem;
  lib.primary = p.primary;

  for (uint64_t r = (uint64_t)blockIdx.x * kCountThreads + threadIdx.x; r < p.n_reads;
       r += (uint64_t)gridDim.x * kCountThreads) {
    uint64_t rd[W], nm[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      rd[k] = p.reads[(size_t)k * p.n_reads + r];
      nm[k] = p.nmask ? p.nmask[(size_t)k * p.n_reads + r] : 0ull;
    }
    const int32_t L = (int32_t)p.lens[r];
    uint32_t best_mm = 255u, count = 0u;
    const uint32_t want_mm = LIST ? (uint32_t)p.best_mm[r] : 0u;
    const uint64_t out_base = LIST ? p.offsets[r] : 0ull;
    if (L > p.max_mm_seed && (!LIST || want_mm != 255u)) {
      const int32_t R = min(L, p.seed_len);
      const int32_t K = p.max_mm_seed + 1;
      int32_t stop_[3] = {0, 0, 0}, end_[3] = {0, 0, 0};
      bool listed[3] = {false, false, false};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k >= K) continue;
        const int32_t a = div_pieces(R * k, K), b = div_pieces(R * (k + 1), K);
        bool has_n = false;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const int32_t lo_b = max(a - 32 * w, 0), hi_b = min(b - 32 * w, 32);
          if (hi_b > lo_b) has_n |= (nm[w] & low_bits(2 * hi_b) & ~low_bits(2 * lo_b)) != 0ull;
        }
        if (has_n) continue;
        uint32_t lo = 0, hi = p.n + 1;
        int32_t j = b;
        uint32_t tab_off = 0;
        const uint32_t kk = p.tabs.k[0] ? pick_table(p.tabs, b - a, tab_off) : 0u;
        if (kk) {
          j = b - (int32_t)kk;
          uint64_t code = pick_word<W>(rd, (uint32_t)j >> 5) >> ((j & 31) * 2);
          if (W > 1 && (j & 31) + (int32_t)kk > 32)
            code |= pick_word<W>(rd, ((uint32_t)j >> 5) + 1) << (64 - (j & 31) * 2);
          code &= (1ull << (2 * kk)) - 1ull;
          const uint32_t* tab = p.ftab + tab_off + lex_code(code, kk);
          lo = tab[0];
          hi = tab[1];
        }
        while (j > a && hi > lo && (hi - lo) > p.wstop) {
          --j;
          const uint32_t c = (uint32_t)(pick_word<W>(rd, (uint32_t)j >> 5) >> ((j & 31) * 2)) & 3u;
          const uint4 vl = lib.block(lo >> 5);
          uint4 vh = vl;
          if ((hi >> 5) != (lo >> 5)) vh = lib.block(hi >> 5);
          lo = lib.lf(c, lo, vl);
          hi = lib.lf(c, hi, vh);
        }
        stop_[k] = j;
        end_[k] = b;
        listed[k] = true;
        if (hi > lo && hi - lo > p.max_rows) {
          // a seed this repetitive cannot have a unique best hit; do not walk 10^5+ rows
          if (!LIST && !p.count32) count = 255u;
          hi = lo + p.max_rows;
        }
        const uint32_t need_before = (uint32_t)j, need_after = (uint32_t)(L - j);
        const bool prefilter = p.ctx && hi - lo >= kCtxMinRows;
        const uint2 probe = prefilter ? context_probe<W>(rd, need_before, need_after) : make_uint2(0u, 0u);
        for (uint32_t i = lo; i < hi; ++i) {
          if (prefilter && (int32_t)context_mismatches(p.ctx[i], probe) > p.max_mm_total) continue;
          const uint64_t row = p.sa[i];
          const uint32_t before = (uint32_t)(row >> 32) & 255u, after = (uint32_t)(row >> 40) & 255u;
          if ((need_before > before) | (need_after > after)) continue;
          const uint32_t s = (uint32_t)row - need_before;
          uint64_t m[W];
          uint32_t mm_total = 0, mm_seed = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            m[w] = 0ull;
            const int32_t nb = min(32, L - 32 * w);
            if (nb > 0) {
              const uint64_t x = lib.window(s + 32u * w) ^ rd[w];
              m[w] = (((x | (x >> 1)) & kOdd) | nm[w]) & low_bits(2 * nb);
              mm_total += (uint32_t)__popcll(m[w]);
              const int32_t ns = min(nb, max(0, p.seed_len - 32 * w));
              mm_seed += (uint32_t)__popcll(m[w] & low_bits(2 * ns));
            }
          }
          if (((int32_t)mm_seed > p.max_mm_seed) | ((int32_t)mm_total > p.max_mm_total)) continue;
          // already counted if an earlier piece's searched bases match exactly here
          bool seen = false;
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (q >= k || !listed[q]) continue;
            bool exact = true;
#pragma unroll
            for (int w = 0; w < W; ++w) {
              const int32_t lo_b = max(stop_[q] - 32 * w, 0), hi_b = min(end_[q] - 32 * w, 32);
              if (hi_b > lo_b) exact &= (m[w] & low_bits(2 * hi_b) & ~low_bits(2 * lo_b)) == 0ull;
            }
            seen |= exact;
          }
          if (seen) continue;
          if (LIST) {
            if (mm_total != want_mm) continue;
            const uint64_t slot = out_base + count++;
            if (slot >= p.out_cap) continue;
            uint32_t sg = (uint32_t)(row >> 48);
            if (sg == 0xFFFFu) {
              sg = p.chunk_seg[s >> 5];
              while (p.seg_start[sg + 1] <= s) ++sg;
            }
            p.out_ref[slot] = (int32_t)p.seg_ref[sg];
            p.out_pos[slot] = (int32_t)(s - p.seg_start[sg] + p.seg_off[sg]);
          } else if (mm_total < best_mm) {
            best_mm = mm_total;
            if (count != 255u || p.count32) count = 1u;
          } else if (mm_total == best_mm && (count < 255u || p.count32)) {
            ++count;
          }
        }
      }
    }
    if (!LIST) {
      p.best_mm[r] = (uint8_t)best_mm;
      if (p.count32) p.count32[r] = best_mm == 255u ? 0u : count;
      else p.count[r] = (uint8_t)(best_mm == 255u ? 0u : count);
    }
  }
}

// ---------------------------------------------------------------------------
// Tally (SUM:34-66).  Bins are privatised in LDS per workgroup (`LDSH`) and flushed
// with one global atomic per non-zero bin.  trimmedUniq, the one bin every lane hits,
// is aggregated across the wave (a ballot popcount, one add per wave).  Category and
// per-miRNA bins go straight to 64-bit LDS atomics: leader-loop aggregation over them
// was measured and costs more shuffles than the bank conflicts it saves on this
// workload (100 M reads: 3.4 ms with a 4-round leader loop on every bin, 0.56 ms with
// one round on the category bin, 0.40 ms with plain LDS atomics).
// ---------------------------------------------------------------------------
template <bool LDSH>
__global__ void __launch_bounds__(kTallyThreads) tally_kernel(const TallyParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  unsigned long long* hist = reinterpret_cast<unsigned long long*>(smem);
  const uint32_t S = p.n_samples, M = p.n_mirna;
  const uint32_t n_bins = 2 * M * S + (p.n_pass + 1) * S + S;
  const uint32_t cat0 = 2 * M * S, uniq0 = cat0 + (p.n_pass + 1) * S;
  unsigned long long* g = reinterpret_cast<unsigned long long*>(p.counts);
  if (LDSH) {
    for (uint32_t i = threadIdx.x; i < n_bins; i += kTallyThreads) hist[i] = 0ull;
    __syncthreads();
  }
  unsigned long long* h = LDSH ? hist : g;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t n_round = ((p.n + kTallyThreads - 1) / kTallyThreads) * kTallyThreads;

  for (uint64_t r = (uint64_t)blockIdx.x * kTallyThreads + threadIdx.x; r < n_round;
       r += (uint64_t)gridDim.x * kTallyThreads) {
    const bool active = r < p.n;  // whole waves stay in the loop: ballots below
    const int32_t pass = active ? p.pass_id[r] : -1;
    // an unclaimed read (pass -1) must not match a disabled (-1) canon/isomiR pass
    const bool canon = active && pass >= 0 && pass == p.canon_pass;
    const bool iso = active && pass >= 0 && pass == p.isomir_pass;
    const uint32_t ref = (canon || iso) ? (uint32_t)p.ref_id[r] : 0u;
    const uint32_t cat = pass < 0 ? p.n_pass : (uint32_t)pass;
    for (uint32_t s = 0; s < S; ++s) {
      const unsigned long long q = active ? p.quant[r * S + s] : 0ull;
      const bool hit = q != 0ull;
      const uint64_t hits = __ballot(hit);
      if (!hits) continue;
      if (lane == 0) atomicAdd(&h[uniq0 + s], (unsigned long long)__popcll(hits));
      if (hit) atomicAdd(&h[cat0 + cat * S + s], q);
      if (hit && (canon || iso)) atomicAdd(&h[ref * S + s], q);
      if (hit && canon) atomicAdd(&h[M * S + ref * S + s], q);
    }
  }
  if (LDSH) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_bins; i += kTallyThreads) {
      const unsigned long long v = hist[i];
      if (v) atomicAdd(&g[i], v);
    }
  }
}

__global__ void export_pass_counts_kernel(const uint64_t* stats, uint32_t n_pass,
                                          uint64_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pass) {
    out[2 * i] = stats[5 * i];
    out[2 * i + 1] = stats[5 * i + 1];
  }
}

// ---------------------------------------------------------------------------
// Launch helpers (host)
// ---------------------------------------------------------------------------
template <int W, bool LDSI, bool LDST, bool STRATA, bool CTX, bool KBITS>
static hipError_t launch_match_k(const MatchParams& p, uint32_t grid, uint32_t lds_bytes,
                                 hipStream_t stream) {
  auto kern = match_kernel<W, LDSI, LDST, STRATA, CTX, KBITS>;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(MatchBlock<LDSI>::kThreads), lds_bytes, stream, p);
  return hipGetLastError();
}

template <int W, bool LDSI, bool LDST, bool STRATA, bool CTX>
static hipError_t launch_match_x(const MatchParams& p, uint32_t grid, uint32_t lds_bytes,
                                 hipStream_t stream) {
  // the bitmap exists for small libraries and is used when the launch reserved LDS for it
  return p.kbits ? launch_match_k<W, LDSI, LDST, STRATA, CTX, true>(p, grid, lds_bytes, stream)
                 : launch_match_k<W, LDSI, LDST, STRATA, CTX, false>(p, grid, lds_bytes, stream);
}

template <int W, bool LDSI, bool LDST, bool STRATA>
static hipError_t launch_match_s(const MatchParams& p, uint32_t grid, uint32_t lds_bytes,
                                 hipStream_t stream) {
  // a library with a row-context array is too large for its text to be staged in LDS
  if (p.ctx && !LDST) return launch_match_x<W, LDSI, false, STRATA, true>(p, grid, lds_bytes, stream);
  return launch_match_x<W, LDSI, LDST, STRATA, false>(p, grid, lds_bytes, stream);
}

template <int W, bool LDSI, bool LDST>
static hipError_t launch_match_t(const MatchParams& p, uint32_t grid, uint32_t lds_bytes,
                                 hipStream_t stream) {
  return p.max_mm_seed == 2 ? launch_match_s<W, LDSI, LDST, true>(p, grid, lds_bytes, stream)
                            : launch_match_s<W, LDSI, LDST, false>(p, grid, lds_bytes, stream);
}

template <int W>
static hipError_t launch_match_w(const MatchParams& p, int lds_mode, uint32_t grid,
                                 uint32_t lds_bytes, hipStream_t stream) {
  switch (lds_mode) {
    case 3: return launch_match_t<W, false, true>(p, grid, lds_bytes, stream);
    case 2: return launch_match_t<W, true, true>(p, grid, lds_bytes, stream);
    case 1: return launch_match_t<W, true, false>(p, grid, lds_bytes, stream);
    default: return launch_match_t<W, false, false>(p, grid, lds_bytes, stream);
  }
}

hipError_t launch_match(const MatchParams& p, uint32_t words_per_read, int lds_mode,
                        uint32_t grid, uint32_t lds_bytes, hipStream_t stream) {
  switch (words_per_read) {
    case 1: return launch_match_w<1>(p, lds_mode, grid, lds_bytes, stream);
    case 2: return launch_match_w<2>(p, lds_mode, grid, lds_bytes, stream);
    case 4: return launch_match_w<4>(p, lds_mode, grid, lds_bytes, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_tally(const TallyParams& p, bool lds_hist, uint32_t grid,
                        uint32_t lds_bytes, hipStream_t stream) {
  if (lds_hist) {
    auto kern = tally_kernel<true>;
    if (lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTallyThreads), lds_bytes, stream, p);
  } else {
    hipLaunchKernelGGL(tally_kernel<false>, dim3(grid), dim3(kTallyThreads), 0, stream, p);
  }
  return hipGetLastError();
}

hipError_t launch_count(const CountParams& p, uint32_t words_per_read, uint32_t grid, uint32_t lds_bytes,
                        hipStream_t stream) {
#define MRG_COUNT(W_)                                                                                     \
  if (p.out_ref)                                                                                         \
    hipLaunchKernelGGL((count_kernel<W_, true>), dim3(grid), dim3(kCountThreads), lds_bytes, stream, p); \
  else                                                                                                   \
    hipLaunchKernelGGL((count_kernel<W_, false>), dim3(grid), dim3(kCountThreads), lds_bytes, stream, p);
  switch (words_per_read) {
    case 1: MRG_COUNT(1) break;
    case 2: MRG_COUNT(2) break;
    case 4: MRG_COUNT(4) break;
    default: return hipErrorInvalidValue;
  }
#undef MRG_COUNT
  return hipGetLastError();
}

hipError_t launch_export_pass_counts(const uint64_t* stats, uint32_t n_pass, uint64_t* out,
                                     hipStream_t stream) {
  hipLaunchKernelGGL(export_pass_counts_kernel, dim3(1), dim3(64), 0, stream, stats, n_pass, out);
  return hipGetLastError();
}

}  // namespace mrg
