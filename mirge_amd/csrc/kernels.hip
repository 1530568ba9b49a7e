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
template <bool LDST, bool STRATA>
struct RowsPerTrip {
  // (two per trip for the non-STRATA LDS-text passes, whose searches mostly end in 0-2 rows, was
  // measured: pass 0 1.18 -> 1.43 ms -- the second trip of a paralog tie costs a full L2 latency)
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


// the 9 bases of the read starting at base `at`, first base in the low two bits
template <int W>
__device__ __forceinline__ uint32_t read_bits9(const uint64_t (&rd)[W], uint32_t at) {
  uint64_t v = pick_word<W>(rd, at >> 5) >> ((at & 31u) * 2u);
  if (W > 1 && (at & 31u) + kKmerBitsK > 32u) v |= pick_word<W>(rd, (at >> 5) + 1u) << (64u - (at & 31u) * 2u);
  return (uint32_t)v & ((1u << (2u * kKmerBitsK)) - 1u);
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
template <int W, class LibT, class P>
__device__ __forceinline__ void verify_row(const LibT& lib, const P& p, const uint64_t row,
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
      L0 = p.uniform_len ? p.uniform_len : (uint32_t)p.lens[r];  // (a batch of one length: no 1-byte gathers)
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
    if (eligible && p.count_processed) ++c_processed;

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
      for (int32_t K = (STRATA ? p.k_first : Kfull); K <= (STRATA ? p.k_last : Kfull); ++K) {
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
          constexpr uint32_t kRowsPerTrip = RowsPerTrip<LDST, STRATA>::value;
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

    // (a launch that stops before the last stratum keeps a hit only if it is final: fewer
    // mismatches than the strata searched so far guarantee to have found)
    const bool aligned = best != ~0ull && (!STRATA || p.k_last > p.max_mm_seed || (int32_t)(best >> 32) < p.k_last);
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
      if (p.packed) {
        p.packed[r] = pack_assignment(p.pass_index, ref, pos, (uint32_t)(best >> 32));
      } else {
        p.pass_id[r] = (int8_t)p.pass_index;
        p.ref_id[r] = (int32_t)ref;
        p.pos[r] = (int32_t)pos;
        p.mm[r] = (uint8_t)(best >> 32);
      }
    } else if (active && !p.idx_out) {
      // last pass: whatever is still unclaimed stays unannotated (no memset needed)
      if (p.packed) {
        p.packed[r] = 0u;
      } else {
        p.pass_id[r] = (int8_t)-1;
        p.ref_id[r] = -1;
        p.pos[r] = -1;
        p.mm[r] = 0;
      }
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
  // Counters: summed over the wave, then over the workgroup in LDS, then ONE global atomic per
  // counter and workgroup.  (One per wave -- 8192 waves on five addresses of one cache line --
  // serialises in L2 at ~10 ns each: that alone was most of a 10 M-read launch.)
  const uint64_t t_processed = wave_sum(c_processed), t_aligned = wave_sum(c_aligned);
  const uint64_t t_steps = wave_sum(c_steps), t_cands = wave_sum(c_cands);
  const uint64_t t_lookups = wave_sum(c_lookups);
  unsigned long long* wg_cnt = reinterpret_cast<unsigned long long*>(wave_slots + 4u * 16u);
  if (threadIdx.x < 5) wg_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  if (lane == 0) {
    if (t_processed) atomicAdd(&wg_cnt[0], (unsigned long long)t_processed);
    if (t_aligned) atomicAdd(&wg_cnt[1], (unsigned long long)t_aligned);
    if (t_steps) atomicAdd(&wg_cnt[2], (unsigned long long)t_steps);
    if (t_cands) atomicAdd(&wg_cnt[3], (unsigned long long)t_cands);
    if (t_lookups) atomicAdd(&wg_cnt[4], (unsigned long long)t_lookups);
  }
  __syncthreads();
  if (threadIdx.x < 5 && wg_cnt[threadIdx.x])
    atomicAdd((unsigned long long*)&p.counters[threadIdx.x], wg_cnt[threadIdx.x]);
  if (p.idx_out && threadIdx.x == 0) p.out_count[blockIdx.x] = ctl[0];
}

// ---------------------------------------------------------------------------
// fused_kernel: consecutive cascade passes over ONE walk of the survivor list (see kernels.hpp).
//
// What the one-launch-per-pass kernels spend their time on for the libraries after the first is
// not memory: it is instruction issue.  Every wave walks the list (list entry -> read, two
// dependent HBM round trips per pass) and then runs all 64 lanes through every seed piece, every
// jump-table load and a four-way unrolled verification although only one lane in four (small,
// bitmap-filtered libraries) has anything to look up and a wave waits for its widest interval.
// Here the work is compacted at three levels:
//   reads  the walk happens once per group and is software-pipelined two chunks deep;
//   items  a ROUND = a run of sub-passes (all the bitmap-filtered ones together, each large
//          library on its own).  Every lane first runs the cheap part of all sub-passes of the
//          round on its read -- length window, poly-T rule, and per search region (computed once
//          and reused while it does not change) which libraries can hold each piece, from four
//          LDS reads of the round's 9-mer table -- and ends up with a bit mask of (sub-pass,
//          piece) WORK ITEMS that need the index.  The items of the 64 reads
//          of a wave are compacted (wave prefix sum, owner found by a six-step search over the
//          prefix): lane i takes item i, fetches the owner's read with a shuffle and that
//          sub-pass's library pointers from an LDS table, and does the jump-table load and the
//          (rare) LF steps;
//   rows   the suffix-array rows of those 64 intervals are compacted the same way: lane i takes
//          row i of the batch, whoever's interval it belongs to, loads it, verifies it against
//          the text and folds a valid alignment into the item's 16-byte LDS slot with a 64-bit
//          atomic min.  A 2-row and a 2000-row interval cost what their rows cost; there is no
//          "wide interval" special case and no lane idles behind a long one.
// The owner replays the item results in cascade order: a sub-pass claimed earlier hides the later
// ones, an exact hit of piece 0 hides piece 1 -- exactly the items the sequential search would not
// have issued, so the per-pass processed / aligned / steps / candidates / lookups counters stay
// those of one launch per pass although the work itself was speculative.
// Libraries are served from L2/HBM (a small one's packed text optionally from LDS); LDS holds per
// round one interleaved 9-mer table ("which libraries of the round have a 9-mer with this code":
// one read answers a 9-mer for all of them), the sub-pass table, the result slots and the counters.  128
// VGPRs per lane (16 waves = one 1024-thread workgroup per CU): the pipelined walk and the
// per-lane library pointers do not fit 64 without spilling, and with dense lanes 16 waves keep
// more useful loads in flight than 32 sparse ones.
// ---------------------------------------------------------------------------
namespace {

constexpr uint32_t kSubWords = 40u;  // LDS words per sub-pass table entry
enum SubWord : uint32_t {
  SW_FTAB = 0, SW_SA = 2, SW_TEXT = 4, SW_CTX = 6, SW_BLOCKS = 8, SW_SUPER = 10, SW_TABK = 12, SW_TABOFF = 16,
  SW_N = 20, SW_PRIMARY = 21, SW_POLICY = 22, SW_TRIMS = 23, SW_TEXT_LDS = 24, SW_SIMPLE = 25, SW_PASSIDX = 26,
  SW_SEGSTART = 28, SW_SEGREF = 30, SW_SEGOFF = 32, SW_CHUNKSEG = 34, SW_SA16 = 36
};
constexpr uint32_t kNoQ = 0xFFu;
constexpr uint32_t kRowSlice = 1u << 20;  // rows of one interval taken into one compaction sweep

struct ItemPolicy {  // what verify_row reads
  int32_t seed_len, max_mm_seed, max_mm_total;
  const uint64_t* nmask;
};

// text access of one item: packed text in LDS when the launch staged it, else global
struct ItemLib {
  const uint32_t* gtext;
  const uint32_t* stext;  // null = not staged
  __device__ __forceinline__ uint64_t window(uint32_t pos) const {
    const uint32_t i = pos >> 4, sh = (pos & 15) * 2;
    uint32_t w0, w1, w2;
    if (stext) {
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

__device__ __forceinline__ const void* lds_ptr(const uint32_t* tab, uint32_t w) {
  return reinterpret_cast<const void*>((uint64_t)tab[w] | ((uint64_t)tab[w + 1] << 32));
}

// number of trailing T of a read (RAP:664-686; an N is never a T)
template <int W>
__device__ __forceinline__ int32_t trailing_t(const uint64_t (&rd)[W], const uint64_t (&nm)[W], int32_t L) {
  int32_t hb = -1;  // highest base that is not T
#pragma unroll
  for (int k = W - 1; k >= 0; --k) {
    const int32_t nb = min(32, max(0, L - 32 * k));
    const uint64_t x = (~rd[k] | nm[k] | (nm[k] << 1)) & low_bits(2 * nb);
    if (hb < 0 && x != 0ull) hb = 32 * k + ((63 - __clzll((long long)x)) >> 1);
  }
  return L - 1 - hb;
}

template <int W>
__device__ __forceinline__ bool piece_has_n(const uint64_t (&nm)[W], int32_t a, int32_t b) {
  bool has_n = false;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const int32_t lo_b = max(a - 32 * w, 0), hi_b = min(b - 32 * w, 32);
    if (hi_b > lo_b) has_n |= (nm[w] & low_bits(2 * hi_b) & ~low_bits(2 * lo_b)) != 0ull;
  }
  return has_n;
}

// 2c bits of the read starting at base `at` (c <= 16)
template <int W>
__device__ __forceinline__ uint32_t read_bits32(const uint64_t (&rd)[W], uint32_t at, uint32_t c) {
  uint64_t v = pick_word<W>(rd, at >> 5) >> ((at & 31u) * 2u);
  if (W > 1 && (at & 31u) + c > 32u) v |= pick_word<W>(rd, (at >> 5) + 1u) << (64u - (at & 31u) * 2u);
  return (uint32_t)(v & low_bits(2u * c));
}

// Mismatches a wide row's stored context (fm_index.hpp: fill_wide_rows) shows against a read whose
// seed search stopped with `need_before` read bases left of the row's position and `need_after`
// from it on: the <= 16 bases before the position and the bases 8..23 after it.  A lower bound of
// the alignment's mismatches (a read N, code 0, can only add true mismatches; bases across a
// segment end only occur in alignments that are invalid anyway): "more than the pass allows" is
// final.
template <int W>
__device__ __forceinline__ uint32_t wide_row_mismatches(const uint64_t (&rd)[W], uint32_t need_before,
                                                        uint32_t need_after, uint32_t left, uint32_t right) {
  uint32_t mm = 0;
  const uint32_t c = min(need_before, 16u);
  if (c) {
    const uint32_t want = read_bits32<W>(rd, need_before - c, c);
    const uint32_t x = (left >> (32u - 2u * c)) ^ want;  // c == 16: shift by 0
    mm += (uint32_t)__popc((x | (x >> 1)) & 0x55555555u & (uint32_t)low_bits(2u * c));
  }
  if (need_after > 8u) {
    const uint32_t c2 = min(need_after, 24u) - 8u;
    const uint32_t want = read_bits32<W>(rd, need_before + 8u, c2);
    const uint32_t x = (right ^ want) & (uint32_t)low_bits(2u * c2);
    mm += (uint32_t)__popc((x | (x >> 1)) & 0x55555555u);
  }
  return mm;
}

// first lane whose inclusive prefix exceeds x (x < total): six shuffles (stratum_kernel, whose 64
// registers have no room for the owner map below)
__device__ __forceinline__ uint32_t owner_of(uint32_t incl, uint32_t x) {
  uint32_t o = 0;
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    const uint32_t v = __shfl(incl, (int)(o + step - 1u), 64);
    if (v <= x) o += (uint32_t)step;
  }
  return o;
}

// Which lane owns element x of a compaction sweep, without a search.  Per wave 128 bytes of LDS:
// r2l[rank] = lane of the rank-th lane that has elements (written once per sweep), ends[b] = 1 when
// some lane's LAST element is element rb + b of the current batch of 64.  The owner of element rb + b
// is the first such lane whose elements end at or after it = the lane of rank (owners finished in
// earlier batches) + (ends flagged strictly before b): two byte stores, two byte loads, a ballot
// and a lane-masked bit count instead of six dependent shuffles.
struct OwnerMap {
  uint8_t* ends;
  uint8_t* r2l;
  uint32_t done;  // owners whose elements ended in earlier batches (wave-uniform)
  __device__ __forceinline__ void begin(bool nonempty, uint32_t lane) {
    const uint64_t m = __ballot(nonempty);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (nonempty) r2l[rank] = (uint8_t)lane;
    done = 0u;
  }
  // incl = inclusive prefix of my element count, rb = first element of the batch
  __device__ __forceinline__ uint32_t find(bool nonempty, uint32_t incl, uint32_t rb, uint32_t lane) {
    ends[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t e = incl - 1u - rb;  // my last element, relative to the batch (wraps when before it)
    if (nonempty && e < 64u) ends[e] = 1;
    __builtin_amdgcn_wave_barrier();
    const uint64_t m = __ballot(ends[lane] != 0);
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t o = r2l[min(done + before, 63u)];
    done += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    return o;
  }
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t /*lane*/) {
  const int x = (int)v;
  int s = x + __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  s += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);         // row_shr:2
  s += __builtin_amdgcn_update_dpp(0, x, 0x113, 0xf, 0xf, false);         // row_shr:3
  s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xe, false);         // row_shr:4, lanes 4-15 of a row
  s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xc, false);         // row_shr:8, lanes 8-15
  s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);         // row_bcast:15 into rows 1 and 3
  s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);         // row_bcast:31 into rows 2 and 3
  return (uint32_t)s;
}

// bits 0..3 of x to bits 0, 16, 32, 48
__device__ __forceinline__ uint64_t spread4(uint32_t x) {
  return (uint64_t)(x & 1u) | ((uint64_t)(x & 2u) << 15) | ((uint64_t)(x & 4u) << 30) | ((uint64_t)(x & 8u) << 45);
}

}  // namespace

// PAIRS: a one-mismatch sub-pass on a large library may carry pair tables (kernels.hpp: SubPass);
// a separate instantiation so that launches without them keep the plain piece items.
template <int W, bool PAIRS>
__global__ void __launch_bounds__(1024, 4) fused_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr uint32_t BLOCK = 1024u;
  // ---- LDS carve: [bitmaps][texts][sub-pass table][result slots][64-bit counters][control]
  uint32_t* skb = smem;
  uint32_t* stab = smem + p.kb_words + p.txt_words;
  uint4* slots = reinterpret_cast<uint4*>(stab + kMaxFused * kSubWords);  // 64 per wave
  unsigned long long* cnt64 = reinterpret_cast<unsigned long long*>(slots + BLOCK);
  uint32_t* ctl = reinterpret_cast<uint32_t*>(cnt64 + kMaxFused * 3u * kFusedCntReplicas);
  uint8_t* omap_lds = reinterpret_cast<uint8_t*>(ctl + 4);  // 128 B per wave
  for (uint32_t s = 0; s < p.n_sub; ++s) {
    const SubPass& sp = p.sub[s];
    if (sp.text_lds_words) {
      const uint4* src = reinterpret_cast<const uint4*>(sp.text);
      uint4* dst = reinterpret_cast<uint4*>(smem + sp.text_lds_off);
      for (uint32_t i = threadIdx.x; i < sp.text_lds_words / 4u; i += BLOCK) dst[i] = src[i];
    }
    if (threadIdx.x == 0) {
      uint32_t* t = stab + s * kSubWords;
      auto put = [&](uint32_t w, const void* ptr) {
        t[w] = (uint32_t)(uint64_t)ptr;
        t[w + 1] = (uint32_t)((uint64_t)ptr >> 32);
      };
      put(SW_FTAB, sp.ftab);
      put(SW_SA, sp.sa);
      put(SW_TEXT, sp.text);
      put(SW_CTX, sp.ctx);
      put(SW_BLOCKS, sp.blocks);
      put(SW_SUPER, sp.super);
      put(SW_SEGSTART, sp.seg_start);
      put(SW_SEGREF, sp.seg_ref);
      put(SW_SEGOFF, sp.seg_off);
      put(SW_CHUNKSEG, sp.chunk_seg);
      put(SW_SA16, sp.sa16);
      for (int i = 0; i < 4; ++i) {
        t[SW_TABK + i] = sp.tabs.k[i];
        t[SW_TABOFF + i] = sp.tabs.off[i];
      }
      t[SW_N] = sp.n;
      t[SW_PRIMARY] = sp.primary;
      t[SW_POLICY] = (uint32_t)min(sp.seed_len, 0xFFFF) | ((uint32_t)sp.max_mm_seed << 16) | ((uint32_t)sp.max_mm_total << 24);
      t[SW_TRIMS] = (uint32_t)sp.trim5 | ((uint32_t)sp.trim3 << 8) | (sp.poly_t ? 1u << 16 : 0u);
      t[SW_TEXT_LDS] = sp.text_lds_words ? sp.text_lds_off : 0xFFFFFFFFu;
      t[SW_SIMPLE] = sp.simple_segs;
      t[SW_PASSIDX] = (uint32_t)sp.pass_index;
    }
  }
  for (uint32_t rnd = 0; rnd < p.n_rounds; ++rnd) {
    if (!p.round_kb_log2[rnd]) continue;
    const uint4* src = reinterpret_cast<const uint4*>(p.round_kb_src[rnd]);
    uint4* dst = reinterpret_cast<uint4*>(skb + p.round_kb_off[rnd]);
    const uint32_t n16 = ((1u << p.round_kb_log2[rnd]) * (uint32_t)p.round_kb_bits[rnd]) / 128u;
    for (uint32_t i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
  }
  for (uint32_t i = threadIdx.x; i < kMaxFused * 3u * kFusedCntReplicas; i += BLOCK) cnt64[i] = 0ull;
  if (threadIdx.x == 0) {
    ctl[0] = 0u;
    ctl[1] = 0u;
  }
  __syncthreads();
  if (p.idx_in) {
    uint32_t mx = 0;
    for (uint32_t sgi = threadIdx.x; sgi < p.in_nseg; sgi += BLOCK) mx = max(mx, p.in_count[sgi]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, off, 64));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&ctl[1], mx);
  }
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* my_slots = slots + wave * 64u;
  OwnerMap omap;
  omap.ends = omap_lds + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave) * 128u;  // wave-uniform: scalar registers
  omap.r2l = omap.ends + 64u;
  omap.done = 0u;
  unsigned long long* my_slot_keys = reinterpret_cast<unsigned long long*>(my_slots);  // key of slot i at [2 i]
  const bool has_nm = p.nmask != nullptr;
  const uint32_t in_nseg = p.idx_in ? p.in_nseg : 1u;
  const uint32_t depth_chunks = p.idx_in ? (ctl[1] + BLOCK - 1) / BLOCK : (p.n_total + BLOCK - 1) / BLOCK;
  const uint32_t n_chunks = in_nseg * depth_chunks;
  // per-lane 0/1 counters of the sub-passes, 16 bits each (a lane sees < 65536 reads per launch:
  // the host falls back to one launch per pass otherwise): [0] sub-passes 0-3, [1] 4-7
  uint64_t acc_offered[2] = {0ull, 0ull}, acc_aligned[2] = {0ull, 0ull};

  // ---- the walk, software-pipelined: read of chunk + grid and list entry of chunk + 2 grid in flight
  auto fetch_index = [&](uint32_t chunk, uint32_t& r_out) -> bool {
    if (chunk >= n_chunks) return false;
    const uint32_t sgi = chunk % in_nseg, depth = chunk / in_nseg;
    const uint32_t t = depth * BLOCK + threadIdx.x;
    const bool act = t < (p.idx_in ? p.in_count[sgi] : p.n_total);
    r_out = 0;
    if (act) r_out = p.idx_in ? p.idx_in[(size_t)sgi * p.in_seg_cap + t] : t;
    return act;
  };
  uint32_t r_c = 0, r_b = 0;
  bool act_b = fetch_index(blockIdx.x, r_b);
  bool act_c = fetch_index(blockIdx.x + gridDim.x, r_c);
  uint64_t rd_b[W], nm_b[W];
  uint32_t L_b = 0;
#pragma unroll
  for (int k = 0; k < W; ++k) rd_b[k] = nm_b[k] = 0ull;
  if (act_b) {
    L_b = p.uniform_len ? p.uniform_len : (uint32_t)p.lens[r_b];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      rd_b[k] = p.reads[(size_t)k * p.n_total + r_b];
      nm_b[k] = has_nm ? p.nmask[(size_t)k * p.n_total + r_b] : 0ull;
    }
  }

  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const bool active = act_b;
    const uint32_t r = r_b;
    uint64_t rd0[W], nm0[W];
    const uint32_t L0 = L_b;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      rd0[k] = rd_b[k];
      nm0[k] = nm_b[k];
    }
    act_b = act_c;
    r_b = r_c;
    L_b = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) rd_b[k] = nm_b[k] = 0ull;
    if (act_b) {
      L_b = p.uniform_len ? p.uniform_len : (uint32_t)p.lens[r_b];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        rd_b[k] = p.reads[(size_t)k * p.n_total + r_b];
        nm_b[k] = has_nm ? p.nmask[(size_t)k * p.n_total + r_b] : 0ull;
      }
    }
    act_c = fetch_index(chunk + 2u * gridDim.x, r_c);

    bool claimed = !active;  // lanes without a read never take part
    // What the filter phase knows about the current search region (R bases after a 5' trim of t5),
    // computed once and reused by every sub-pass that searches the same region -- for reads of at
    // most 28 nt that is all the -n 1 / -v 1 / -n 0 passes: per piece (the two halves, and the
    // whole region for one-piece policies) which libraries of the round can hold it, as a bit per
    // sub-pass (first-9 and last-9 tests in the round's interleaved table, pieces of < 9 bases
    // pass, a piece holding an N never matches), and the same without any table.
    int32_t g_R = -1, g_t5 = -1;
    uint32_t g_pm = 0;   // bits 0-7 piece 0 of two, 8-15 piece 1 of two, 16-23 the whole region
    uint32_t g_nn = 0;   // bit 0 / 8 / 16: that piece holds no N (sub-passes without a table)
    uint32_t g_pa = 0;   // pair-searched sub-pass: bit pr = both anchors of pair pr hold no N
#pragma unroll 1
    for (uint32_t rnd = 0; rnd < p.n_rounds; ++rnd) {
      const uint32_t s0 = p.round_first[rnd], ns = p.round_count[rnd];
      const uint32_t kb_log2 = p.round_kb_log2[rnd], kb_bits = p.round_kb_bits[rnd];
      const uint32_t* kb = skb + p.round_kb_off[rnd];
      g_R = -1;  // the library bits belong to one round's table
      // ================= phase A: which (sub-pass, piece) items does my read need? =================
      uint32_t need = 0;   // bit 2q + k: piece k of sub-pass s0 + q goes to the index
      uint32_t elig = 0;   // bit q: my read is in sub-pass s0 + q's FASTA if nothing claims it before
#pragma unroll 1
      for (uint32_t q = 0; q < ns; ++q) {
        const SubPass& sp = p.sub[s0 + q];
        bool el = !claimed && (int32_t)L0 >= sp.min_len && (int32_t)L0 <= sp.max_len;
        int32_t L = (int32_t)L0;
        if (sp.poly_t) {
          const int32_t tail = trailing_t<W>(rd0, nm0, L);
          el = el && tail >= 3 && (L - tail) >= 11;
          L -= tail;
        }
        L -= sp.trim5 + sp.trim3;
        if (el) elig |= 1u << q;
        const int32_t R = min(L, sp.seed_len);
        const bool go = el && L > sp.max_mm_seed;
        if (go && (R != g_R || sp.trim5 != g_t5)) {
          uint64_t rd[W], nm[W];
#pragma unroll
          for (int k = 0; k < W; ++k) {
            rd[k] = rd0[k];
            nm[k] = nm0[k];
          }
          if (sp.trim5) {
            shift_out_5p<W>(rd, (uint32_t)sp.trim5);
            if (has_nm) shift_out_5p<W>(nm, (uint32_t)sp.trim5);
          }
          g_R = R;
          g_t5 = sp.trim5;
          const int32_t h = R >> 1;  // pieces [0, h), [h, R) and the whole [0, R)
          const bool n0 = has_nm && piece_has_n<W>(nm, 0, h), n1 = has_nm && piece_has_n<W>(nm, h, R);
          g_nn = (n0 ? 0u : 1u) | (n1 ? 0u : 1u << 8) | ((n0 || n1) ? 0u : 1u << 16);
          if (PAIRS && sp.pair_anchor) {
            const int32_t A = (int32_t)sp.pair_anchor;
            const bool a0 = has_nm && piece_has_n<W>(nm, 0, A), a1 = has_nm && piece_has_n<W>(nm, A, 2 * A),
                       a2 = has_nm && piece_has_n<W>(nm, 2 * A, 3 * A);
            g_pa = ((a0 || a1) ? 0u : 1u) | ((a1 || a2) ? 0u : 2u) | ((a0 || a2) ? 0u : 4u);
          }
          uint32_t mF = 0xFFu, mL = 0xFFu, m0L = 0xFFu, m1F = 0xFFu;
          if (kb_log2) {
            const uint32_t cmask = (1u << kb_log2) - 1u, emask = (1u << kb_bits) - 1u;
            auto libs_of = [&](uint32_t at) {
              const uint32_t bit = (read_bits9<W>(rd, at) & cmask) * kb_bits;
              return (kb[bit >> 5] >> (bit & 31u)) & emask;
            };
            if (R >= (int32_t)kKmerBitsK) {
              mF = libs_of(0u);
              mL = libs_of((uint32_t)R - kKmerBitsK);
            }
            if (h >= (int32_t)kKmerBitsK) m0L = libs_of((uint32_t)h - kKmerBitsK);
            if (R - h >= (int32_t)kKmerBitsK) m1F = libs_of((uint32_t)h);
          }
          const int32_t k9 = (int32_t)kKmerBitsK;
          const uint32_t P0 = h < k9 ? 0xFFu : (m0L & (h == k9 ? 0xFFu : mF));
          const uint32_t P1 = (R - h) < k9 ? 0xFFu : (mL & ((R - h) == k9 ? 0xFFu : m1F));
          const uint32_t Pw = R < k9 ? 0xFFu : (mL & (R == k9 ? 0xFFu : mF));
          g_pm = (n0 ? 0u : P0) | ((n1 ? 0u : P1) << 8) | (((n0 || n1) ? 0u : Pw) << 16);
        }
        // this sub-pass's view: bit 0 / 8 / 16 = piece 0 / piece 1 / whole region goes to the index
        const uint32_t m = sp.kb_bit != 0xFFu ? (g_pm >> sp.kb_bit) : g_nn;
        const uint32_t gb = go ? 1u : 0u;
        if (PAIRS && sp.pair_anchor && R >= 3 * (int32_t)sp.pair_anchor && R < 4 * (int32_t)sp.pair_anchor) {
          // three anchor pairs instead of two short pieces (the sub-pass is alone in its round: bits 0-2)
          need |= go ? g_pa : 0u;
        } else if (sp.max_mm_seed == 0) {
          need |= (gb & (m >> 16)) << (2u * q);
        } else {
          need |= ((gb & m) | ((gb & (m >> 8)) << 1)) << (2u * q);
        }
      }

      const bool pair_round = PAIRS && ns == 1u && p.sub[s0].pair_anchor != 0u;
      // ================= phase B: compact the items of the wave, one item per lane =================
      const uint32_t cnt = (uint32_t)__popc(need);
      const uint32_t incl = wave_incl_scan(cnt, lane);
      const uint32_t excl = incl - cnt;
      const uint32_t total = __shfl(incl, 63, 64);

      // owner-side replay state (cascade order = item order)
      uint64_t win_key = ~0ull;   // result of the sub-pass that claims the read
      uint32_t win_q = kNoQ;      // which one
      uint64_t q_best = ~0ull;    // best result inside the sub-pass being replayed
      uint32_t q_cur = kNoQ;

      OwnerMap imap = omap;  // (the rows sweeps below use the same LDS bytes between two item batches)
      imap.done = 0u;
      for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t item = base + lane;
        const bool has_item = item < total;
        // (re-register: the rows sweeps of the previous batch overwrote the rank table)
        const uint32_t i_done = imap.done;
        imap.begin(cnt != 0u, lane);
        imap.done = i_done;
        uint32_t o = imap.find(cnt != 0u, incl, base, lane);
        o = has_item ? o : lane;
        const uint32_t o_excl = __shfl(excl, (int)o, 64);
        uint32_t o_need = __shfl(need, (int)o, 64);
        uint64_t rd[W], nm[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
          rd[k] = __shfl(rd0[k], (int)o, 64);
          nm[k] = has_nm ? __shfl(nm0[k], (int)o, 64) : 0ull;
        }
        int32_t L = (int32_t)__shfl(L0, (int)o, 64);
        uint32_t combo = 0;
        if (has_item) {
          for (uint32_t jj = item - o_excl; jj > 0; --jj) o_need &= o_need - 1u;
          combo = (uint32_t)__ffs((int)o_need) - 1u;
        }
        // (a pair-searched sub-pass is alone in its round: its item bits 0-2 are the item number)
        const uint32_t q = pair_round ? 0u : combo >> 1, kpiece = pair_round ? combo : combo & 1u;
        const uint32_t* tab = stab + (s0 + q) * kSubWords;
        const uint32_t polw = tab[SW_POLICY], trimw = tab[SW_TRIMS];
        uint32_t c_steps = 0, lo = 0, hi = 0, nb_na = 0;
        if (has_item) {
          // the read this sub-pass searches: poly-T strip, then -5 / -3
          if (trimw >> 16) L -= trailing_t<W>(rd, nm, L);
          const uint32_t t5 = trimw & 0xFFu;
          L -= (int32_t)(t5 + ((trimw >> 8) & 0xFFu));
          if (t5) {
            shift_out_5p<W>(rd, t5);
            if (has_nm) shift_out_5p<W>(nm, t5);
          }
          const int32_t R = min(L, (int32_t)(polw & 0xFFFFu));
          const int32_t K = (int32_t)((polw >> 16) & 0xFFu) + 1;
          const uint32_t pA = pair_round ? p.sub[s0].pair_anchor : 0u;
          if (PAIRS && pA && R >= 3 * (int32_t)pA && R < 4 * (int32_t)pA) {
            // anchor pair kpiece = (0,1) (1,2) (0,2): one load of the gap's jump table
            const uint32_t ia = kpiece == 1u ? 1u : 0u, ja = kpiece == 0u ? 1u : 2u, t = ja - ia - 1u;
            const uint32_t kb = 2u * pA;
            const uint32_t vi = read_bits32<W>(rd, ia * pA, pA), vj = read_bits32<W>(rd, ja * pA, pA);
            const uint32_t* ft = p.sub[s0].pair_jump + (size_t)t * ((1ull << (2u * kb)) + 1ull) + (vi | (vj << kb));
            const uint32_t roff = p.sub[s0].pair_row_off[t];
            lo = ft[0] + roff;
            hi = ft[1] + roff;
            c_steps = 0x80000000u;
            nb_na = (ia * pA) | ((uint32_t)(L - (int32_t)(ia * pA)) << 8) | (pA << 16) | (1u << 24);  // bit 24: rows of the pair lists
          } else {
          const int32_t a = div_pieces(R * (int32_t)kpiece, K), b = div_pieces(R * ((int32_t)kpiece + 1), K);
          hi = tab[SW_N] + 1u;
          int32_t j = b;
          // the largest jump table the piece is long enough for
          uint32_t tk = 0, tab_off = 0;
          if (tab[SW_TABK]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t ki = tab[SW_TABK + i];
              const bool take = (b - a) >= (int32_t)ki;
              tk = take ? ki : tk;
              tab_off = take ? tab[SW_TABOFF + i] : tab_off;
            }
          }
          if (tk) {
            j = b - (int32_t)tk;
            uint64_t code = pick_word<W>(rd, (uint32_t)j >> 5) >> ((j & 31) * 2);
            if (W > 1 && (j & 31) + (int32_t)tk > 32)
              code |= pick_word<W>(rd, ((uint32_t)j >> 5) + 1) << (64 - (j & 31) * 2);
            code &= (1ull << (2 * tk)) - 1ull;
            const uint32_t* ft = reinterpret_cast<const uint32_t*>(lds_ptr(tab, SW_FTAB)) + tab_off + lex_code(code, tk);
            lo = ft[0];
            hi = ft[1];
            c_steps = 0x80000000u;  // bit 31: the item did a jump-table load
          }
          if (j > a && hi > lo && (hi - lo) > p.wstop) {
            Lib<false, false> lib;
            lib.gblocks = reinterpret_cast<const uint32_t*>(lds_ptr(tab, SW_BLOCKS));
            lib.gtext = nullptr;
            lib.sblocks = nullptr;
            lib.stext = nullptr;
            lib.ssuper = reinterpret_cast<const uint32_t*>(lds_ptr(tab, SW_SUPER));
            lib.primary = tab[SW_PRIMARY];
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
          }
          nb_na = (uint32_t)j | ((uint32_t)(L - j) << 8) | ((uint32_t)min(b - j, 255) << 16);  // + exactly matched bases
          }
        }
        uint32_t rem = (has_item && hi > lo) ? hi - lo : 0u;
        // result slot of my item: key = mm:8 | text position:32 | segment:16 | before:8 (all ones = none)
        my_slots[lane] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, rem, c_steps);
        __builtin_amdgcn_wave_barrier();

        // ---------- rows: every suffix-array row of the 64 intervals, one per lane ----------
        while (__ballot(rem != 0u)) {
          const uint32_t w = min(rem, kRowSlice);
          const uint32_t rincl = wave_incl_scan(w, lane);
          const uint32_t rtotal = __shfl(rincl, 63, 64);
          omap.begin(w != 0u, lane);
          for (uint32_t rb = 0; rb < rtotal; rb += 64u) {
            const uint32_t x = rb + lane;
            const bool has_row = x < rtotal;
            uint32_t t = omap.find(w != 0u, rincl, rb, lane);
            t = has_row ? t : lane;
            const uint32_t t_incl = __shfl(rincl, (int)t, 64), t_w = __shfl(w, (int)t, 64);
            const uint32_t t_lo = __shfl(lo, (int)t, 64);
            const uint32_t t_nbna = __shfl(nb_na, (int)t, 64);
            const uint32_t t_q = __shfl(q, (int)t, 64);
            const int32_t t_L = __shfl(L, (int)t, 64);
            uint64_t t_rd[W], t_nm[W];
#pragma unroll
            for (int k = 0; k < W; ++k) {
              t_rd[k] = __shfl(rd[k], (int)t, 64);
              t_nm[k] = has_nm ? __shfl(nm[k], (int)t, 64) : 0ull;
            }
            if (has_row) {
              const uint32_t* ttab = stab + (s0 + t_q) * kSubWords;
              const uint32_t tpol = ttab[SW_POLICY];
              ItemPolicy pol;
              pol.seed_len = (int32_t)(tpol & 0xFFFFu);
              pol.max_mm_seed = (int32_t)((tpol >> 16) & 0xFFu);
              pol.max_mm_total = (int32_t)(tpol >> 24);
              pol.nmask = p.nmask;
              const uint32_t i = t_lo + (x - (t_incl - t_w));
              const uint32_t need_before = t_nbna & 0xFFu, need_after = (t_nbna >> 8) & 0xFFu, exact_part = (t_nbna >> 16) & 0xFFu;
              uint64_t row = 0ull;
              uint64_t best = ~0ull;
              uint32_t best_seg = 0xFFFFu, best_before = 255u;
              const uint4* sa16 = reinterpret_cast<const uint4*>(lds_ptr(ttab, SW_SA16));
              if (PAIRS && (t_nbna >> 24)) {
                // a row of a pair list (8-byte row: the text decides)
                row = p.sub[s0].pair_rows[i];
                ItemLib lib;
                lib.gtext = reinterpret_cast<const uint32_t*>(lds_ptr(ttab, SW_TEXT));
                lib.stext = nullptr;
                verify_row<W>(lib, pol, row, t_rd, t_nm, t_L, need_before, need_after, best, best_seg, best_before);
              } else if (sa16) {
                // large library: one 16-byte load = the row and 32 bases around the seed; a false
                // candidate (nearly all of them) never asks for its text window
                const uint4 wr = sa16[i];
                row = (uint64_t)wr.x | ((uint64_t)wr.y << 32);
                const uint32_t mm = wide_row_mismatches<W>(t_rd, need_before, need_after, wr.z, wr.w);
                bool any_n = false;
#pragma unroll
                for (int k = 0; k < W; ++k) any_n |= t_nm[k] != 0ull;
                if (need_before <= 16u && need_after <= 24u && exact_part >= 8u && !any_n && t_L <= pol.seed_len) {
                  // the stored context shows every base the exact part of the seed (>= 8 bases from
                  // the row's position on) does not: `mm` IS the alignment's mismatch count (the
                  // read lies inside the seed region, so seed and total mismatches coincide)
                  const uint32_t before = wr.y & 255u, after = (wr.y >> 8) & 255u;
                  if (need_before <= before && need_after <= after && (int32_t)mm <= pol.max_mm_seed &&
                      (int32_t)mm <= pol.max_mm_total) {
                    best = ((uint64_t)mm << 32) | (uint64_t)(wr.x - need_before);
                    best_seg = wr.y >> 16;
                    best_before = before < 255u ? before - need_before : 255u;
                  }
                } else if ((int32_t)mm <= pol.max_mm_total) {
                  ItemLib lib;
                  lib.gtext = reinterpret_cast<const uint32_t*>(lds_ptr(ttab, SW_TEXT));
                  lib.stext = nullptr;
                  verify_row<W>(lib, pol, row, t_rd, t_nm, t_L, need_before, need_after, best, best_seg, best_before);
                }
              } else {
                row = reinterpret_cast<const uint64_t*>(lds_ptr(ttab, SW_SA))[i];
                ItemLib lib;
                lib.gtext = reinterpret_cast<const uint32_t*>(lds_ptr(ttab, SW_TEXT));
                const uint32_t tl = ttab[SW_TEXT_LDS];
                lib.stext = tl != 0xFFFFFFFFu ? smem + tl : nullptr;
                verify_row<W>(lib, pol, row, t_rd, t_nm, t_L, need_before, need_after, best, best_seg, best_before);
              }
              if (best != ~0ull) {
                const uint64_t key = ((best >> 32) << 56) | ((best & 0xFFFFFFFFull) << 24) |
                                     ((uint64_t)(best_seg & 0xFFFFu) << 8) | (uint64_t)(best_before & 0xFFu);
                atomicMin(&my_slot_keys[2u * t], (unsigned long long)key);
              }
            }
          }
          lo += w;
          rem -= w;
        }
        __builtin_amdgcn_wave_barrier();

        // ================= phase C: owners replay their items of this batch in cascade order =================
        // my items are [excl, excl + cnt); those inside [base, base + 64) sit in slots item - base
        {
          const uint32_t first = max(excl, base), last = min(excl + cnt, base + 64u);
          uint32_t bits = need;
          for (uint32_t jj = excl; jj < first && jj < excl + cnt; ++jj) bits &= bits - 1u;  // replayed in earlier batches
          for (uint32_t it = first; it < last; ++it) {
            const uint32_t cb = (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            const uint32_t iq = pair_round ? 0u : cb >> 1, ik = pair_round ? cb : cb & 1u;
            const uint4 sv = my_slots[it - base];
            if (iq != q_cur) {
              // the previous sub-pass is complete: did it claim the read?
              if (win_q == kNoQ && q_best != ~0ull) {
                win_key = q_best;
                win_q = q_cur;
              }
              q_cur = iq;
              q_best = ~0ull;
            }
            if (win_q != kNoQ) continue;                       // claimed before this sub-pass: never issued
            if (ik >= 1u && (q_best >> 56) == 0ull) continue;  // piece / pair 0 was exact: the others never issued
            const uint64_t key = (uint64_t)sv.x | ((uint64_t)sv.y << 32);
            q_best = min(q_best, key);
            unsigned long long* c64 =
                cnt64 + (size_t)(s0 + iq) * 3u * kFusedCntReplicas + (lane & (kFusedCntReplicas - 1u));
            if (sv.w & 0x7FFFFFFFu) atomicAdd(c64, (unsigned long long)(sv.w & 0x7FFFFFFFu));
            if (sv.z) atomicAdd(c64 + kFusedCntReplicas, (unsigned long long)sv.z);
            if (sv.w & 0x80000000u) atomicAdd(c64 + 2 * kFusedCntReplicas, 1ull);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (win_q == kNoQ && q_best != ~0ull) {
        win_key = q_best;
        win_q = q_cur;
      }

      // ================= phase D: the 0/1 counters of the round and the outputs =================
      {
        const uint32_t offered = (elig & (win_q == kNoQ ? 0xFFu : ((2u << win_q) - 1u))) << s0;
        const uint32_t aligned = (win_q == kNoQ ? 0u : (1u << win_q)) << s0;
        acc_offered[0] += spread4(offered & 15u);
        acc_offered[1] += spread4(offered >> 4);
        acc_aligned[0] += spread4(aligned & 15u);
        acc_aligned[1] += spread4(aligned >> 4);
      }
      if (win_q != kNoQ) {
        const uint32_t* tab = stab + (s0 + win_q) * kSubWords;
        const uint32_t st = (uint32_t)(win_key >> 24);
        uint32_t sg = (uint32_t)(win_key >> 8) & 0xFFFFu;
        const uint32_t before = (uint32_t)win_key & 0xFFu;
        const uint32_t* seg_start = reinterpret_cast<const uint32_t*>(lds_ptr(tab, SW_SEGSTART));
        if (sg == 0xFFFFu) {
          sg = reinterpret_cast<const uint32_t*>(lds_ptr(tab, SW_CHUNKSEG))[st >> 5];
          while (seg_start[sg + 1] <= st) ++sg;
        }
        const uint32_t simple = tab[SW_SIMPLE];
        uint32_t ref = sg, pos;
        if (simple && before < 255u) {
          pos = before;
        } else {
          uint32_t off = 0;
          if (!simple) {
            ref = reinterpret_cast<const uint32_t*>(lds_ptr(tab, SW_SEGREF))[sg];
            off = reinterpret_cast<const uint32_t*>(lds_ptr(tab, SW_SEGOFF))[sg];
          }
          pos = st - seg_start[sg] + off;
        }
        if (p.packed) {
          p.packed[r] = pack_assignment((int32_t)tab[SW_PASSIDX], ref, pos, (uint32_t)(win_key >> 56));
        } else {
          p.pass_id[r] = (int8_t)tab[SW_PASSIDX];
          p.ref_id[r] = (int32_t)ref;
          p.pos[r] = (int32_t)pos;
          p.mm[r] = (uint8_t)(win_key >> 56);
        }
        claimed = true;
      }
    }

    if (active && !claimed && !p.idx_out) {
      // the group ends the cascade: whatever is still unclaimed stays unannotated
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
      const bool survive = active && !claimed;
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
  // ---- counters: the per-lane 16-bit fields summed over the wave, then over the workgroup in
  // LDS (the result slots are free now), then one global atomic per counter and workgroup
  __syncthreads();
  unsigned long long* wg_cnt = reinterpret_cast<unsigned long long*>(slots);  // [n_sub][2]
  if (threadIdx.x < 2u * kMaxFused) wg_cnt[threadIdx.x] = 0ull;
  __syncthreads();
#pragma unroll 1
  for (uint32_t s = 0; s < p.n_sub; ++s) {
    const uint64_t of = (acc_offered[s >> 2] >> (16u * (s & 3u))) & 0xFFFFull;
    const uint64_t al = (acc_aligned[s >> 2] >> (16u * (s & 3u))) & 0xFFFFull;
    const uint64_t t_of = wave_sum(of), t_al = wave_sum(al);
    if (lane == 0) {
      if (t_of) atomicAdd(&wg_cnt[2 * s], (unsigned long long)t_of);
      if (t_al) atomicAdd(&wg_cnt[2 * s + 1], (unsigned long long)t_al);
    }
  }
  __syncthreads();
  if (threadIdx.x < 2u * p.n_sub && wg_cnt[threadIdx.x])
    atomicAdd((unsigned long long*)&p.sub[threadIdx.x >> 1].counters[threadIdx.x & 1u], wg_cnt[threadIdx.x]);
  if (p.idx_out && threadIdx.x == 0) p.out_count[blockIdx.x] = ctl[0];
  for (uint32_t i = threadIdx.x; i < p.n_sub * 3u; i += BLOCK) {
    const uint32_t s = i / 3u, c = i % 3u;
    unsigned long long v = 0ull;
    const unsigned long long* src = cnt64 + ((size_t)s * 3u + c) * kFusedCntReplicas;
    for (uint32_t q = 0; q < kFusedCntReplicas; ++q) v += src[q];
    if (v) atomicAdd((unsigned long long*)&p.sub[s].counters[2u + c], v);
  }
}

// ---------------------------------------------------------------------------
// stratum_kernel: the strata [k_first, k_last] of a stratum-first (2-mismatch) pass with the
// suffix-array rows compacted over the wave.
//
// The last stratum of `-v 2` on a small library is candidate verification and nothing else: three
// 6..7-base pieces per read, ~20 rows each.  With one read per lane (match_kernel) every row load
// is a 64-address gather -- the launch was bound by the vector L1's tag rate (7.9e8 accesses for
// 4.4e8 rows), not by L2 or the ALUs.  Here a wave still holds 64 reads and every lane does its own
// jump-table load and LF steps, but the rows of the 64 intervals of one piece are dealt out as in
// fused_kernel: lane i takes row i of the batch, whoever's interval it belongs to (consecutive
// lanes read consecutive rows: a few cache lines per wave instruction), fetches the owner's read
// with shuffles, verifies, and folds a valid alignment into the owner's 8-byte LDS key with a
// 64-bit atomic min.  Pieces are swept one after the other, so the sequential search's early exits
// (an exact hit of an earlier piece, a stratum complete below its bound) are taken by the owner
// between sweeps and the processed / aligned / steps / candidates / lookups counters are those of
// match_kernel<STRATA> for the same strata.
// ---------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(1024, (W == 1 ? 8 : 4)) stratum_kernel(const MatchParams p, const uint32_t lds_text) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr uint32_t BLOCK = 1024u;
  // ---- LDS carve: [superblocks][text][9-mer bitmap][control][one 8-byte key per lane][counters] ----
  const uint32_t sup_words = p.nsup * 4;
  const uint32_t txt_words = lds_text ? p.text_words : 0u;
  uint32_t* ssuper = smem;
  uint32_t* stext = ssuper + sup_words;
  uint32_t* skbits = stext + txt_words;
  uint32_t* ctl = skbits + (p.kbits ? kKmerBitsWords : 0u);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ctl + 4);
  unsigned long long* wg_cnt = keys + BLOCK;
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.super);
    uint4* dst = reinterpret_cast<uint4*>(ssuper);
    for (uint32_t i = threadIdx.x; i < sup_words / 4; i += BLOCK) dst[i] = src[i];
  }
  if (lds_text) {
    const uint4* src = reinterpret_cast<const uint4*>(p.text);
    uint4* dst = reinterpret_cast<uint4*>(stext);
    for (uint32_t i = threadIdx.x; i < txt_words / 4; i += BLOCK) dst[i] = src[i];
  }
  if (p.kbits) {
    const uint4* src = reinterpret_cast<const uint4*>(p.kbits);
    uint4* dst = reinterpret_cast<uint4*>(skbits);
    for (uint32_t i = threadIdx.x; i < kKmerBitsWords / 4; i += BLOCK) dst[i] = src[i];
  }
  if (threadIdx.x == 0) {
    ctl[0] = 0u;
    ctl[1] = 0u;
  }
  if (threadIdx.x < 5) wg_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  if (p.idx_in) {
    uint32_t mx = 0;
    for (uint32_t sgi = threadIdx.x; sgi < p.in_nseg; sgi += BLOCK) mx = max(mx, p.in_count[sgi]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, off, 64));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&ctl[1], mx);
  }
  __syncthreads();

  Lib<false, false> fm;  // LF steps: occ blocks from L2, superblock table from LDS
  fm.gblocks = p.blocks;
  fm.gtext = nullptr;
  fm.sblocks = nullptr;
  fm.stext = nullptr;
  fm.ssuper = ssuper;
  fm.primary = p.primary;
  ItemLib lib;
  lib.gtext = p.text;
  lib.stext = lds_text ? stext : nullptr;

  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long* my_keys = keys + wave * 64u;
  const bool has_nm = p.nmask != nullptr;
  uint32_t c_processed = 0, c_aligned = 0, c_steps = 0, c_cands = 0, c_lookups = 0;

  const uint32_t in_nseg = p.idx_in ? p.in_nseg : 1u;
  const uint32_t depth_chunks = p.idx_in ? (ctl[1] + BLOCK - 1) / BLOCK : (p.n_total + BLOCK - 1) / BLOCK;
  const uint32_t n_chunks = in_nseg * depth_chunks;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t sgi = chunk % in_nseg, depth = chunk / in_nseg;
    const uint32_t t = depth * BLOCK + threadIdx.x;
    const bool active = t < (p.idx_in ? p.in_count[sgi] : p.n_total);
    uint32_t r = 0;
    uint64_t rd[W], nm[W];
    uint32_t L0 = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) rd[k] = nm[k] = 0ull;
    if (active) {
      r = p.idx_in ? p.idx_in[(size_t)sgi * p.in_seg_cap + t] : t;
      L0 = p.uniform_len ? p.uniform_len : (uint32_t)p.lens[r];
    }
    // ---- which reads this pass's FASTA would contain (RAP:543-554, 664-686) ----
    bool eligible = active && (int32_t)L0 >= p.min_len && (int32_t)L0 <= p.max_len;
    if (eligible) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        rd[k] = p.reads[(size_t)k * p.n_total + r];
        nm[k] = has_nm ? p.nmask[(size_t)k * p.n_total + r] : 0ull;
      }
    }
    int32_t L = (int32_t)L0;
    if (p.poly_t) {
      const int32_t tail = trailing_t<W>(rd, nm, L);
      eligible = eligible && tail >= 3 && (L - tail) >= 11;
      L -= tail;
    }
    L -= p.trim5 + p.trim3;
    shift_out_5p<W>(rd, (uint32_t)p.trim5);
    if (has_nm) shift_out_5p<W>(nm, (uint32_t)p.trim5);
    if (eligible && p.count_processed) ++c_processed;

    // key of the best alignment so far: mm:8 | text position:32 | segment:16 | before:8
    uint64_t best = ~0ull;
    bool searching = eligible && L > p.max_mm_seed;
    const int32_t R = min(L, p.seed_len);

    // ---------- rows: the rows [lo, lo + rem) of `rows` of every lane, one per lane and trip;
    // a valid alignment goes to its owner's `best` ----------
    auto sweep = [&](const uint64_t* __restrict__ rows, uint32_t lo, uint32_t rem, uint32_t nb) {
      const bool had_rows = rem != 0u;
      if (!__ballot(had_rows)) return;
      const uint32_t nb_L = nb | ((uint32_t)L << 8);
      my_keys[lane] = ~0ull;
      __builtin_amdgcn_wave_barrier();
      while (__ballot(rem != 0u)) {
        const uint32_t w = min(rem, kRowSlice);
        const uint32_t rincl = wave_incl_scan(w, lane);
        const uint32_t rtotal = __shfl(rincl, 63, 64);
        for (uint32_t rb = 0; rb < rtotal; rb += 64u) {
          const uint32_t x = rb + lane;
          const bool has_row = x < rtotal;
          uint32_t o = owner_of(rincl, x);
          o = has_row ? o : lane;
          const uint32_t o_incl = __shfl(rincl, (int)o, 64), o_w = __shfl(w, (int)o, 64);
          const uint32_t o_lo = __shfl(lo, (int)o, 64);
          const uint32_t o_nbL = __shfl(nb_L, (int)o, 64);
          uint64_t o_rd[W], o_nm[W];
#pragma unroll
          for (int q = 0; q < W; ++q) {
            o_rd[q] = __shfl(rd[q], (int)o, 64);
            o_nm[q] = has_nm ? __shfl(nm[q], (int)o, 64) : 0ull;
          }
          if (has_row) {
            const uint32_t need_before = o_nbL & 0xFFu;
            const int32_t o_L = (int32_t)(o_nbL >> 8);
            const uint64_t row = rows[o_lo + (x - (o_incl - o_w))];
            uint64_t rbest = ~0ull;
            uint32_t rseg = 0xFFFFu, rbefore = 255u;
            verify_row<W>(lib, p, row, o_rd, o_nm, o_L, need_before, (uint32_t)o_L - need_before, rbest, rseg, rbefore);
            if (rbest != ~0ull) {
              const uint64_t key = ((rbest >> 32) << 56) | ((rbest & 0xFFFFFFFFull) << 24) |
                                   ((uint64_t)(rseg & 0xFFFFu) << 8) | (uint64_t)(rbefore & 0xFFu);
              atomicMin(&my_keys[o], (unsigned long long)key);
            }
          }
        }
        lo += w;
        rem -= w;
      }
      __builtin_amdgcn_wave_barrier();
      if (had_rows) best = min(best, (uint64_t)my_keys[lane]);
    };

    // ---------- reads that hold the four anchors: the six anchor pairs (fm_index.hpp) ----------
    // anchor length of my read: pair_anchor when the seed region holds four of them, one base less
    // (second table set) for the next shorter reads, 0 = pigeonhole pieces
    uint32_t A = 0u;
    if (searching && p.pair_anchor) {
      if (R >= (int32_t)(4u * p.pair_anchor)) A = p.pair_anchor;
      else if (p.pair_jump_s && R >= (int32_t)(4u * (p.pair_anchor - 1u))) A = p.pair_anchor - 1u;
    }
    const bool by_pairs = A != 0u;
    if (__ballot(by_pairs)) {
      const bool second = A != p.pair_anchor;
      const uint32_t kb = 2u * A, amask = (1u << kb) - 1u, n_codes1 = (1u << (2u * kb)) + 1u;
      const uint32_t* my_jump = second ? p.pair_jump_s : p.pair_jump;
      // Stratum first, as in the piece search: an exact alignment matches EVERY pair, so pair (0,1)
      // alone sees all of them; one mismatch leaves (0,1) or (2,3) clean, so after those two every
      // alignment with <= 1 mismatch has been seen.  A best hit below the bound is final and the
      // remaining pairs are not looked up.
      bool open_pairs = by_pairs;
#pragma unroll 1
      for (uint32_t pr = 0; pr < 6u; ++pr) {
        // (i, j) = (0,1) (2,3) (1,2) (0,2) (1,3) (0,3): table j - i - 1
        const uint32_t i = (0x010120u >> (4u * pr)) & 15u, j = (0x332231u >> (4u * pr)) & 15u;
        const uint32_t ai = (uint32_t)(rd[0] >> (i * kb)) & amask, aj = (uint32_t)(rd[0] >> (j * kb)) & amask;
        bool go = open_pairs;
        // an anchor holding an N is never the exact one
        if (has_nm && ((((uint32_t)(nm[0] >> (i * kb)) | (uint32_t)(nm[0] >> (j * kb))) & amask) != 0u)) go = false;
        uint32_t lo = 0, hi = 0;
        const uint32_t t = j - i - 1u;
        if (go) {
          const uint32_t* tab = my_jump + t * n_codes1 + (ai | (aj << kb));
          const uint32_t roff = second ? p.pair_row_off_s[t] : p.pair_row_off[t];
          lo = tab[0] + roff;
          hi = tab[1] + roff;
          ++c_lookups;
        }
        c_cands += hi - lo;
        sweep(p.pair_rows, lo, hi - lo, i * A);  // (row lanes index the one array both sets live in)
        if (pr < 2u && (uint32_t)(best >> 56) <= pr) open_pairs = false;
        if (!__ballot(open_pairs)) break;
      }
      if (by_pairs) searching = false;  // every alignment with <= 2 seed mismatches has been seen
    }

    // ---------- the others: stratum-first pigeonhole pieces ----------
    if (__ballot(searching))
    for (int32_t K = p.k_first; K <= p.k_last; ++K) {
      bool in_stratum = searching;
      for (int32_t k = 0; k < K; ++k) {
        const int32_t a = div_pieces(R * k, K), b = div_pieces(R * (k + 1), K);
        bool go = in_stratum;
        if (go && has_nm && piece_has_n<W>(nm, a, b)) go = false;  // a piece holding an N can never be the exact one
        if (go && p.kbits && b - a >= (int32_t)kKmerBitsK) {
          const uint32_t c9 = read_bits9<W>(rd, (uint32_t)b - kKmerBitsK);
          if (((skbits[c9 >> 5] >> (c9 & 31u)) & 1u) == 0u) go = false;
          if (go && b - a > (int32_t)kKmerBitsK) {
            const uint32_t c0 = read_bits9<W>(rd, (uint32_t)a);
            if (((skbits[c0 >> 5] >> (c0 & 31u)) & 1u) == 0u) go = false;
          }
        }
        // ---- exact backward search of read[a,b) ----
        uint32_t lo = 0, hi = 0;
        int32_t j = b;
        if (go) {
          hi = p.n + 1;
          uint32_t tab_off = 0;
          const uint32_t tk = p.tabs.k[0] ? pick_table(p.tabs, b - a, tab_off) : 0u;
          if (tk) {
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
            const uint4 vl = fm.block(lo >> 5);
            uint4 vh = vl;
            if ((hi >> 5) != (lo >> 5)) vh = fm.block(hi >> 5);
            lo = fm.lf(c, lo, vl);
            hi = fm.lf(c, hi, vh);
            ++c_steps;
          }
        }
        const uint32_t rem = hi > lo ? hi - lo : 0u;
        c_cands += rem;
        sweep(p.sa, lo, rem, (uint32_t)j);
        if ((best >> 56) == 0ull) in_stratum = false;  // an exact hit is always seen by piece 0
      }
      if ((uint32_t)(best >> 56) < (uint32_t)K) searching = false;  // complete below K mismatches (unaligned = 255)
    }

    // (a launch that stops before the last stratum keeps a hit only if it is final)
    const bool aligned = best != ~0ull && (by_pairs || p.k_last > p.max_mm_seed || (int32_t)(best >> 56) < p.k_last);
    if (aligned) {
      ++c_aligned;
      const uint32_t s = (uint32_t)(best >> 24);
      uint32_t sg = (uint32_t)(best >> 8) & 0xFFFFu;
      const uint32_t before = (uint32_t)best & 0xFFu;
      if (sg == 0xFFFFu) {  // more than 65535 segments: walk the chunk map
        sg = p.chunk_seg[s >> 5];
        while (p.seg_start[sg + 1] <= s) ++sg;
      }
      uint32_t ref = sg, pos;
      if (p.simple_segs && before < 255u) {
        pos = before;
      } else {
        uint32_t off = 0;
        if (!p.simple_segs) {
          ref = p.seg_ref[sg];
          off = p.seg_off[sg];
        }
        pos = s - p.seg_start[sg] + off;
      }
      if (p.packed) {
        p.packed[r] = pack_assignment(p.pass_index, ref, pos, (uint32_t)(best >> 56));
      } else {
        p.pass_id[r] = (int8_t)p.pass_index;
        p.ref_id[r] = (int32_t)ref;
        p.pos[r] = (int32_t)pos;
        p.mm[r] = (uint8_t)(best >> 56);
      }
    } else if (active && !p.idx_out) {
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
        if (survive)
          p.idx_out[(size_t)blockIdx.x * p.out_seg_cap + wbase +
                    (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = r;
      }
    }
  }
  const uint64_t t_processed = wave_sum(c_processed), t_aligned = wave_sum(c_aligned);
  const uint64_t t_steps = wave_sum(c_steps), t_cands = wave_sum(c_cands);
  const uint64_t t_lookups = wave_sum(c_lookups);
  if (lane == 0) {
    if (t_processed) atomicAdd(&wg_cnt[0], (unsigned long long)t_processed);
    if (t_aligned) atomicAdd(&wg_cnt[1], (unsigned long long)t_aligned);
    if (t_steps) atomicAdd(&wg_cnt[2], (unsigned long long)t_steps);
    if (t_cands) atomicAdd(&wg_cnt[3], (unsigned long long)t_cands);
    if (t_lookups) atomicAdd(&wg_cnt[4], (unsigned long long)t_lookups);
  }
  __syncthreads();
  if (threadIdx.x < 5 && wg_cnt[threadIdx.x])
    atomicAdd((unsigned long long*)&p.counters[threadIdx.x], wg_cnt[threadIdx.x]);
  if (p.idx_out && threadIdx.x == 0) p.out_count[blockIdx.x] = ctl[0];
}

constexpr uint32_t kCountTodoMark = 254u;  // best_mm of a read count_variants_kernel left to count_kernel
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
    if (!LIST && p.only_todo && p.best_mm[r] != (uint8_t)kCountTodoMark) continue;  // (count_variants_kernel answered this read)
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
// count_variants_kernel (round 6): the one-mismatch genome run (`-n 1 -a -3 2`, writeDataToCSV.py:1263) for one-word reads
// without N whose whole length is seed region (L <= seed_len), through the library's LARGEST jump table instead of
// the pigeonhole pieces.  count_kernel halves a 20-nt read into pieces of 10 bases; in a 300 Mbp part a 10-mer has
// ~290 rows, so a read cost ~570 row + text verifications, one after the other in its lane: 65 M reads/s
// (profiles/r01_genome_part_300m.json).  With K = the largest table's k <= L (14 for a genome part):
//   A = the read's first K bases, exact       -> every alignment whose mismatch (if any) lies behind base K;
//   B = its last K bases, exact               -> those with exactly one mismatch in front of base L - K;
//   one variant of A per (position p in [L - K, K), other base)  -> those with exactly one mismatch, at p
// -- disjoint classes that cover every alignment with at most one mismatch, 2 + 3 (2 K - L) table lookups of ~1 row each
// (26 for a 20-nt read).  Half a wave (32 lanes) per read, a lane per lookup: nothing is walked serially, and no LF step
// is made at all.  Reads this does not fit (shorter than a usable table, longer than the seed, two mismatches...) are
// marked kCountTodo in best_mm and left to count_kernel, which then skips every other read.
// ---------------------------------------------------------------------------
constexpr uint32_t kCountTodo = kCountTodoMark;

__global__ void __launch_bounds__(kCountThreads) count_variants_kernel(const CountParams p) {
  const uint32_t lane32 = threadIdx.x & 31u;
  const uint64_t groups = (uint64_t)gridDim.x * (kCountThreads / 32u);
  for (uint64_t r = (uint64_t)blockIdx.x * (kCountThreads / 32u) + (threadIdx.x >> 5); r < p.n_reads; r += groups) {
    const uint64_t rd = p.reads[r];
    const int32_t L = (int32_t)p.lens[r];
    // the largest table a read of L bases can use (tables in ascending k)
    uint32_t K = 0, tab_off = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool take = p.tabs.k[i] != 0u && (int32_t)p.tabs.k[i] <= L;
      K = take ? p.tabs.k[i] : K;
      tab_off = take ? p.tabs.off[i] : tab_off;
    }
    if (K < 8u || L > p.seed_len || L > 32 || L <= p.max_mm_seed) {  // (not this kernel's read)
      if (lane32 == 0u) p.best_mm[r] = (L <= p.max_mm_seed) ? 255u : (uint8_t)kCountTodo;
      if (lane32 == 0u && L <= p.max_mm_seed) p.count[r] = 0u;
      continue;
    }
    const uint32_t tail = (uint32_t)L - K;                    // B starts here; positions [tail, K) need the variants
    const uint32_t n_var = K > tail ? 3u * (K - tail) : 0u;
    const uint32_t n_items = 2u + n_var;
    const uint64_t lmask = low_bits(2u * (uint32_t)L);
    uint32_t best = 255u, cnt = 0u;
    bool sat = false;
    for (uint32_t it = lane32; it < n_items; it += 32u) {
      if (it == 1u && tail == 0u) continue;                  // (L == K: B is A)
      uint64_t q = rd;
      uint32_t off = 0u;                                     // read offset of the K-mer looked up
      if (it == 1u) {
        off = tail;
      } else if (it >= 2u) {
        const uint32_t v = it - 2u;
        q = rd ^ ((uint64_t)(v % 3u + 1u) << (2u * (tail + v / 3u)));
      }
      const uint32_t* tab = p.ftab + tab_off + lex_code((q >> (2u * off)) & low_bits(2u * K), K);
      const uint32_t lo = tab[0];
      uint32_t hi = tab[1];
      if (hi > lo && hi - lo > p.max_rows) {  // (count_kernel's rule: a seed this repetitive cannot have a unique best hit)
        sat = true;
        hi = lo + p.max_rows;
      }
      for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t row = p.sa[i];
        const uint32_t before = (uint32_t)(row >> 32) & 255u, after = (uint32_t)(row >> 40) & 255u;
        if ((off > before) | ((uint32_t)L - off > after)) continue;
        const uint32_t s = (uint32_t)row - off;
        const uint32_t w = s >> 4, sh = (s & 15u) * 2u;
        const uint64_t lo64 = (uint64_t)p.text[w] | ((uint64_t)p.text[w + 1] << 32);
        const uint64_t win = (lo64 >> sh) | ((((uint64_t)p.text[w + 2]) << 1) << (63u - sh));
        const uint64_t x = win ^ rd;
        const uint64_t mb = ((x | (x >> 1)) & 0x5555555555555555ull) & lmask;
        const uint32_t mm = (uint32_t)__popcll(mb);
        if ((int32_t)mm > p.max_mm_seed || (int32_t)mm > p.max_mm_total) continue;
        if (it != 0u && mm != 1u) continue;                  // B and the variants own the alignments with their one mismatch
        // (a read of more than 2 K bases: A and B do not meet, and a mismatch between them leaves BOTH exact -- A's)
        if (it == 1u && (uint32_t)(__ffsll((long long)mb) - 1) / 2u >= K) continue;
        if (mm < best) {
          best = mm;
          cnt = 1u;
        } else if (mm == best) {
          ++cnt;
        }
      }
    }
    // the 32 lanes' (best, count) -> the read's
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const uint32_t ob = __shfl_xor(best, o, 32), oc = __shfl_xor(cnt, o, 32);
      const bool os = __shfl_xor((int)sat, o, 32) != 0;
      cnt = ob < best ? oc : (ob == best ? cnt + oc : cnt);
      best = min(best, ob);
      sat |= os;
    }
    if (lane32 == 0u) {
      p.best_mm[r] = (uint8_t)best;
      const uint32_t c = best == 255u ? 0u : ((sat && !p.count32) ? 255u : cnt);
      if (p.count32) p.count32[r] = best == 255u ? 0u : cnt;
      else p.count[r] = (uint8_t)min(c, 255u);
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
  const uint32_t n_cat = (p.n_pass + 1) * S;
  const uint32_t n_bins = 2 * M * S + n_cat + S;
  const uint32_t cat0 = 2 * M * S, uniq0 = cat0 + n_cat;
  // LDS layout: [miRNA bins][category bins x kTallyCatReplicas][trimmedUniq].  The ~10 category bins
  // are hit by EVERY lane: one copy per (lane & 31) turns a 64-way same-address LDS atomic into
  // a 2-way one.
  constexpr uint32_t R = kTallyCatReplicas;
  const uint32_t l_cat0 = cat0, l_uniq0 = cat0 + n_cat * R, l_bins = l_uniq0 + S;
  unsigned long long* g = reinterpret_cast<unsigned long long*>(p.counts);
  if (p.export_stats && blockIdx.x == 0 && threadIdx.x < 2u * p.export_n_pass)
    p.export_out[threadIdx.x] = p.export_stats[(threadIdx.x >> 1) * 5u + (threadIdx.x & 1u)];
  if (LDSH) {
    for (uint32_t i = threadIdx.x; i < l_bins; i += kTallyThreads) hist[i] = 0ull;
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t rep = LDSH ? (lane & (R - 1u)) : 0u;

  // one read's contribution (whole waves call it: the ballot counts the reads with a non-zero count)
  auto add = [&](bool active, int32_t pass, uint32_t ref_raw, unsigned long long q, uint32_t s) {
    // an unclaimed read (pass -1) must not match a disabled (-1) canon/isomiR pass
    const bool canon = active && pass >= 0 && pass == p.canon_pass;
    const bool iso = active && pass >= 0 && pass == p.isomir_pass;
    const uint32_t ref = (canon || iso) ? ref_raw : 0u;
    const uint32_t cat = pass < 0 ? p.n_pass : (uint32_t)pass;
    const bool hit = active && q != 0ull;
    const uint64_t hits = __ballot(hit);
    if (!hits) return;
    if (LDSH) {
      if (lane == 0) atomicAdd(&hist[l_uniq0 + s], (unsigned long long)__popcll(hits));
      if (hit) atomicAdd(&hist[l_cat0 + (cat * S + s) * R + rep], q);
      // one LDS atomic per read: a canonical read goes to the iscan bin only, quant = iscan + isomiR
      // reads is formed at the flush
      if (hit && iso && !canon) atomicAdd(&hist[ref * S + s], q);
      if (hit && canon) atomicAdd(&hist[M * S + ref * S + s], q);
    } else {
      if (lane == 0) atomicAdd(&g[uniq0 + s], (unsigned long long)__popcll(hits));
      if (hit) atomicAdd(&g[cat0 + cat * S + s], q);
      if (hit && (canon || iso)) atomicAdd(&g[ref * S + s], q);
      if (hit && canon) atomicAdd(&g[M * S + ref * S + s], q);
    }
  };
  uint64_t r_first = 0;  // reads before it were taken four per lane
  if (p.vec4) {
    // one sample, 16-byte aligned arrays: a lane takes four consecutive reads per trip -- a 4-byte,
    // and two 16-byte loads instead of 1 + 4 + 4 bytes per lane (64-byte wave requests stream badly)
    const uint64_t n4 = p.n >> 2;
    const uint64_t n4_round = ((n4 + kTallyThreads - 1) / kTallyThreads) * kTallyThreads;
    const uint32_t* pass4 = reinterpret_cast<const uint32_t*>(p.pass_id);
    const uint4* ref4 = reinterpret_cast<const uint4*>(p.ref_id);
    const uint4* packed4 = reinterpret_cast<const uint4*>(p.packed);
    const uint4* quant4 = reinterpret_cast<const uint4*>(p.quant);
    for (uint64_t g4 = (uint64_t)blockIdx.x * kTallyThreads + threadIdx.x; g4 < n4_round;
         g4 += (uint64_t)gridDim.x * kTallyThreads) {
      const bool active = g4 < n4;
      uint32_t pw = 0xFFFFFFFFu;
      uint4 rf = make_uint4(0u, 0u, 0u, 0u), qv = make_uint4(0u, 0u, 0u, 0u);
      if (active) {
        if (p.packed) {
          // one 16-byte load of packed words instead of 4 + 16 bytes of pass ids and entries
          const uint4 k = packed4[g4];
          pw = (((k.x >> 28) - 1u) & 0xFFu) | ((((k.y >> 28) - 1u) & 0xFFu) << 8) | ((((k.z >> 28) - 1u) & 0xFFu) << 16) |
               ((((k.w >> 28) - 1u) & 0xFFu) << 24);
          rf = make_uint4((k.x >> 8) & 0x3FFFFu, (k.y >> 8) & 0x3FFFFu, (k.z >> 8) & 0x3FFFFu, (k.w >> 8) & 0x3FFFFu);
        } else {
          pw = pass4[g4];
          rf = ref4[g4];
        }
        qv = quant4[g4];
      }
      add(active, (int32_t)(int8_t)(pw & 0xFFu), rf.x, qv.x, 0u);
      add(active, (int32_t)(int8_t)((pw >> 8) & 0xFFu), rf.y, qv.y, 0u);
      add(active, (int32_t)(int8_t)((pw >> 16) & 0xFFu), rf.z, qv.z, 0u);
      add(active, (int32_t)(int8_t)(pw >> 24), rf.w, qv.w, 0u);
    }
    r_first = n4 << 2;
  }
  const uint64_t n_rest = p.n - r_first;
  const uint64_t n_round = ((n_rest + kTallyThreads - 1) / kTallyThreads) * kTallyThreads;
  for (uint64_t k = (uint64_t)blockIdx.x * kTallyThreads + threadIdx.x; k < n_round;
       k += (uint64_t)gridDim.x * kTallyThreads) {
    const uint64_t r = r_first + k;
    const bool active = k < n_rest;  // whole waves stay in the loop: ballots
    const uint32_t pk = (active && p.packed) ? p.packed[r] : 0u;
    const int32_t pass = !active ? -1 : (p.packed ? (int32_t)(pk >> 28) - 1 : (int32_t)p.pass_id[r]);
    const bool wants_ref = active && pass >= 0 && (pass == p.canon_pass || pass == p.isomir_pass);
    const uint32_t ref = wants_ref ? (p.packed ? (pk >> 8) & 0x3FFFFu : (uint32_t)p.ref_id[r]) : 0u;
    for (uint32_t s = 0; s < S; ++s) add(active, pass, ref, active ? p.quant[r * S + s] : 0ull, s);
  }
  if (LDSH) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_bins; i += kTallyThreads) {
      unsigned long long v = 0ull;
      if (i < cat0) {
        v = hist[i];
        if (i < M * S) v += hist[M * S + i];
      } else if (i < uniq0) {
        const unsigned long long* src = hist + l_cat0 + (size_t)(i - cat0) * R;
        for (uint32_t q = 0; q < R; ++q) v += src[q];
      } else {
        v = hist[l_uniq0 + (i - uniq0)];
      }
      if (v) atomicAdd(&g[i], v);
    }
  }
}

// ---------------------------------------------------------------------------
// edit_tally_kernel: the per-read part of the A-to-I report (writeDataToCSV.py:145-229,
// A2IEditing + judgeAllign :35-69) on the alignments the cascade already produced.
// For a read claimed by the exact-miRNA or the isomiR pass, entry e, entry offset o of its first
// base (pos, minus the -5 trim of the isomiR pass): the mature sequence is entry[flank5 :
// len - flank3] and read base j sits at mature index d + j, d = o - flank5.  judgeAllign keeps
// the read when it starts at most 1 nt after the mature start (d <= 1) and, over the mature
// minus its last 3 nt, shows at most 1 mismatch and at least len - 4 (d == 1: len - 5) matches.
// Kept reads add their per-sample count to count_true (and 1 to seq_true, and the count to
// canonical when the whole read is a substring of the mature sequence) and, for every mature
// position i < len - 5 where the mature base is `from_base` and the read shows `to_base`, to the
// position bin (e, i).  The three per-entry totals are privatised in LDS (every kept read hits
// them); the position bins are rare events and go straight to L2 atomics.
// The reference aligns read and mature with pairwise2.localms (gap penalties of -20: an ungapped
// diagonal); the cascade's alignment is that diagonal for every read it claimed.
// ---------------------------------------------------------------------------
// LDSH: the three per-entry totals are privatised in LDS (count_true and canonical 64-bit,
// seq_true 32-bit: a workgroup sees < 2^32 reads); LDSL: the library's packed text and entry
// starts are staged in LDS too (a miRNA library is ~30 KB), so a read costs no gather at all.
template <bool LDSH, bool LDSL>
__global__ void __launch_bounds__(kEditThreads) edit_tally_kernel(const EditParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t S = p.n_samples;
  const uint32_t n_tot = p.n_bins * S;
  // LDS carve: [count_true u64 x n_tot][canonical u64 x n_tot][seq_true u32 x n_tot (padded)][text][seg_start]
  unsigned long long* h_ct = reinterpret_cast<unsigned long long*>(smem);
  unsigned long long* h_cn = h_ct + (LDSH ? n_tot : 0u);
  uint32_t* h_st = reinterpret_cast<uint32_t*>(h_cn + (LDSH ? n_tot : 0u));
  uint32_t* s_text = h_st + (LDSH ? ((n_tot + 3u) & ~3u) : 0u);
  uint32_t* s_seg = s_text + (LDSL ? p.text_words : 0u);
  // position bins: a small LDS hash (key = index into the global position array) absorbs the hot
  // (entry, position) pairs -- one abundant miRNA with one edited site would otherwise be tens of
  // thousands of L2 atomics on one cache line, ~10 ns each, serialised
  uint32_t* hk = s_seg + (LDSL ? ((p.n_entries + 4u) & ~3u) : 0u);
  unsigned long long* hv = reinterpret_cast<unsigned long long*>(hk + kEditHashSlots);
  unsigned long long* g = reinterpret_cast<unsigned long long*>(p.counts);
  unsigned long long* gpos = g + (size_t)n_tot * 3u;
  for (uint32_t i = threadIdx.x; i < kEditHashSlots; i += kEditThreads) {
    hk[i] = 0xFFFFFFFFu;
    hv[i] = 0ull;
  }
  if (LDSH)
    for (uint32_t i = threadIdx.x; i < n_tot; i += kEditThreads) {
      h_ct[i] = 0ull;
      h_cn[i] = 0ull;
      h_st[i] = 0u;
    }
  if (LDSL) {
    for (uint32_t i = threadIdx.x; i < p.text_words; i += kEditThreads) s_text[i] = p.text[i];
    for (uint32_t i = threadIdx.x; i <= p.n_entries; i += kEditThreads) s_seg[i] = p.seg_start[i];
  }
  __syncthreads();
  const uint32_t* text = LDSL ? s_text : p.text;
  const uint32_t* seg_start = LDSL ? s_seg : p.seg_start;
  // one read: the fields that every read has are passed in (requested before any of them is looked at)
  auto one = [&](uint64_t r, int32_t pass, uint32_t e, int32_t L, int32_t pos_r, uint64_t r0, bool have_q,
                 unsigned long long q_pre) {
    uint64_t hits = 0ull;
    uint32_t bin = 0;
    bool kept = false, canonical = false;
    do {
      if (r >= p.n) break;
      if (pass < 0 || (pass != p.canon_pass && pass != p.isomir_pass)) break;
      if (p.keep && !p.keep[r]) break;
      const uint32_t e0 = seg_start[e], e1 = seg_start[e + 1];
      const int32_t Lm = (int32_t)(e1 - e0) - (int32_t)(p.flank5 + p.flank3);
      if (Lm <= 0 || Lm > (int32_t)kEditPositions) break;  // (the host rejects such libraries)
      const int32_t d = pos_r - (pass == p.isomir_pass ? p.isomir_trim5 : 0) - (int32_t)p.flank5;
      if (d > 1) break;        // head shift (judgeAllign)
      if (d + L <= 0) break;   // the read ends before the mature sequence starts
      // 32 mature bases from the text, 2 bits each, mature index 0 in the low bits
      uint64_t tw;
      {
        const uint32_t q = e0 + p.flank5, i = q >> 4, sh = (q & 15u) * 2u;
        const uint64_t lo64 = (uint64_t)text[i] | ((uint64_t)text[i + 1] << 32);
        tw = (lo64 >> sh) | ((((uint64_t)text[i + 2]) << 1) << (63 - sh));
      }
      // the read in mature coordinates (base j -> index d + j), and its N mask
      const uint64_t r1 = p.words_per_read > 1 ? p.reads[p.n + r] : 0ull;
      const uint64_t m0 = p.nmask ? p.nmask[r] : 0ull, m1 = (p.nmask && p.words_per_read > 1) ? p.nmask[p.n + r] : 0ull;
      uint64_t rw, nw;
      if (d >= 0) {
        rw = d ? (r0 << (2 * d)) : r0;
        nw = d ? (m0 << (2 * d)) : m0;
      } else {
        const int32_t s2 = -2 * d;
        if (s2 >= 64) {
          const int32_t s3 = s2 - 64;
          rw = s3 < 64 ? (r1 >> s3) : 0ull;
          nw = s3 < 64 ? (m1 >> s3) : 0ull;
        } else {
          rw = (r0 >> s2) | (r1 << (64 - s2));
          nw = (m0 >> s2) | (m1 << (64 - s2));
        }
      }
      const int32_t c_lo = max(0, d), c_hi = min(Lm, d + L);  // mature indices the read covers
      if (c_hi <= c_lo) break;
      const uint64_t cover = low_bits(2 * c_hi) & ~low_bits(2 * c_lo) & kOdd;
      const uint64_t x = tw ^ rw;
      const uint64_t diff = (((x | (x >> 1)) & kOdd) | (nw & kOdd)) & cover;
      // judgeAllign's window, in its own arithmetic (W2C:35-69): with both sequences padded to the
      // common frame (head_t / head_s leading dashes, frame length plen) it compares frame positions
      // head_t .. min(end1, end2), end1 = plen - head_t - 1 - 3, end2 = last read base -- so a read
      // that starts before the mature sequence is judged over a shorter stretch, and one that runs
      // past its end is judged up to 3 bases before the READ's end, every base beyond the mature
      // sequence counting as a mismatch.  hi_m = that last position in mature coordinates.
      const int32_t head_t = max(0, -d), head_s = max(0, d);
      const int32_t plen = max(head_t + Lm, head_s + L);
      const int32_t hi_m = min(plen - head_t - 4, head_s + L - 1) - head_t;
      const uint64_t judged = cover & low_bits(2 * max(min(hi_m + 1, Lm), 0));
      const int32_t beyond = max(0, hi_m - max(Lm, c_lo) + 1);  // judged positions past the mature end
      const int32_t mism = __popcll(diff & judged) + beyond, mat = __popcll(~diff & judged);
      const int32_t need = (Lm - 4) - (d == 1 ? 1 : 0);
      if (mism > 1 || mat < need) break;
      kept = true;
      // canonical: the whole read is a substring of the mature sequence, at whatever offset (:170 `in`)
      if (L <= Lm && (m0 | m1) == 0ull) {
        for (int32_t o = 0; o + L <= Lm; ++o) {
          const uint64_t y = ((tw >> (2 * o)) ^ r0) & low_bits(2 * L);
          canonical |= y == 0ull;
        }
      }
      // positions i < Lm - 5 with mature == from_base and read == to_base (an N is never to_base)
      const uint64_t f = p.from_base, t = p.to_base;
      const uint64_t is_from = ~((tw ^ (f * kOdd)) | ((tw ^ (f * kOdd)) >> 1)) & kOdd;
      const uint64_t is_to = ~((rw ^ (t * kOdd)) | ((rw ^ (t * kOdd)) >> 1)) & kOdd & ~(nw & kOdd);
      hits = is_from & is_to & cover & low_bits(2 * max(Lm - 5, 0));
      bin = p.remap ? p.remap[e] : e;
    } while (false);
    for (uint32_t s = 0; s < S; ++s) {
      const unsigned long long q = kept ? (have_q ? q_pre : (unsigned long long)p.quant[r * S + s]) : 0ull;
      if (q) {
        const size_t k = (size_t)bin * S + s;
        if (LDSH) {
          atomicAdd(&h_ct[k], q);
          atomicAdd(&h_st[k], 1u);
          if (canonical) atomicAdd(&h_cn[k], q);
        } else {
          atomicAdd(&g[k * 3u + 0u], q);
          atomicAdd(&g[k * 3u + 1u], 1ull);
          if (canonical) atomicAdd(&g[k * 3u + 2u], q);
        }
      }
      for (uint64_t hb = q ? hits : 0ull; hb; hb &= hb - 1ull) {
        const uint32_t key = (bin * kEditPositions + ((uint32_t)(__ffsll((long long)hb) - 1) >> 1)) * S + s;
        uint32_t slot = (key * 2654435761u) >> (32u - kEditHashLog2);
        bool done = false;
        for (uint32_t t = 0; t < 4u && !done; ++t, slot = (slot + 1u) & (kEditHashSlots - 1u)) {
          const uint32_t prev = atomicCAS(&hk[slot], 0xFFFFFFFFu, key);
          if (prev == 0xFFFFFFFFu || prev == key) {
            atomicAdd(&hv[slot], q);
            done = true;
          }
        }
        if (!done) atomicAdd(&gpos[key], q);   // four occupied slots: a rare pair goes straight to L2
      }
    }
  };
  uint64_t r_first = 0;
  if (p.vec4) {
    // one sample, one-word reads, 16-byte aligned arrays: four consecutive reads per lane and trip
    // (4-byte loads of the pass ids and lengths, 16-byte loads of everything else)
    const uint64_t n4 = p.n >> 2;
    for (uint64_t g4 = (uint64_t)blockIdx.x * kEditThreads + threadIdx.x; g4 < n4; g4 += (uint64_t)gridDim.x * kEditThreads) {
      uint32_t pw;
      uint4 ev, pv;
      if (p.packed) {
        const uint4 k = reinterpret_cast<const uint4*>(p.packed)[g4];
        pw = (((k.x >> 28) - 1u) & 0xFFu) | ((((k.y >> 28) - 1u) & 0xFFu) << 8) | ((((k.z >> 28) - 1u) & 0xFFu) << 16) |
             ((((k.w >> 28) - 1u) & 0xFFu) << 24);
        ev = make_uint4((k.x >> 8) & 0x3FFFFu, (k.y >> 8) & 0x3FFFFu, (k.z >> 8) & 0x3FFFFu, (k.w >> 8) & 0x3FFFFu);
        pv = make_uint4(k.x & 0xFFu, k.y & 0xFFu, k.z & 0xFFu, k.w & 0xFFu);
      } else {
        pw = reinterpret_cast<const uint32_t*>(p.pass_id)[g4];
        ev = reinterpret_cast<const uint4*>(p.ref_id)[g4];
        pv = reinterpret_cast<const uint4*>(p.pos)[g4];
      }
      const uint32_t lw = reinterpret_cast<const uint32_t*>(p.lens)[g4];
      const uint4 qv = reinterpret_cast<const uint4*>(p.quant)[g4];
      const uint4 ra = reinterpret_cast<const uint4*>(p.reads)[2 * g4], rb = reinterpret_cast<const uint4*>(p.reads)[2 * g4 + 1];
      const uint64_t r = g4 << 2;
      one(r, (int32_t)(int8_t)(pw & 0xFFu), ev.x, (int32_t)(lw & 0xFFu), (int32_t)pv.x, (uint64_t)ra.x | ((uint64_t)ra.y << 32), true, qv.x);
      one(r + 1, (int32_t)(int8_t)((pw >> 8) & 0xFFu), ev.y, (int32_t)((lw >> 8) & 0xFFu), (int32_t)pv.y,
          (uint64_t)ra.z | ((uint64_t)ra.w << 32), true, qv.y);
      one(r + 2, (int32_t)(int8_t)((pw >> 16) & 0xFFu), ev.z, (int32_t)((lw >> 16) & 0xFFu), (int32_t)pv.z,
          (uint64_t)rb.x | ((uint64_t)rb.y << 32), true, qv.z);
      one(r + 3, (int32_t)(int8_t)(pw >> 24), ev.w, (int32_t)(lw >> 24), (int32_t)pv.w, (uint64_t)rb.z | ((uint64_t)rb.w << 32), true,
          qv.w);
    }
    r_first = n4 << 2;
  }
  for (uint64_t r = r_first + (uint64_t)blockIdx.x * kEditThreads + threadIdx.x; r < p.n; r += (uint64_t)gridDim.x * kEditThreads) {
    if (p.packed) {
      const uint32_t k = p.packed[r];
      one(r, (int32_t)(k >> 28) - 1, (k >> 8) & 0x3FFFFu, (int32_t)p.lens[r], (int32_t)(k & 0xFFu), p.reads[r], false, 0ull);
    } else {
      one(r, (int32_t)p.pass_id[r], (uint32_t)p.ref_id[r], (int32_t)p.lens[r], p.pos[r], p.reads[r], false, 0ull);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kEditHashSlots; i += kEditThreads)
    if (hk[i] != 0xFFFFFFFFu && hv[i]) atomicAdd(&gpos[hk[i]], hv[i]);
  if (LDSH) {
    for (uint32_t i = threadIdx.x; i < n_tot; i += kEditThreads) {
      if (h_ct[i]) atomicAdd(&g[(size_t)i * 3u + 0u], h_ct[i]);
      if (h_st[i]) atomicAdd(&g[(size_t)i * 3u + 1u], (unsigned long long)h_st[i]);
      if (h_cn[i]) atomicAdd(&g[(size_t)i * 3u + 2u], h_cn[i]);
    }
  }
}

template <bool LDSH, bool LDSL>
static hipError_t launch_edit_k(const EditParams& p, uint32_t grid, uint32_t lds_bytes, hipStream_t stream) {
  auto kern = edit_tally_kernel<LDSH, LDSL>;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kEditThreads), lds_bytes, stream, p);
  return hipGetLastError();
}

hipError_t launch_edit_tally(const EditParams& p, bool lds_hist, bool lds_lib, uint32_t grid, uint32_t lds_bytes,
                             hipStream_t stream) {
  if (lds_hist) return lds_lib ? launch_edit_k<true, true>(p, grid, lds_bytes, stream)
                               : launch_edit_k<true, false>(p, grid, lds_bytes, stream);
  return lds_lib ? launch_edit_k<false, true>(p, grid, lds_bytes, stream)
                 : launch_edit_k<false, false>(p, grid, lds_bytes, stream);
}

__global__ void export_pass_counts_kernel(const uint64_t* stats, uint32_t n_pass,
                                          uint64_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pass) {
    out[2 * i] = stats[5 * i];
    out[2 * i + 1] = stats[5 * i + 1];
  }
}

// One-word N-free reads to one list, the rest to the other (SplitParams).  A wave appends its
// members with one LDS atomic per list; the order inside a segment is whatever the waves' turns
// make it -- no consumer depends on it.
__global__ __launch_bounds__(1024) void split_kernel(SplitParams p) {
  __shared__ uint32_t fill[2];
  if (threadIdx.x < 2) fill[threadIdx.x] = 0u;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * p.seg_cap;
  for (uint64_t c0 = (uint64_t)blockIdx.x * 1024u; c0 < p.n_total; c0 += (uint64_t)gridDim.x * 1024u) {
    const uint64_t r = c0 + threadIdx.x;
    const bool live = r < p.n_total;
    bool is_short = false;
    if (live) {
      const uint32_t len = p.lens[r];
      is_short = len >= p.min_len && len <= 32u && (!p.nmask || p.nmask[r] == 0ull);
      if (p.long_ok && len > 32u && len <= 63u && ((p.long_ok >> (len - 33u)) & 1ull))
        is_short = !p.nmask || (p.nmask[r] == 0ull && (!p.nmask_hi || p.nmask_hi[r] == 0ull));
    }
    const uint64_t m_short = __ballot(live && is_short), m_rest = __ballot(live && !is_short);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t o_short = 0, o_rest = 0;
    if (lane == 0) {
      if (m_short) o_short = atomicAdd(&fill[0], (uint32_t)__popcll(m_short));
      if (m_rest) o_rest = atomicAdd(&fill[1], (uint32_t)__popcll(m_rest));
    }
    o_short = __shfl(o_short, 0);
    o_rest = __shfl(o_rest, 0);
    const uint64_t below = (1ull << lane) - 1ull;
    if (live && is_short) p.idx_short[base + o_short + __popcll(m_short & below)] = (uint32_t)r;
    if (live && !is_short) p.idx_rest[base + o_rest + __popcll(m_rest & below)] = (uint32_t)r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    p.cnt_short[blockIdx.x] = fill[0];
    p.cnt_rest[blockIdx.x] = fill[1];
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
    case 8: return launch_match_w<8>(p, lds_mode, grid, lds_bytes, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_stratum(const MatchParams& p, uint32_t words_per_read, bool lds_text, uint32_t grid,
                          uint32_t lds_bytes, hipStream_t stream) {
#define MRG_STRATUM(W_)                                                                                 \
  {                                                                                                     \
    auto kern = stratum_kernel<W_>;                                                                     \
    if (lds_bytes > 48 * 1024) {                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   \
      if (e != hipSuccess) return e;                                                                    \
    }                                                                                                   \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds_bytes, stream, p, lds_text ? 1u : 0u);         \
  }
  switch (words_per_read) {
    case 1: MRG_STRATUM(1) break;
    case 2: MRG_STRATUM(2) break;
    case 4: MRG_STRATUM(4) break;
    case 8: MRG_STRATUM(8) break;
    default: return hipErrorInvalidValue;
  }
#undef MRG_STRATUM
  return hipGetLastError();
}

hipError_t launch_fused(const FusedParams& p, uint32_t words_per_read, uint32_t grid, uint32_t lds_bytes,
                        hipStream_t stream) {
#define MRG_FUSED_K(W_, P_)                                                                             \
  {                                                                                                     \
    auto kern = fused_kernel<W_, P_>;                                                                   \
    if (lds_bytes > 48 * 1024) {                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   \
      if (e != hipSuccess) return e;                                                                    \
    }                                                                                                   \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds_bytes, stream, p);                             \
  }
#define MRG_FUSED(W_)                 \
  if (pairs) MRG_FUSED_K(W_, true)    \
  else MRG_FUSED_K(W_, false)
  bool pairs = false;
  for (uint32_t q = 0; q < p.n_sub; ++q) pairs |= p.sub[q].pair_anchor != 0u;
  switch (words_per_read) {
    case 1: MRG_FUSED(1) break;
    case 2: MRG_FUSED(2) break;
    case 4: MRG_FUSED(4) break;
    case 8: MRG_FUSED(8) break;
    default: return hipErrorInvalidValue;
  }
#undef MRG_FUSED_K
#undef MRG_FUSED
  return hipGetLastError();
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

hipError_t launch_count_variants(const CountParams& p, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(count_variants_kernel, dim3(grid), dim3(kCountThreads), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_count(const CountParams& p, uint32_t words_per_read, uint32_t grid, uint32_t lds_bytes,
                        hipStream_t stream) {
  if (lds_bytes > 48 * 1024) {  // a genome part beyond ~200 Mbp: superblock table above the default dynamic-LDS cap
    const void* kerns[8] = {reinterpret_cast<const void*>(count_kernel<1, true>), reinterpret_cast<const void*>(count_kernel<1, false>),
                            reinterpret_cast<const void*>(count_kernel<2, true>), reinterpret_cast<const void*>(count_kernel<2, false>),
                            reinterpret_cast<const void*>(count_kernel<4, true>), reinterpret_cast<const void*>(count_kernel<4, false>),
                            reinterpret_cast<const void*>(count_kernel<8, true>), reinterpret_cast<const void*>(count_kernel<8, false>)};
    for (const void* k : kerns) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
  }
#define MRG_COUNT(W_)                                                                                     \
  if (p.out_ref)                                                                                         \
    hipLaunchKernelGGL((count_kernel<W_, true>), dim3(grid), dim3(kCountThreads), lds_bytes, stream, p); \
  else                                                                                                   \
    hipLaunchKernelGGL((count_kernel<W_, false>), dim3(grid), dim3(kCountThreads), lds_bytes, stream, p);
  switch (words_per_read) {
    case 1: MRG_COUNT(1) break;
    case 2: MRG_COUNT(2) break;
    case 4: MRG_COUNT(4) break;
    case 8: MRG_COUNT(8) break;
    default: return hipErrorInvalidValue;
  }
#undef MRG_COUNT
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void list_thin_kernel(const uint4* __restrict__ fat_in, uint32_t* __restrict__ thin_out,
                                                         const uint32_t* __restrict__ in_count, uint32_t* __restrict__ out_count,
                                                         uint32_t n_seg, uint32_t seg_cap) {
  for (uint32_t sg = blockIdx.x; sg < n_seg; sg += gridDim.x) {
    const uint32_t cnt = in_count[sg];
    const size_t base = (size_t)sg * seg_cap;
    for (uint32_t t = threadIdx.x; t < cnt; t += 256u) thin_out[base + t] = fat_in[base + t].x;
    if (threadIdx.x == 0) out_count[sg] = cnt;
  }
}

hipError_t launch_list_thin(const uint4* fat_in, uint32_t* thin_out, const uint32_t* in_count, uint32_t* out_count, uint32_t n_seg,
                            uint32_t seg_cap, hipStream_t stream) {
  if (!n_seg) return hipSuccess;
  hipLaunchKernelGGL(list_thin_kernel, dim3(n_seg < 2048u ? n_seg : 2048u), dim3(256), 0, stream, fat_in, thin_out, in_count, out_count,
                     n_seg, seg_cap);
  return hipGetLastError();
}

hipError_t launch_split(const SplitParams& p, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(1024), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_export_pass_counts(const uint64_t* stats, uint32_t n_pass, uint64_t* out,
                                     hipStream_t stream) {
  hipLaunchKernelGGL(export_pass_counts_kernel, dim3(1), dim3(64), 0, stream, stats, n_pass, out);
  return hipGetLastError();
}

}  // namespace mrg
