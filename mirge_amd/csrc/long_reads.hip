// long_read_kernel: the cascade for reads of any length (gfx950, wave64).
//
// The reference offers EVERY unannotated read to every pass behind the first, whatever its length
// (writeSeqToAnnot, runAnnotationPipeline.py:543-554; the length filters of :574 are `< 26`, `> 25` or none), and
// bowtie aligns them end to end: `-v V` = at most V mismatches over the whole (trimmed) read, `-n N` = at most N in
// the first 28 bases and at most 2 overall.  The packed batches of mrg_cascade_run describe a read's
// length in one byte; what does not fit comes here: an untrimmed long-cycle run, a read-through, a concatemer --
// a handful of reads per sample, so this kernel is written for any length, not for speed.
//
// One wave per read, the read left in global memory (ragged 2-bit words + optional N mask), every pass of the
// cascade in turn until one claims it:
//   eligibility   the pass's length window and poly-T rule (RAP:664-676), decided for the whole wave (scalar);
//   seeds         max_mm_seed + 1 pigeonhole pieces of the seed region: one of them is free of mismatches in every
//                 valid alignment.  A piece is searched backwards from its end -- its last k bases through the
//                 library's largest jump table, then LF steps over the 16-byte occ blocks, every lane the same
//                 addresses -- until the interval is at most `wstop` rows wide or the piece is used up;
//   candidates    the interval's suffix-array rows dealt over the lanes (row lo + lane, + 64, ...): text position
//                 and N-free segment from the row, room on both sides from the segment table (the row's own
//                 distance fields saturate at 255), then the text against the read, 32 bases per step, with the
//                 seed's and the read's mismatch budgets; best = fewest mismatches, then lowest text position
//                 (= lowest entry, lowest offset);
//   claim         wave minimum; the winning lane decodes (entry, offset) and writes the four outputs.
// Counters (processed, aligned, LF steps, candidate rows, table lookups) go straight to global memory with one
// atomic per read and pass: there are few reads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_util.hpp"
#include "kernels.hpp"

namespace mrg {

namespace {

constexpr uint32_t kLongThreads = 256u;

// 32 bases of a ragged read from base j on (bases beyond the last word read as A = 0)
__device__ __forceinline__ uint64_t ragged_window(const uint64_t* __restrict__ w, uint32_t nwords, uint32_t j) {
  const uint32_t i = j >> 5, sh = (j & 31u) * 2u;
  const uint64_t lo = i < nwords ? w[i] : 0ull;
  if (sh == 0u) return lo;
  const uint64_t hi = i + 1u < nwords ? w[i + 1u] : 0ull;
  return (lo >> sh) | (hi << (64u - sh));
}

// first BWT row of the c-suffixes + rank of c before row i (the occ block of row i and the superblock table
// from global memory; kernels.hip: Lib::lf)
__device__ __forceinline__ uint32_t lf_global(const LongPass& ps, uint32_t c, uint32_t i) {
  const uint4 v = *reinterpret_cast<const uint4*>(ps.blocks + (size_t)(i >> 5) * 4u);
  const uint32_t r = i & 31u;
  const uint32_t pair = (c & 2u) ? v.y : v.x;
  const uint32_t cnt = (c & 1u) ? (pair >> 16) : (pair & 0xffffu);
  uint32_t e = ((c & 1u) ? v.z : ~v.z) & ((c & 2u) ? v.w : ~v.w);
  e &= (1u << r) - 1u;
  uint32_t o = ps.super[(size_t)(i >> 16) * 4u + c] + cnt + (uint32_t)__popc(e);
  o -= (uint32_t)((c == 0u) & (i > ps.primary) & ((i >> 5) == (ps.primary >> 5)));  // the sentinel row is stored as symbol 0
  return o;
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

}  // namespace

__global__ void __launch_bounds__(kLongThreads) long_read_kernel(const LongParams p) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * kLongThreads + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * kLongThreads) >> 6;
  for (uint32_t r = wave; r < p.n; r += n_waves) {
    const uint64_t off = p.word_off[r];
    const uint32_t L0 = p.lens[r];
    const uint32_t nwords = (L0 + 31u) >> 5;
    const uint64_t* __restrict__ w = p.words + off;
    const uint64_t* __restrict__ nm = p.nmask ? p.nmask + off : nullptr;
    int32_t tail = -1;  // trailing T of the read, counted when the first poly-T pass asks
    bool claimed = false;
    for (uint32_t q = 0; q < p.n_pass && !claimed; ++q) {
      const LongPass& ps = p.pass[q];
      // ---- which reads this pass's FASTA would contain (RAP:543-554, :664-686) ----
      if ((int64_t)L0 < (int64_t)ps.min_len) continue;
      if (ps.max_len < 255 && (int64_t)L0 > (int64_t)ps.max_len) continue;  // (255 and beyond: no upper bound)
      int32_t L = (int32_t)L0;
      if (ps.poly_t) {
        if (tail < 0) {
          // highest base that is not T (an N is never a T): 64 words per sweep, from the read's end
          int32_t hb = -1;
          for (int32_t top = (int32_t)nwords - 1; top >= 0 && hb < 0; top -= 64) {
            const int32_t k = top - (int32_t)lane;
            uint64_t x = 0ull;
            if (k >= 0) {
              const int32_t nb = min(32, (int32_t)L0 - 32 * k);
              const uint64_t m = nm ? nm[k] : 0ull;
              x = (~w[k] | m | (m << 1)) & dev::low_bits(2u * (uint32_t)nb);
            }
            const uint64_t any = __ballot(x != 0ull);
            if (any) {
              const int src = __ffsll((long long)any) - 1;  // the lowest lane holds the highest word
              const uint64_t xs = __shfl(x, src, 64);
              hb = 32 * (top - src) + ((63 - __clzll((long long)xs)) >> 1);
            }
          }
          tail = (int32_t)L0 - 1 - hb;
        }
        if (tail < 3 || L - tail < 11) continue;
        L -= tail;
      }
      L -= ps.trim5 + ps.trim3;
      const uint32_t a0 = (uint32_t)ps.trim5;  // effective base i of the pass = base a0 + i of the read
      uint64_t c_steps = 0, c_cands = 0, c_lookups = 0;

      uint64_t best = ~0ull;  // (mismatches << 32) | text position
      uint32_t best_sg = 0;
      if (L > ps.max_mm_seed && L > 0) {
        const int32_t R = min(L, ps.seed_len);
        const int32_t K = ps.max_mm_seed + 1;
        for (int32_t k = 0; k < K; ++k) {
          const int32_t a = (int32_t)(((int64_t)R * k) / K), b = (int32_t)(((int64_t)R * (k + 1)) / K);
          if (b <= a) continue;
          // ---- exact backward search of the piece [a, b), as far as it has to go ----
          uint32_t lo = 0, hi = ps.n + 1u;
          int32_t j = b;
          bool dead = false;  // an N inside the searched part: the piece cannot be the clean one
          uint32_t tab_off = 0;
          const uint32_t tk = ps.tabs.k[0] ? dev::pick_table(ps.tabs, b - a, tab_off) : 0u;
          if (tk) {
            j = b - (int32_t)tk;
            const uint64_t code = ragged_window(w, nwords, a0 + (uint32_t)j) & dev::low_bits(2u * tk);
            if (nm && (ragged_window(nm, nwords, a0 + (uint32_t)j) & dev::low_bits(2u * tk)) != 0ull) dead = true;
            const uint32_t* tab = ps.ftab + tab_off + dev::lex_code(code, tk);
            lo = tab[0];
            hi = tab[1];
            ++c_lookups;
          }
          while (!dead && j > a && hi > lo && (hi - lo) > p.wstop) {
            --j;
            const uint32_t at = a0 + (uint32_t)j;
            if (nm && ((nm[at >> 5] >> ((at & 31u) * 2u)) & 1ull)) {
              dead = true;
              break;
            }
            const uint32_t c = (uint32_t)(w[at >> 5] >> ((at & 31u) * 2u)) & 3u;
            lo = lf_global(ps, c, lo);
            hi = lf_global(ps, c, hi);
            ++c_steps;
          }
          if (dead || hi <= lo) continue;
          // ---- locate + verify every row of the interval, one row per lane and trip ----
          const uint32_t need_before = (uint32_t)j, need_after = (uint32_t)(L - j);
          c_cands += hi - lo;
          for (uint32_t i = lo + lane; i < hi; i += 64u) {
            const uint64_t row = ps.sa[i];
            const uint32_t s0 = (uint32_t)row;
            // the row's own distances to the ends of its segment (they saturate at 255) reject most rows -- and the
            // rows of the suffixes that are only the sentinel or run into it, which belong to no segment at all
            const uint32_t r_before = (uint32_t)(row >> 32) & 255u, r_after = (uint32_t)(row >> 40) & 255u;
            if ((r_after < 255u && need_after > r_after) || (r_before < 255u && need_before > r_before)) continue;
            uint32_t sg = (uint32_t)(row >> 48);
            if (sg == 0xFFFFu) {  // more than 65535 segments: walk the chunk map
              sg = ps.chunk_seg[s0 >> 5];
              while (ps.seg_start[sg + 1u] <= s0) ++sg;
            }
            const uint32_t seg_lo = ps.seg_start[sg], seg_hi = ps.seg_start[sg + 1u];
            if (s0 - seg_lo < need_before || seg_hi - s0 < need_after) continue;  // leaves the N-free segment
            const uint32_t s = s0 - need_before;
            int32_t mm_total = 0, mm_seed = 0;
            bool ok = true;
            for (int32_t t = 0; t < L && ok; t += 32) {
              const int32_t nb = min(32, L - t);
              uint64_t m = dev::mismatch_bits(dev::text_window(ps.text, s + (uint32_t)t), ragged_window(w, nwords, a0 + (uint32_t)t));
              if (nm) m |= ragged_window(nm, nwords, a0 + (uint32_t)t) & dev::kOddBits;
              m &= dev::low_bits(2u * (uint32_t)nb);
              mm_total += (int32_t)__popcll(m);
              const int32_t ns = min(nb, max(0, R - t));  // bases of this step that lie in the seed region
              mm_seed += (int32_t)__popcll(m & dev::low_bits(2u * (uint32_t)ns));
              ok = mm_total <= ps.max_mm_total && mm_seed <= ps.max_mm_seed;
            }
            if (!ok) continue;
            const uint64_t key = ((uint64_t)(uint32_t)mm_total << 32) | s;
            if (key < best) {
              best = key;
              best_sg = sg;
            }
          }
          if (__ballot((best >> 32) == 0ull)) break;  // an exact hit is always seen by piece 0
        }
      }
      const uint64_t wbest = wave_min_u64(best);
      const bool aligned = wbest != ~0ull;
      if (aligned) {
        const uint64_t who = __ballot(best == wbest);
        if ((int)lane == __ffsll((long long)who) - 1) {
          const uint32_t s = (uint32_t)wbest;
          uint32_t ref = best_sg, o = 0;
          if (!ps.simple_segs) {
            ref = ps.seg_ref[best_sg];
            o = ps.seg_off[best_sg];
          }
          p.pass_id[r] = (int8_t)q;
          p.ref_id[r] = (int32_t)ref;
          p.pos[r] = (int32_t)(s - ps.seg_start[best_sg] + o);
          p.mm[r] = (uint8_t)(wbest >> 32);
        }
        claimed = true;
      }
      if (lane == 0) {
        unsigned long long* c = reinterpret_cast<unsigned long long*>(p.counters) + (size_t)q * 5u;
        atomicAdd(&c[0], 1ull);
        if (aligned) atomicAdd(&c[1], 1ull);
        if (c_steps) atomicAdd(&c[2], (unsigned long long)c_steps);
        if (c_cands) atomicAdd(&c[3], (unsigned long long)c_cands);
        if (c_lookups) atomicAdd(&c[4], (unsigned long long)c_lookups);
        if (p.pass_counts) {
          unsigned long long* pc = reinterpret_cast<unsigned long long*>(p.pass_counts) + (size_t)q * 2u;
          atomicAdd(&pc[0], 1ull);
          if (aligned) atomicAdd(&pc[1], 1ull);
        }
      }
    }
    if (!claimed && lane == 0) {
      p.pass_id[r] = (int8_t)-1;
      p.ref_id[r] = -1;
      p.pos[r] = -1;
      p.mm[r] = 0;
    }
  }
}

hipError_t launch_long_reads(const LongParams& p, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(long_read_kernel, dim3(grid), dim3(kLongThreads), 0, stream, p);
  return hipGetLastError();
}

}  // namespace mrg
