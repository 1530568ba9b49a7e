/*
 * oracle/bowtie_model.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Exhaustive-scan restatement of the alignment semantics the reference relies
 * on at its bowtie call sites (src/mirge/utils/runAnnotationPipeline.py:577-599
 * and :688: `-n N` / `-v V`, `-5`, `-3`, `-f`, `--norc`, `-a --best --strata`).
 *
 * PARITY UNPINNED at this boundary: the arithmetic lives in bowtie 1
 * (v1.1.1/1.1.2, /root/reference/README.md:49), a third-party binary that is
 * neither vendored in the reference nor present in this image, and the
 * reference has no tests or golden vectors.  What is restated here is bowtie
 * 1's published behaviour (manual, "The -n alignment mode" / "The -v alignment
 * mode"), for FASTA input (every base quality 'I' = Phred 40, Maq-rounded 30):
 *
 *   -v V : valid iff Hamming distance over the whole (trimmed) read <= V
 *   -n N : seed = first min(seed_len, len) bases; valid iff mismatches in the
 *          seed <= N and mismatches overall <= max_total (floor(70/30) = 2
 *          for -e 70)
 *   --norc : forward strand only; a read N mismatches everything; an alignment
 *            may not overlap a reference N; an alignment lies inside ONE entry
 *   a read not longer than the allowed seed mismatches is skipped (unaligned)
 *
 * Which of several valid alignments bowtie reports depends on its RNG and BWT
 * row order; this model (and the product) report the one with the fewest
 * mismatches, then the lowest entry index, then the lowest offset.  "Does the
 * read align at all" (what the cascade outcome and every count depend on) is
 * independent of that choice.
 *
 * No packing, no index: every entry, every offset, base by base.
 */
#include <stdint.h>
#include <string.h>

static int is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/* One read against one library.  Returns 1 if aligned. */
int orc_align_one(const char *lib, const uint32_t *lib_off, uint32_t n_ref, const char *read,
                  int len, int seed_len, int max_mm_seed, int max_mm_total, int32_t *out_ref,
                  int32_t *out_pos, int32_t *out_mm) {
  int best_mm = 1 << 30;
  int32_t best_ref = -1, best_pos = -1;
  if (len <= max_mm_seed || len <= 0) return 0;
  int seed = len < seed_len ? len : seed_len;
  for (uint32_t e = 0; e < n_ref; ++e) {
    const char *ref = lib + lib_off[e];
    int rlen = (int)(lib_off[e + 1] - lib_off[e]);
    for (int o = 0; o + len <= rlen; ++o) {
      int mm_seed = 0, mm_total = 0, ok = 1;
      for (int i = 0; i < len; ++i) {
        char rc = ref[o + i];
        if (!is_acgt(rc)) {
          ok = 0;
          break;
        }
        if (read[i] != rc || !is_acgt(read[i])) {
          ++mm_total;
          if (i < seed) ++mm_seed;
          if (mm_total > max_mm_total || mm_seed > max_mm_seed) {
            ok = 0;
            break;
          }
        }
      }
      if (ok && mm_total < best_mm) {
        best_mm = mm_total;
        best_ref = (int32_t)e;
        best_pos = o;
      }
    }
  }
  if (best_ref < 0) return 0;
  *out_ref = best_ref;
  *out_pos = best_pos;
  *out_mm = best_mm;
  return 1;
}

/* Many reads (concatenated ASCII, read r = reads[read_off[r] .. read_off[r+1])). */
void orc_align_batch(const char *lib, const uint32_t *lib_off, uint32_t n_ref, const char *reads,
                     const uint64_t *read_off, uint64_t n_reads, int seed_len, int max_mm_seed,
                     int max_mm_total, int32_t *out_ref, int32_t *out_pos, int32_t *out_mm) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
    int32_t ref = -1, pos = -1, mm = -1;
    int len = (int)(read_off[r + 1] - read_off[r]);
    if (!orc_align_one(lib, lib_off, n_ref, reads + read_off[r], len, seed_len, max_mm_seed,
                       max_mm_total, &ref, &pos, &mm)) {
      ref = -1;
      pos = -1;
      mm = -1;
    }
    out_ref[r] = ref;
    out_pos[r] = pos;
    out_mm[r] = mm;
  }
}

/* All valid alignments of one read in the best stratum (for `-a --best --strata`
 * consumers such as parseAlignment3, RAP:41-52).  Returns the count; fills at
 * most cap entries in (entry, offset) order. */
int orc_align_all_best(const char *lib, const uint32_t *lib_off, uint32_t n_ref, const char *read,
                       int len, int seed_len, int max_mm_seed, int max_mm_total, int32_t *refs,
                       int32_t *poss, int cap, int32_t *out_mm) {
  int32_t r0, p0, m0;
  if (!orc_align_one(lib, lib_off, n_ref, read, len, seed_len, max_mm_seed, max_mm_total, &r0, &p0,
                     &m0))
    return 0;
  *out_mm = m0;
  int seed = len < seed_len ? len : seed_len;
  int count = 0;
  for (uint32_t e = 0; e < n_ref; ++e) {
    const char *ref = lib + lib_off[e];
    int rlen = (int)(lib_off[e + 1] - lib_off[e]);
    for (int o = 0; o + len <= rlen; ++o) {
      int mm_seed = 0, mm_total = 0, ok = 1;
      for (int i = 0; i < len && ok; ++i) {
        char rc = ref[o + i];
        if (!is_acgt(rc)) ok = 0;
        else if (read[i] != rc || !is_acgt(read[i])) {
          ++mm_total;
          if (i < seed) ++mm_seed;
          if (mm_total > m0 || mm_seed > max_mm_seed) ok = 0;
        }
      }
      if (ok && mm_total == m0) {
        if (count < cap) {
          refs[count] = (int32_t)e;
          poss[count] = o;
        }
        ++count;
      }
    }
  }
  return count;
}


/* Best stratum of one read against one library (forward strand): fewest total
 * mismatches among the valid alignments and how many alignments reach it.  Used for the
 * genome filters of the -ai path (writeDataToCSV.py:1263, :1488: `-n N -a -3 2`, where the
 * reference keeps a read when its minimum-mismatch alignment is unique, :1277-1287). */
void orc_best_stratum(const char *lib, const uint32_t *lib_off, uint32_t n_ref, const char *read,
                      int len, int seed_len, int max_mm_seed, int max_mm_total, int32_t *out_mm,
                      int64_t *out_count) {
  int best = 1 << 30;
  int64_t count = 0;
  if (len > max_mm_seed && len > 0) {
    int seed = len < seed_len ? len : seed_len;
    for (uint32_t e = 0; e < n_ref; ++e) {
      const char *ref = lib + lib_off[e];
      int rlen = (int)(lib_off[e + 1] - lib_off[e]);
      for (int o = 0; o + len <= rlen; ++o) {
        int mm_seed = 0, mm_total = 0, ok = 1;
        for (int i = 0; i < len && ok; ++i) {
          char rc = ref[o + i];
          if (!is_acgt(rc)) ok = 0;
          else if (read[i] != rc || !is_acgt(read[i])) {
            ++mm_total;
            if (i < seed) ++mm_seed;
            if (mm_total > max_mm_total || mm_seed > max_mm_seed) ok = 0;
          }
        }
        if (!ok) continue;
        if (mm_total < best) {
          best = mm_total;
          count = 1;
        } else if (mm_total == best) {
          ++count;
        }
      }
    }
  }
  *out_mm = count ? best : 255;
  *out_count = count;
}

/* Every valid alignment of one read (forward strand) with its mismatch count, in
 * (entry, offset) order: what `bowtie -a` without --best/--strata lists.  Returns the
 * number found; fills at most cap. */
int orc_list_valid(const char *lib, const uint32_t *lib_off, uint32_t n_ref, const char *read, int len,
                   int seed_len, int max_mm_seed, int max_mm_total, int32_t *refs, int32_t *poss,
                   int32_t *mms, int cap) {
  int count = 0;
  if (len <= max_mm_seed || len <= 0) return 0;
  int seed = len < seed_len ? len : seed_len;
  for (uint32_t e = 0; e < n_ref; ++e) {
    const char *ref = lib + lib_off[e];
    int rlen = (int)(lib_off[e + 1] - lib_off[e]);
    for (int o = 0; o + len <= rlen; ++o) {
      int mm_seed = 0, mm_total = 0, ok = 1;
      for (int i = 0; i < len && ok; ++i) {
        char rc = ref[o + i];
        if (!is_acgt(rc)) ok = 0;
        else if (read[i] != rc || !is_acgt(read[i])) {
          ++mm_total;
          if (i < seed) ++mm_seed;
          if (mm_total > max_mm_total || mm_seed > max_mm_seed) ok = 0;
        }
      }
      if (!ok) continue;
      if (count < cap) {
        refs[count] = (int32_t)e;
        poss[count] = o;
        mms[count] = mm_total;
      }
      ++count;
    }
  }
  return count;
}
