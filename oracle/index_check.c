/*
 * oracle/index_check.c -- TEST INFRASTRUCTURE, not product code.
 *
 * An index of a library checked against the library's FASTA strings, array by array, by definition
 * rather than by construction: the product builds its suffix array with SA-IS and derives everything
 * else from it (mirge_amd/csrc/fm_index.cpp); this file builds nothing -- it verifies that what the
 * GPU uploads (mrg_index_get_view) IS the FM index of those strings:
 *
 *   text, seg_*   the maximal runs of A/C/G/T of every entry, back to back, in entry order
 *   chunk_seg     segment of every 32-base chunk start
 *   sa            a permutation of 0..n whose suffixes (text + '$', '$' smallest) ascend, each row
 *                 carrying its distances to the ends of its segment (clamped to 255) and the segment id
 *   blocks/super  the BWT the suffix array implies (bit planes, 16-bit counts per 32 rows, C[c] + counts
 *                 per 65536 rows), the sentinel row `primary`
 *   ftab          T[c] = first row whose suffix starts with k-mer c or a later one, for every table
 *   kbits         the 9-mers that occur in the text
 *
 * Why: the 100 M-read parity gate of bench.py compares the GPU with oracle/fm_cpu.c, and that port reads
 * these same arrays (a different algorithm over the same index).  A construction bug -- a suffix out of
 * order, a wrong segment distance -- would be invisible to that comparison and visible only to the
 * exhaustive-scan samples.  With this check every library the bench and the tests use is pinned to its
 * strings at full size (137 Mbp: a few seconds on the host cores).
 *
 * Role in the reference: bowtie-build's output, which the reference trusts (`bowtie-inspect` is its only
 * look inside, summarize.py:6); restated layout: mirge_amd/csrc/fm_index.hpp.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  const uint32_t *blocks, *super, *text;
  const uint64_t *sa;
  const uint32_t *ftab;
  const uint32_t *seg_start, *seg_ref, *seg_off, *chunk_seg;
  const uint32_t *kbits;
  uint32_t n, primary;
  uint8_t ftab_ks[4];
} orc_lib; /* = fm_cpu.c */

enum {
  CHK_OK = 0,
  CHK_SEGMENTS = 1,   /* segment tables do not describe the entries' A/C/G/T runs */
  CHK_TEXT = 2,       /* a text base differs from its entry's */
  CHK_CHUNK_SEG = 3,
  CHK_SA_RANGE = 4,   /* a row's position is out of range or occurs twice */
  CHK_SA_ORDER = 5,   /* two neighbouring rows are out of order */
  CHK_SA_FIELDS = 6,  /* a row's segment distances / id are wrong */
  CHK_BWT = 7,        /* bit planes, counts, superblock counts or the sentinel row */
  CHK_FTAB = 8,
  CHK_KBITS = 9,
};

static inline int code_of(char ch) {
  switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

static inline uint32_t base_at(const orc_lib *l, uint32_t p) { return (l->text[p >> 4] >> ((p & 15) * 2)) & 3u; }

/* 32 bases from p (first base in the low two bits); the text is padded behind n */
static inline uint64_t window(const orc_lib *l, uint32_t p) {
  uint32_t i = p >> 4, sh = (p & 15) * 2;
  uint64_t lo64 = (uint64_t)l->text[i] | ((uint64_t)l->text[i + 1] << 32);
  return sh ? (lo64 >> sh) | ((uint64_t)l->text[i + 2] << (64 - sh)) : lo64;
}

/* suffix a < suffix b in text + '$' ('$' smallest: of two suffixes one of which is a prefix of the other the shorter one) */
static int suffix_less(const orc_lib *l, uint32_t a, uint32_t b) {
  const uint32_t n = l->n;
  uint32_t left = n - (a > b ? a : b); /* bases both suffixes have */
  while (left) {
    uint64_t x = window(l, a) ^ window(l, b);
    uint32_t take = left < 32 ? left : 32;
    if (take < 32) x &= (1ull << (2 * take)) - 1ull;
    if (x) {
      uint32_t k = (uint32_t)__builtin_ctzll(x) >> 1;
      return base_at(l, a + k) < base_at(l, b + k);
    }
    a += take;
    b += take;
    left -= take;
  }
  return a > b; /* equal on the common part: the one that ends first */
}

/* report: [0] failing check, [1] where (row / position / segment / table entry), [2] [3] got / want where that helps */
static int fail(uint64_t *report, int code, uint64_t where, uint64_t got, uint64_t want) {
#pragma omp critical
  {
    if (!report[0]) {
      report[0] = (uint64_t)code;
      report[1] = where;
      report[2] = got;
      report[3] = want;
    }
  }
  return code;
}

/* seqs: the entries' strings back to back, entry r = [seq_off[r], seq_off[r + 1]).  n_seg: segments in the view.
 * Returns 0 or the first failing check (details in report[0..3]; report[4..7]: rows compared, longest common
 * prefix met, k-mers checked, 0). */
int orc_check_index(const orc_lib *l, const char *seqs, const uint64_t *seq_off, uint32_t n_ref, uint32_t n_seg,
                    uint64_t *report) {
  memset(report, 0, 8 * sizeof(uint64_t));
  const uint32_t n = l->n;
  const uint64_t m = (uint64_t)n + 1;

  /* ---- segments and text, from the strings ---- */
  {
    uint32_t sg = 0;
    uint64_t pos = 0;
    for (uint32_t r = 0; r < n_ref; ++r) {
      const char *s = seqs + seq_off[r];
      const uint64_t L = seq_off[r + 1] - seq_off[r];
      uint64_t i = 0;
      while (i < L) {
        if (code_of(s[i]) < 0) {
          ++i;
          continue;
        }
        if (sg >= n_seg || l->seg_start[sg] != pos || l->seg_ref[sg] != r || l->seg_off[sg] != i)
          return fail(report, CHK_SEGMENTS, sg, sg < n_seg ? l->seg_start[sg] : 0, pos);
        while (i < L && code_of(s[i]) >= 0) {
          if (pos >= n || base_at(l, (uint32_t)pos) != (uint32_t)code_of(s[i])) return fail(report, CHK_TEXT, pos, 0, 0);
          ++pos;
          ++i;
        }
        ++sg;
      }
    }
    if (sg != n_seg || pos != n || l->seg_start[n_seg] != n) return fail(report, CHK_SEGMENTS, sg, pos, n);
    /* padding behind the text reads as zeros (windows at the last positions rely on it) */
    for (uint32_t p = n; p < ((n + 15) / 16 + 4) * 16; ++p)
      if (base_at(l, p)) return fail(report, CHK_TEXT, p, 1, 0);
  }
  /* ---- chunk -> segment ---- */
  {
    const uint32_t nchunk = (n >> 5) + 2;
    uint32_t sg = 0;
    for (uint32_t ch = 0; ch < nchunk; ++ch) {
      const uint64_t p = (uint64_t)ch << 5;
      while (sg + 1 < n_seg && l->seg_start[sg + 1] <= p) ++sg;
      if (l->chunk_seg[ch] != sg) return fail(report, CHK_CHUNK_SEG, ch, l->chunk_seg[ch], sg);
    }
  }
  /* ---- suffix array: a permutation of 0..n ---- */
  {
    uint64_t *seen = (uint64_t *)calloc((m + 63) / 64, 8);
    for (uint64_t i = 0; i < m; ++i) {
      const uint32_t p = (uint32_t)l->sa[i];
      if (p > n || (seen[p >> 6] >> (p & 63)) & 1ull) {
        free(seen);
        return fail(report, CHK_SA_RANGE, i, p, n);
      }
      seen[p >> 6] |= 1ull << (p & 63);
    }
    free(seen);
    if ((uint32_t)l->sa[0] != n) return fail(report, CHK_SA_ORDER, 0, (uint32_t)l->sa[0], n);
  }
  /* ---- ... in ascending order of the suffixes; row fields ---- */
  {
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 65536) reduction(| : bad)
    for (int64_t i = 0; i < (int64_t)m; ++i) {
      if (bad) continue;
      const uint64_t row = l->sa[i];
      const uint32_t p = (uint32_t)row;
      if (i + 1 < (int64_t)m && !suffix_less(l, p, (uint32_t)l->sa[i + 1])) bad |= fail(report, CHK_SA_ORDER, (uint64_t)i, p, (uint32_t)l->sa[i + 1]);
      uint64_t want;
      if (p < n) {
        uint32_t sg = l->chunk_seg[p >> 5];
        while (l->seg_start[sg + 1] <= p) ++sg;
        const uint32_t before = p - l->seg_start[sg] < 255u ? p - l->seg_start[sg] : 255u;
        const uint32_t after = l->seg_start[sg + 1] - p < 255u ? l->seg_start[sg + 1] - p : 255u;
        const uint32_t sid = n_seg <= 0xFFFFu ? sg : 0xFFFFu;
        want = (uint64_t)p | (uint64_t)before << 32 | (uint64_t)after << 40 | (uint64_t)sid << 48;
      } else {
        want = (uint64_t)p | (uint64_t)0xFFFFu << 48;
      }
      if (row != want) bad |= fail(report, CHK_SA_FIELDS, (uint64_t)i, row >> 32, want >> 32);
    }
    if (bad) return (int)report[0];
    report[4] = m;
  }
  /* ---- the BWT the suffix array implies ---- */
  {
    uint32_t C[4] = {0, 0, 0, 0};
    const uint64_t nblk = (m >> 5) + 1, nsup = (m >> 16) + 1;
    /* symbol of every row (4 = the sentinel row) and the symbol counts of every superblock */
    uint8_t *bwt = (uint8_t *)malloc(m);
    uint32_t *cnt = (uint32_t *)calloc(nsup * 4, 4);
    int64_t n_primary = 0, primary = -1;
#pragma omp parallel for schedule(static) reduction(+ : n_primary) reduction(max : primary)
    for (int64_t sb = 0; sb < (int64_t)nsup; ++sb) {
      const uint64_t lo = (uint64_t)sb << 16, hi = lo + 65536 < m ? lo + 65536 : m;
      for (uint64_t i = lo; i < hi; ++i) {
        const uint32_t p = (uint32_t)l->sa[i];
        if (p == 0) {
          bwt[i] = 4;
          ++n_primary;
          primary = (int64_t)i;
          continue;
        }
        const uint32_t c = base_at(l, p - 1);
        bwt[i] = (uint8_t)c;
        ++cnt[sb * 4 + c];
      }
    }
    uint64_t tot[4] = {0, 0, 0, 0};
    for (uint64_t sb = 0; sb < nsup; ++sb)
      for (int c = 0; c < 4; ++c) {
        const uint32_t k = cnt[sb * 4 + c];
        cnt[sb * 4 + c] = (uint32_t)tot[c]; /* symbols c in front of the superblock */
        tot[c] += k;
      }
    uint32_t sum = 1; /* row 0 is the sentinel suffix */
    for (int c = 0; c < 4; ++c) {
      C[c] = sum;
      sum += (uint32_t)tot[c];
    }
    int bad = 0;
    if (n_primary != 1 || primary != (int64_t)l->primary) bad = fail(report, CHK_BWT, (uint64_t)primary, l->primary, (uint64_t)primary);
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t sb = 0; sb < (int64_t)nsup; ++sb) {
      if (bad) continue;
      for (int c = 0; c < 4; ++c)
        if (l->super[sb * 4 + c] != C[c] + cnt[sb * 4 + c]) bad |= fail(report, CHK_BWT, (uint64_t)sb << 16, l->super[sb * 4 + c], C[c] + cnt[sb * 4 + c]);
      uint32_t run[4] = {0, 0, 0, 0};
      const uint64_t b_lo = (uint64_t)sb << 11, b_hi = b_lo + 2048 < nblk ? b_lo + 2048 : nblk;
      for (uint64_t b = b_lo; b < b_hi; ++b) {
        const uint32_t *blk = l->blocks + b * 4;
        for (int c = 0; c < 4; ++c) {
          const uint32_t got = (blk[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
          if (got != run[c]) bad |= fail(report, CHK_BWT, b << 5, got, run[c]);
        }
        uint32_t lo = 0, hi = 0;
        for (uint64_t i = b << 5; i < (b << 5) + 32 && i < m; ++i) {
          const uint32_t c = bwt[i];
          if (c > 3) continue; /* the sentinel row: stored as symbol 0, never counted */
          lo |= (c & 1u) << (i & 31);
          hi |= ((c >> 1) & 1u) << (i & 31);
          ++run[c];
        }
        if (blk[2] != lo || blk[3] != hi) bad |= fail(report, CHK_BWT, b << 5, blk[2], lo);
      }
    }
    free(bwt);
    free(cnt);
    if (bad) return (int)report[0];
  }
  /* ---- jump tables: T[c] = first row whose k-mer code is c or a later one (codes: first base most significant; a
   * suffix shorter than k counts as padded with A -- it sorts in front of every k-mer it is a prefix of, the
   * verification of a candidate row drops it by its distance to the segment end) ---- */
  {
    uint64_t off = 0, checked = 0;
    for (int t = 0; t < 4; ++t) {
      const uint32_t k = l->ftab_ks[t];
      if (!k) continue;
      const uint32_t *T = l->ftab + off;
      const uint64_t n_codes = 1ull << (2 * k);
      off += n_codes + 1;
      if (T[0] != 0 || T[n_codes] != m) return fail(report, CHK_FTAB, (uint64_t)t << 56, T[n_codes], m);
      int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
      for (int64_t c = 0; c < (int64_t)n_codes; ++c)
        if (T[c] > T[c + 1]) bad |= fail(report, CHK_FTAB, ((uint64_t)t << 56) | (uint64_t)c, T[c], T[c + 1]);
      if (bad) return (int)report[0];
      /* every row lies in the interval of its own code: with the rows in order and T monotone that fixes every entry */
#pragma omp parallel for schedule(static) reduction(| : bad)
      for (int64_t i = 0; i < (int64_t)m; ++i) {
        const uint32_t p = (uint32_t)l->sa[i];
        const uint32_t have = n - p < k ? n - p : k;
        const uint64_t w = window(l, p);
        uint64_t code = 0;
        for (uint32_t q = 0; q < have; ++q) code = (code << 2) | ((w >> (2 * q)) & 3ull);
        code <<= 2 * (k - have);
        if (!(T[code] <= (uint64_t)i && (uint64_t)i < T[code + 1])) bad |= fail(report, CHK_FTAB, ((uint64_t)t << 56) | code, T[code], (uint64_t)i);
      }
      if (bad) return (int)report[0];
      checked += n_codes;
    }
    report[6] = checked;
  }
  /* ---- 9-mer presence bitmap (first base in the low two bits) ---- */
  if (l->kbits) {
    const uint32_t words = (1u << 18) / 32u;
    uint32_t *want = (uint32_t *)calloc(words, 4);
    for (uint32_t p = 0; p + 9 <= n; ++p) {
      const uint32_t c = (uint32_t)window(l, p) & ((1u << 18) - 1u);
      want[c >> 5] |= 1u << (c & 31);
    }
    for (uint32_t i = 0; i < words; ++i)
      if (want[i] != l->kbits[i]) {
        const uint32_t got = l->kbits[i], w = want[i];
        free(want);
        return fail(report, CHK_KBITS, i, got, w);
      }
    free(want);
  }
  return CHK_OK;
}
