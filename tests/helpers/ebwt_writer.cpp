// TEST INFRASTRUCTURE: writes a bowtie 1 `<prefix>.1.ebwt` for entries given as strings, so that the
// product's reader (mirge_amd/csrc/ebwt.cpp) can be exercised without a bowtie installation.
// Stand-alone on purpose (no product code is linked): a naive suffix sort, bowtie-build's file
// layout restated from bowtie 1.1.x -- header, plen, rstarts, the BWT in 64-byte sides (even sides
// backward, odd sides forward) whose last 8 bytes carry the occurrence counts at the
// backward/forward boundary of the side pair (A and C in the backward side, G and T in the forward
// one), zOff, fchr, ftab, eftab, names.  Built by tests/conftest.py with g++ into tests/_build/.
// Not a bowtie-built fixture: the round trip pins the reader against THIS restatement only.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

template <class T>
void put(FILE* f, T v) {
  fwrite(&v, sizeof v, 1, f);
}

int code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

}  // namespace

extern "C" int ebwt_write(const char* prefix, const char* const* names, const char* const* seqs, uint32_t n_ref, int32_t ftab_chars,
                          int32_t line_rate, int32_t lines_per_side) {
  std::vector<uint8_t> joined;
  std::vector<uint32_t> plen, rstarts;
  for (uint32_t t = 0; t < n_ref; ++t) {
    const std::string s = seqs[t];
    plen.push_back((uint32_t)s.size());
    size_t i = 0;
    while (i < s.size()) {
      while (i < s.size() && code(s[i]) < 0) ++i;
      if (i >= s.size()) break;
      rstarts.push_back((uint32_t)joined.size());
      rstarts.push_back(t);
      rstarts.push_back((uint32_t)i);
      while (i < s.size() && code(s[i]) >= 0) joined.push_back((uint8_t)code(s[i++]));
    }
  }
  const uint32_t len = (uint32_t)joined.size();
  // suffix order with '$' (the end of the text) AFTER every base, as bowtie sorts it
  std::vector<uint32_t> sa(len + 1);
  for (uint32_t i = 0; i <= len; ++i) sa[i] = i;
  std::sort(sa.begin(), sa.end(), [&](uint32_t a, uint32_t b) {
    while (a < len && b < len) {
      if (joined[a] != joined[b]) return joined[a] < joined[b];
      ++a;
      ++b;
    }
    return b == len && a != len;  // the one that runs out first is LARGER ('$' > T)
  });
  const uint32_t side_sz = (1u << line_rate) * (uint32_t)lines_per_side, side_bwt_sz = side_sz - 8, side_bwt_len = side_bwt_sz * 4;
  const uint64_t bwt_len = (uint64_t)len + 1;
  const uint64_t n_pairs = (bwt_len + 2ull * side_bwt_len - 1) / (2ull * side_bwt_len);
  std::vector<uint8_t> ebwt(n_pairs * 2 * side_sz, 0);
  uint32_t z_off = 0, occ[4] = {0, 0, 0, 0};
  for (uint64_t row = 0; row < n_pairs * 2 * side_bwt_len; ++row) {
    const uint64_t side = row / side_bwt_len;
    const uint32_t k = (uint32_t)(row % side_bwt_len);
    if (k == 0 && (side & 1)) {
      // backward -> forward boundary of the pair: A, C into the backward side's tail, G, T into the forward side's
      uint32_t* ac = reinterpret_cast<uint32_t*>(&ebwt[(side - 1) * side_sz + side_bwt_sz]);
      uint32_t* gt = reinterpret_cast<uint32_t*>(&ebwt[side * side_sz + side_bwt_sz]);
      ac[0] = occ[0];
      ac[1] = occ[1];
      gt[0] = occ[2];
      gt[1] = occ[3];
    }
    if (row >= bwt_len) continue;
    uint8_t c = 0;
    if (sa[row] == 0) {
      z_off = (uint32_t)row;  // the '$' of the BWT: stored as A, never counted
    } else {
      c = joined[sa[row] - 1];
      ++occ[c];
    }
    if (side & 1) ebwt[side * side_sz + (k >> 2)] |= (uint8_t)(c << ((k & 3) * 2));
    else ebwt[side * side_sz + side_bwt_sz - 1 - (k >> 2)] |= (uint8_t)(c << ((3 - (k & 3)) * 2));
  }
  // ftab[i] = first row whose suffix is not smaller than the i-th ftabChars-mer ('$' sorting last
  // also inside short suffixes); eftab left empty
  const uint64_t n_ftab = (1ull << (2 * ftab_chars)) + 1;
  std::vector<uint32_t> ftab(n_ftab, 0);
  {
    uint64_t next = 0;
    for (uint64_t row = 0; row < bwt_len; ++row) {
      // the largest k-mer this row's suffix is not smaller than
      uint64_t key = 0;
      bool none = false;
      for (int j = 0; j < ftab_chars; ++j) {
        const uint64_t p = (uint64_t)sa[row] + j;
        if (p >= len) {
          // P$ with j bases P: larger than every k-mer starting with P, smaller than the next one
          key = ((key + 1) << (2 * (ftab_chars - j))) - 1;
          none = false;
          break;
        }
        key = (key << 2) | joined[p];
      }
      if (none) continue;
      key = std::min<uint64_t>(key, n_ftab - 2);
      while (next <= key) ftab[next++] = (uint32_t)row;
    }
    while (next < n_ftab) ftab[next++] = (uint32_t)bwt_len;
  }
  const std::string path = std::string(prefix) + ".1.ebwt";
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return -1;
  put<int32_t>(f, 1);
  put<uint32_t>(f, len);
  put<int32_t>(f, line_rate);
  put<int32_t>(f, lines_per_side);
  put<int32_t>(f, 5);
  put<int32_t>(f, ftab_chars);
  put<int32_t>(f, -4);  // EBWT_ENTIRE_REV, what bowtie-build 1.x sets
  put<uint32_t>(f, (uint32_t)plen.size());
  fwrite(plen.data(), 4, plen.size(), f);
  put<uint32_t>(f, (uint32_t)(rstarts.size() / 3));
  fwrite(rstarts.data(), 4, rstarts.size(), f);
  fwrite(ebwt.data(), 1, ebwt.size(), f);
  put<uint32_t>(f, z_off);
  put<uint32_t>(f, 0);
  put<uint32_t>(f, occ[0]);
  put<uint32_t>(f, occ[0] + occ[1]);
  put<uint32_t>(f, occ[0] + occ[1] + occ[2]);
  put<uint32_t>(f, len);
  fwrite(ftab.data(), 4, ftab.size(), f);
  std::vector<uint32_t> eftab(2 * (size_t)ftab_chars, 0);
  fwrite(eftab.data(), 4, eftab.size(), f);
  for (uint32_t t = 0; t < n_ref; ++t) fprintf(f, "%s\n", names[t]);
  fputc('\0', f);
  fputc('\n', f);
  const bool ok = !ferror(f);
  fclose(f);
  return ok ? 0 : -2;
}
