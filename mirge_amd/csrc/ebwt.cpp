// Reader (and a test-only writer) for bowtie 1 `.1.ebwt` index files.
//
// Role in the reference: miRge.Libs/<species>/index.Libs ships ONLY bowtie 1 indexes
// (MAIN:262-281 checks for `<prefix>.1.ebwt`); entry names and sequences are recovered from them
// at run time with `bowtie-inspect` (SUM:6 -n, RAP:610-611,630, W2C:649).  This file is that
// recovery for the engine: names + sequences out of `<prefix>.1.ebwt`, which then go through
// mrg::build_index like a FASTA file would -- so the library directory of the reference works
// unchanged, without a bowtie installation.
//
// Format, restated from bowtie 1.1.x (ebwt.h: Ebwt::readIntoMemory / buildToDisk / restore /
// joinedToTextOff; bowtie_inspect.cpp: print_index_sequences); bowtie's source is NOT in
// /root/reference and the image has neither bowtie nor a sample index, so this restatement is
// VALIDATED BY ROUND TRIP ONLY (write_ebwt below -> read_ebwt) until a bowtie-built fixture exists:
//   int32  1                      endianness hint
//   uint32 len                    joined text length (unambiguous bases of all references)
//   int32  lineRate, linesPerSide, offRate, ftabChars, flags (negated flag bits; colour = 2)
//   uint32 nPat,  uint32 plen[nPat]          reference lengths, ambiguous bases included
//   uint32 nFrag, uint32 rstarts[3 * nFrag]  per unambiguous stretch: joined offset, reference, offset in it
//   uint8  ebwt[numSides << lineRate]        the BWT in "sides" of (1 << lineRate) bytes: the last 8
//                                            bytes of a side hold two occurrence counts, the rest
//                                            four 2-bit characters per byte; even sides are
//                                            "backward" (row k of the side at byte sideBwtSz-1-k/4,
//                                            bit pair 3-(k&3)), odd sides "forward" (byte k/4, pair k&3)
//   uint32 zOff                   row whose BWT character is the '$' (stored as A)
//   uint32 fchr[5], uint32 ftab[4^ftabChars + 1], uint32 eftab[2 * ftabChars]
//   reference names, one per line, closed by a NUL
// bowtie sorts '$' AFTER every base; restore() walks LF from row len (the "$" suffix) to zOff.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "fm_index.hpp"

namespace mrg {
namespace {

template <class T>
T get(std::ifstream& in) {
  T v;
  in.read(reinterpret_cast<char*>(&v), sizeof v);
  if (!in) throw std::runtime_error("truncated .ebwt file");
  return v;
}
template <class T>
void put(std::ofstream& out, T v) {
  out.write(reinterpret_cast<const char*>(&v), sizeof v);
}

struct Sides {
  uint32_t side_sz, side_bwt_sz, side_bwt_len;
  explicit Sides(int line_rate) : side_sz(1u << line_rate), side_bwt_sz(side_sz - 8), side_bwt_len(side_bwt_sz * 4) {}
  uint64_t total(uint64_t bwt_len) const {
    const uint64_t pairs = (bwt_len + 2ull * side_bwt_len - 1) / (2ull * side_bwt_len);
    return pairs * 2 * side_sz;
  }
  // byte offset and bit shift of BWT row i
  void locate(uint64_t i, uint64_t& byte, uint32_t& shift) const {
    const uint64_t side = i / side_bwt_len;
    const uint32_t k = (uint32_t)(i % side_bwt_len);
    if (side & 1) {  // forward
      byte = side * side_sz + (k >> 2);
      shift = (k & 3) * 2;
    } else {         // backward
      byte = side * side_sz + side_bwt_sz - 1 - (k >> 2);
      shift = (3 - (k & 3)) * 2;
    }
  }
};

}  // namespace

void read_ebwt(const std::string& prefix, std::vector<std::string>& names, std::vector<std::string>& seqs) {
  const std::string path = prefix + ".1.ebwt";
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + path);
  if (get<int32_t>(in) != 1) throw std::runtime_error(path + ": not a little-endian bowtie 1 index");
  const uint32_t len = get<uint32_t>(in);
  const int32_t line_rate = get<int32_t>(in);
  (void)get<int32_t>(in);  // linesPerSide
  (void)get<int32_t>(in);  // offRate
  const int32_t ftab_chars = get<int32_t>(in);
  const int32_t flags = get<int32_t>(in);
  if (line_rate < 4 || line_rate > 16 || ftab_chars < 1 || ftab_chars > 14)
    throw std::runtime_error(path + ": implausible header (lineRate / ftabChars)");
  if (flags < 0 && ((-flags) & 2)) throw std::runtime_error(path + ": colourspace indexes are not supported");
  const uint32_t n_pat = get<uint32_t>(in);
  std::vector<uint32_t> plen(n_pat);
  in.read(reinterpret_cast<char*>(plen.data()), (std::streamsize)n_pat * 4);
  const uint32_t n_frag = get<uint32_t>(in);
  std::vector<uint32_t> rstarts((size_t)n_frag * 3);
  in.read(reinterpret_cast<char*>(rstarts.data()), (std::streamsize)rstarts.size() * 4);
  const Sides sd(line_rate);
  const uint64_t bwt_len = (uint64_t)len + 1;
  std::vector<uint8_t> ebwt(sd.total(bwt_len));
  in.read(reinterpret_cast<char*>(ebwt.data()), (std::streamsize)ebwt.size());
  if (!in) throw std::runtime_error(path + ": truncated (BWT)");
  const uint32_t z_off = get<uint32_t>(in);
  if (z_off > len) throw std::runtime_error(path + ": zOff out of range");
  {
    const std::streamoff here = in.tellg();
    const std::streamoff skip = (std::streamoff)(5 + ((1ull << (2 * ftab_chars)) + 1) + 2ull * ftab_chars) * 4;
    in.seekg(0, std::ios::end);
    const std::streamoff size = in.tellg();
    if (here < 0 || here + skip > size) throw std::runtime_error(path + ": truncated (fchr / ftab)");
    in.seekg(here + skip, std::ios::beg);
  }
  names.clear();
  {
    std::string cur;
    char c;
    while (in.get(c)) {
      if (c == '\0') break;
      if (c == '\n') {
        names.push_back(cur);
        cur.clear();
      } else {
        cur.push_back(c);
      }
    }
    if (!cur.empty()) names.push_back(cur);
  }
  // ---- BWT rows, rank checkpoints every 64 rows ('$' placeholder excluded), LF walk ----
  std::vector<uint8_t> bwt(bwt_len);
  for (uint64_t i = 0; i < bwt_len; ++i) {
    uint64_t byte;
    uint32_t shift;
    sd.locate(i, byte, shift);
    bwt[i] = (ebwt[byte] >> shift) & 3;
  }
  std::vector<uint8_t>().swap(ebwt);
  const uint64_t n_chk = bwt_len / 64 + 1;
  std::vector<uint32_t> chk(n_chk * 4);
  uint32_t run[4] = {0, 0, 0, 0};
  for (uint64_t i = 0; i < bwt_len; ++i) {
    if ((i & 63) == 0) std::memcpy(&chk[(i >> 6) * 4], run, sizeof run);
    if (i != z_off) ++run[bwt[i]];
  }
  if ((bwt_len & 63) == 0) std::memcpy(&chk[(bwt_len >> 6) * 4], run, sizeof run);
  uint32_t fchr[5] = {0, run[0], run[0] + run[1], run[0] + run[1] + run[2], len};
  if ((uint64_t)run[0] + run[1] + run[2] + run[3] != len) throw std::runtime_error(path + ": BWT length mismatch");
  std::vector<uint8_t> text(len);
  uint64_t i = len, jumps = 0;
  while (i != z_off) {
    if (jumps >= len) throw std::runtime_error(path + ": the LF walk does not close (corrupt BWT?)");
    const uint8_t c = bwt[i];
    uint32_t occ = chk[(i >> 6) * 4 + c];
    for (uint64_t j = i & ~63ull; j < i; ++j) occ += (bwt[j] == c && j != z_off);
    text[len - jumps - 1] = c;
    i = (uint64_t)fchr[c] + occ;
    ++jumps;
  }
  if (jumps != len) throw std::runtime_error(path + ": the LF walk ended early (corrupt BWT?)");
  // ---- joined text -> references (bowtie-inspect's print_index_sequences) ----
  static const char L[4] = {'A', 'C', 'G', 'T'};
  seqs.assign(n_pat, std::string());
  for (uint32_t t = 0; t < n_pat; ++t) seqs[t].assign(plen[t], 'N');
  for (uint32_t f = 0; f < n_frag; ++f) {
    const uint64_t a = rstarts[3 * f], b = f + 1 < n_frag ? rstarts[3 * (f + 1)] : len;
    const uint32_t t = rstarts[3 * f + 1], off = rstarts[3 * f + 2];
    if (t >= n_pat || a > b || b > len || (uint64_t)off + (b - a) > plen[t])
      throw std::runtime_error(path + ": fragment table out of range");
    for (uint64_t k = a; k < b; ++k) seqs[t][off + (k - a)] = L[text[k]];
  }
  if (names.size() < n_pat)
    for (size_t t = names.size(); t < n_pat; ++t) names.push_back(std::to_string(t));
  names.resize(n_pat);
  // bowtie-build keeps the whole FASTA header; bowtie reports (and bowtie-inspect -n prints) it
  // up to the first whitespace
  for (auto& nm : names) {
    const size_t ws = nm.find_first_of(" \t");
    if (ws != std::string::npos) nm.resize(ws);
  }
}

// TEST-ONLY writer: enough of bowtie-build to round-trip the reader (the occurrence counts inside
// the sides and the ftab / eftab tables are filled with zeros: a real bowtie could not search this
// file, and nothing here reads them).
void write_ebwt(const std::string& prefix, const std::vector<std::string>& names, const std::vector<std::string>& seqs,
                int ftab_chars) {
  std::vector<uint8_t> joined;
  std::vector<uint32_t> plen, rstarts;
  for (size_t t = 0; t < seqs.size(); ++t) {
    const std::string& s = seqs[t];
    plen.push_back((uint32_t)s.size());
    size_t i = 0;
    while (i < s.size()) {
      auto code = [](char c) -> int {
        switch (c) {
          case 'A': case 'a': return 0;
          case 'C': case 'c': return 1;
          case 'G': case 'g': return 2;
          case 'T': case 't': return 3;
          default: return -1;
        }
      };
      while (i < s.size() && code(s[i]) < 0) ++i;
      if (i >= s.size()) break;
      rstarts.push_back((uint32_t)joined.size());
      rstarts.push_back((uint32_t)t);
      rstarts.push_back((uint32_t)i);
      while (i < s.size() && code(s[i]) >= 0) joined.push_back((uint8_t)code(s[i++]));
    }
  }
  const uint32_t len = (uint32_t)joined.size();
  // suffix order with '$' after every base: symbols 1..4 = ACGT, 5 = bowtie's '$', 0 = SA-IS sentinel
  std::vector<int32_t> s(len + 2), sa(len + 2);
  for (uint32_t i = 0; i < len; ++i) s[i] = joined[i] + 1;
  s[len] = 5;
  s[len + 1] = 0;
  suffix_array(s.data(), sa.data(), (int32_t)len + 2, 6);
  const int line_rate = 6;
  const Sides sd(line_rate);
  const uint64_t bwt_len = (uint64_t)len + 1;
  std::vector<uint8_t> ebwt(sd.total(bwt_len), 0);
  uint32_t z_off = 0, counts[4] = {0, 0, 0, 0};
  for (uint64_t row = 0; row < bwt_len; ++row) {
    const int32_t p = sa[row + 1];  // sa[0] is the SA-IS sentinel
    uint8_t c = 0;
    if (p == 0) {
      z_off = (uint32_t)row;
    } else {
      c = joined[p - 1];
      ++counts[c];
    }
    uint64_t byte;
    uint32_t shift;
    sd.locate(row, byte, shift);
    ebwt[byte] |= (uint8_t)(c << shift);
  }
  const std::string path = prefix + ".1.ebwt";
  std::ofstream out(path, std::ios::binary);
  if (!out) throw std::runtime_error("cannot write " + path);
  put<int32_t>(out, 1);
  put<uint32_t>(out, len);
  put<int32_t>(out, line_rate);
  put<int32_t>(out, 1);
  put<int32_t>(out, 5);
  put<int32_t>(out, ftab_chars);
  put<int32_t>(out, -4);  // EBWT_ENTIRE_REV, what bowtie-build 1.x sets
  put<uint32_t>(out, (uint32_t)plen.size());
  out.write(reinterpret_cast<const char*>(plen.data()), (std::streamsize)plen.size() * 4);
  put<uint32_t>(out, (uint32_t)(rstarts.size() / 3));
  out.write(reinterpret_cast<const char*>(rstarts.data()), (std::streamsize)rstarts.size() * 4);
  out.write(reinterpret_cast<const char*>(ebwt.data()), (std::streamsize)ebwt.size());
  put<uint32_t>(out, z_off);
  put<uint32_t>(out, 0);
  put<uint32_t>(out, counts[0]);
  put<uint32_t>(out, counts[0] + counts[1]);
  put<uint32_t>(out, counts[0] + counts[1] + counts[2]);
  put<uint32_t>(out, len);
  std::vector<uint32_t> zeros(((size_t)1 << (2 * ftab_chars)) + 1 + 2 * (size_t)ftab_chars, 0);
  out.write(reinterpret_cast<const char*>(zeros.data()), (std::streamsize)zeros.size() * 4);
  for (const auto& n : names) out << n << '\n';
  out << '\0' << '\n';
  if (!out) throw std::runtime_error("short write to " + path);
}

}  // namespace mrg
