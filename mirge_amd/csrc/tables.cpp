// Streaming writers of the per-read output tables (host side, no GPU).
//
// Reference: writeDataToCSV.py:582-619 (mapped.csv) and :1172-1188 (unmapped.csv) walk the
// `seqDic` dict of dicts and format one text row per unique read.  At the 10^7-10^8 unique reads
// of BASELINE configs 3-5 neither that dict nor a Python loop over rows is workable, so the rows
// are formatted here straight from the columnar arrays the cascade produced (packed reads,
// pass_id, ref_id, per-sample counts), in the array order, through one large write buffer.
//   row = uniqueSequence,annotFlag,<slot 1>,...,<slot n_slots>,<count sample 1>,...
// mapped: annotFlag 1 and the claiming pass's slot holds the library entry name (RAP:341-345);
// unmapped: annotFlag 0 and every slot empty.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "tables.hpp"

namespace mrg {

uint64_t write_read_table(const char* path, bool mapped, const char* header, bool append, const uint64_t* reads,
                          uint32_t W, uint64_t stride, const uint8_t* lens, const uint64_t* nmask, uint64_t n,
                          const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant, uint32_t n_samples,
                          uint32_t n_slots, const char* const* names, const uint64_t* names_off) {
  FILE* f = std::fopen(path, append ? "ab" : "wb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  std::vector<char> buf(8u << 20);
  std::setvbuf(f, buf.data(), _IOFBF, buf.size());
  if (header && !append) std::fputs(header, f);
  static const char kBase[4] = {'A', 'C', 'G', 'T'};
  std::string row;
  row.reserve(512);
  char num[24];
  uint64_t rows = 0;
  for (uint64_t r = 0; r < n; ++r) {
    const int pass = pass_id[r];
    if (mapped != (pass >= 0)) continue;
    row.clear();
    const uint32_t L = lens[r];
    for (uint32_t i = 0; i < L; ++i) {
      const uint64_t w = reads[(uint64_t)(i >> 5) * stride + r];
      const bool is_n = nmask && ((nmask[(uint64_t)(i >> 5) * stride + r] >> ((i & 31) * 2)) & 1ull);
      row.push_back(is_n ? 'N' : kBase[(w >> ((i & 31) * 2)) & 3ull]);
    }
    row.push_back(',');
    row.push_back(mapped ? '1' : '0');
    for (uint32_t s = 0; s < n_slots; ++s) {
      row.push_back(',');
      if (mapped && (int)s == pass) {
        const uint64_t k = names_off[s] + (uint64_t)ref_id[r];
        if (k >= names_off[s + 1]) {
          std::fclose(f);
          throw std::runtime_error("write_read_table: entry index out of range for its pass");
        }
        row.append(names[k]);
      }
    }
    for (uint32_t s = 0; s < n_samples; ++s) {
      row.push_back(',');
      const int len = std::snprintf(num, sizeof num, "%u", quant[r * n_samples + s]);
      row.append(num, (size_t)len);
    }
    row.push_back('\n');
    if (std::fwrite(row.data(), 1, row.size(), f) != row.size()) {
      std::fclose(f);
      throw std::runtime_error(std::string("short write to ") + path);
    }
    ++rows;
  }
  if (std::fclose(f) != 0) throw std::runtime_error(std::string("cannot close ") + path);
  return rows;
}

}  // namespace mrg
