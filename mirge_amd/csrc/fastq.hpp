// Host-side FASTQ ingest (internal header); see fastq.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mrg {

struct FastqData {
  uint64_t n_total = 0;   // records in the file ("totalReads", MAIN:362)
  uint64_t n_kept = 0;    // records that survive trimming + min length ("trimmedReads")
  int phred = 33;         // what the reference's sniffing reports (TRM:104-106)
  uint32_t words_per_read = 1, max_len = 0;
  bool has_n = false;
  std::vector<uint64_t> words, nmask;  // [W][n_kept]
  std::vector<uint8_t> lens;
};

// Index of the first base cut from the 3' end (cutadapt / BWA rule).
size_t quality_trim_3p(const char* qual, size_t len, int cutoff, int base);
void load_fastq(const std::string& path, int qual_cutoff, int min_len, FastqData& out);

}  // namespace mrg
