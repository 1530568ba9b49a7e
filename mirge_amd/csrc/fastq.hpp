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
  // trimmed reads longer than 255 nt: survive trimming (they count towards "trimmedReads" on the
  // host) but cannot be packed; the host carries them as unannotated reads
  std::vector<std::string> long_reads;
};

// Index of the first base cut from the 3' end (cutadapt / BWA rule).
size_t quality_trim_3p(const char* qual, size_t len, int cutoff, int base);

// cutadapt's 3' adapter search (`-a ADAPTER`): see fastq.cpp.
struct AdapterMatch {
  bool found = false;
  size_t read_start = 0, read_stop = 0;  // the read is cut at read_start
  int adapter_stop = 0, matches = 0, errors = 0;
};
AdapterMatch locate_adapter_3p(const std::string& adapter, const char* read, size_t n, double max_error_rate,
                               int min_overlap);

// `-ad` after MAIN:123-127: "none", "+N" (drop the first N bases), or adapter[,adapter...].
struct TrimSpec {
  int cut = 0;
  std::vector<std::string> adapters;
};
TrimSpec parse_trim_spec(const char* adapter);
size_t apply_trim_spec(const TrimSpec& spec, std::string& read);  // returns the new length

// Offset of the last record boundary in a buffer of FASTQ text: buf[0, cut) holds whole records (what a
// caller that feeds text blocks to the device parser uploads), the rest is carried into the next block.
// at_eof: the buffer ends the file (everything is taken).  0 = no complete record in the buffer yet.
size_t fastq_block_cut(const char* buf, size_t len, bool at_eof);

// threads <= 0: one per hardware thread, at most 32
// part / n_parts: this reader's share of a file several readers ingest (a byte range of a plain file, every n_parts-th
// block of records of a gzip file); n_total / n_kept then count this share, phred is 0 unless part == 0
void load_fastq(const std::string& path, int qual_cutoff, int min_len, const char* adapter, int threads,
                FastqData& out, int part = 0, int n_parts = 1);

}  // namespace mrg
