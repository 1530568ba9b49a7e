// FASTQ ingest on the host: parse (plain or gzip), 3' quality trimming, minimum length,
// 2-bit packing into the structure-of-arrays layout the kernels read.
//
// Reference role: trim_file (utils/trim_file.py:89-134) with `-ad none`, whose only
// modifier is cutadapt's QualityTrimmer(0, 10, phred) followed by the 16-nt minimum
// (:30-33, :52), and the FASTQ read loop of quantReads (utils/quantReads.py:4-24).
// cutadapt is a third-party dependency that is absent from the image (parity unpinned):
// its published 3' rule (BWA's) is restated -- walk from the 3' end accumulating
// (cutoff - q), stop when the sum turns negative, cut at the position of the maximum.
// Adapter removal (cutadapt's error-tolerant matching) is not built.
#include "fastq.hpp"

#include <zlib.h>

#include <cstring>
#include <stdexcept>

namespace mrg {

namespace {

struct LineReader {
  gzFile f;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  explicit LineReader(const std::string& path) : buf(1 << 20) {
    f = gzopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    gzbuffer(f, 1 << 20);
  }
  ~LineReader() {
    if (f) gzclose(f);
  }
  bool fill() {
    int got = gzread(f, buf.data(), (unsigned)buf.size());
    if (got < 0) throw std::runtime_error("read error (corrupt gzip?)");
    pos = 0;
    end = (size_t)got;
    return got > 0;
  }
  // next line without its terminator; false at EOF
  bool next(std::string& line) {
    line.clear();
    for (;;) {
      if (pos == end && !fill()) return !line.empty();
      const char* s = buf.data() + pos;
      const char* nl = (const char*)memchr(s, '\n', end - pos);
      if (nl) {
        line.append(s, nl - s);
        pos += (size_t)(nl - s) + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
      }
      line.append(s, end - pos);
      pos = end;
    }
  }
};

inline int code_of(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

}  // namespace

size_t quality_trim_3p(const char* qual, size_t len, int cutoff, int base) {
  long s = 0, best = 0;
  size_t stop = len;
  for (size_t i = len; i-- > 0;) {
    s += cutoff - ((int)(unsigned char)qual[i] - base);
    if (s < 0) break;
    if (s > best) {
      best = s;
      stop = i;
    }
  }
  return stop;
}

void load_fastq(const std::string& path, int qual_cutoff, int min_len, FastqData& out) {
  out = FastqData();
  LineReader rd(path);
  std::string name, seq, plus, qual;
  std::vector<std::string> kept_seq;
  bool worker_phred64 = false, any64 = false;
  while (rd.next(name)) {
    if (name.empty()) continue;
    if (name[0] != '@') throw std::runtime_error(path + ": record " + std::to_string(out.n_total + 1) + " does not start with '@'");
    if (!rd.next(seq) || !rd.next(plus) || !rd.next(qual))
      throw std::runtime_error(path + ": truncated record " + std::to_string(out.n_total + 1));
    if (seq.size() != qual.size())
      throw std::runtime_error(path + ": sequence and quality lengths differ in record " + std::to_string(out.n_total + 1));
    // trim_file.py:104-106 sniffs the first 1000 records for a quality character > 'J' (74);
    // the trimming workers are created while the first record is being read (:107-110), so
    // only that record decides the base they trim with
    if (out.n_total < 1000) {
      bool hi = false;
      for (char c : qual) hi |= (unsigned char)c > 74;
      if (hi) any64 = true;
      if (out.n_total == 0) worker_phred64 = hi;
    }
    ++out.n_total;
    size_t stop = quality_trim_3p(qual.data(), qual.size(), qual_cutoff, worker_phred64 ? 64 : 33);
    if ((int)stop < min_len) continue;
    if (stop > 32 * 4) throw std::runtime_error(path + ": a trimmed read of " + std::to_string(stop) + " nt exceeds the 128-nt limit");
    kept_seq.emplace_back(seq.data(), stop);
    if (stop > out.max_len) out.max_len = (uint32_t)stop;
  }
  out.phred = any64 ? 64 : 33;
  out.n_kept = kept_seq.size();
  out.words_per_read = out.max_len <= 32 ? 1 : (out.max_len <= 64 ? 2 : 4);
  const uint32_t W = out.words_per_read;
  const uint64_t n = out.n_kept;
  out.words.assign((size_t)W * n, 0);
  out.nmask.assign((size_t)W * n, 0);
  out.lens.resize(n);
  for (uint64_t r = 0; r < n; ++r) {
    const std::string& s = kept_seq[r];
    out.lens[r] = (uint8_t)s.size();
    for (size_t i = 0; i < s.size(); ++i) {
      int c = code_of(s[i]);
      const size_t at = (size_t)(i >> 5) * n + r;
      if (c < 0) {
        out.nmask[at] |= 1ull << ((i & 31) * 2);
        out.has_n = true;
      } else {
        out.words[at] |= (uint64_t)c << ((i & 31) * 2);
      }
    }
  }
}

}  // namespace mrg
