// FASTQ ingest on the host: parse (plain or gzip), 3' quality trimming, minimum length,
// 2-bit packing into the structure-of-arrays layout the kernels read.
//
// Reference role: trim_file (utils/trim_file.py:89-134): cutadapt's QualityTrimmer(0, 10,
// phred), then UnconditionalCutter for `-ad +N` or AdapterCutter(error rate 0.12) for an
// adapter sequence (:30-41), then the 16-nt minimum (:33, :52); and the FASTQ read loop of
// quantReads (utils/quantReads.py:4-24).
// cutadapt (v1.11-1.16, README.md:49) is a third-party dependency that is absent from the
// image (parity unpinned); its published algorithms are restated:
//   * 3' quality rule (BWA's): walk from the 3' end accumulating (cutoff - q), stop when the
//     sum turns negative, cut at the position of the maximum;
//   * 3' adapter (`-a`): leftmost exact occurrence if there is one, else the semiglobal
//     alignment of Aligner.locate (unit costs, adapter may start anywhere in the read and may
//     run off its 3' end; among end points with overlap >= 3 and errors <= 0.12 * overlap the
//     one with most matches, then fewest errors, first found wins); the read is cut where the
//     adapter starts.  With several comma-separated adapters the one with most matches wins.
#include "fastq.hpp"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
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

AdapterMatch locate_adapter_3p(const std::string& adapter, const char* read, size_t n, double max_error_rate,
                               int min_overlap) {
  AdapterMatch none;
  const int m = (int)adapter.size();
  if (m == 0) return none;
  // exact occurrence first (Adapter.match_to)
  if (n >= (size_t)m) {
    for (size_t p = 0; p + m <= n; ++p)
      if (std::memcmp(read + p, adapter.data(), (size_t)m) == 0) {
        AdapterMatch r;
        r.found = true;
        r.read_start = p;
        r.read_stop = p + m;
        r.adapter_stop = m;
        r.matches = m;
        r.errors = 0;
        return r;
      }
  }
  struct Entry {
    int cost, matches, origin;
  };
  const int k = (int)(max_error_rate * m);
  std::vector<Entry> col((size_t)m + 1);
  for (int i = 0; i <= m; ++i) col[i] = Entry{i, 0, 0};  // adapter must start at its base 0; free start in the read
  int best_cost = m + (int)n, best_matches = 0, best_origin = 0, best_ref_stop = m;
  size_t best_query_stop = n;
  auto consider = [&](int i, size_t j) {
    const int length = i + std::min(col[i].origin, 0);
    const int cost = col[i].cost, matches = col[i].matches;
    if (length >= min_overlap && cost <= length * max_error_rate &&
        (matches > best_matches || (matches == best_matches && cost < best_cost))) {
      best_matches = matches;
      best_cost = cost;
      best_origin = col[i].origin;
      best_ref_stop = i;
      best_query_stop = j;
      return true;
    }
    return false;
  };
  int last = std::min(m, k + 1);
  bool exact_full = false;
  for (size_t j = 1; j <= n && !exact_full; ++j) {
    Entry diag = col[0];
    col[0].origin = (int)j;
    const char c = read[j - 1];
    for (int i = 1; i <= last; ++i) {
      Entry cur;
      if (adapter[i - 1] == c) {
        cur = Entry{diag.cost, diag.matches + 1, diag.origin};
      } else {
        const int c_diag = diag.cost + 1, c_del = col[i].cost + 1, c_ins = col[i - 1].cost + 1;
        if (c_diag <= c_del && c_diag <= c_ins) cur = Entry{c_diag, diag.matches, diag.origin};
        else if (c_ins <= c_del) cur = Entry{c_ins, col[i - 1].matches, col[i - 1].origin};
        else cur = Entry{c_del, col[i].matches, col[i].origin};
      }
      diag = col[i];
      col[i] = cur;
    }
    while (last >= 0 && col[last].cost > k) --last;
    if (last < m) {
      ++last;
    } else if (consider(m, j) && best_cost == 0 && best_matches == m) {
      exact_full = true;
    }
  }
  if (!exact_full) {
    // adapter running off the 3' end of the read: any row of the last column.  Rows beyond
    // `last` were not updated in the final columns; like the original they are still examined.
    for (int i = 0; i <= m; ++i) consider(i, n);
  }
  if (best_cost == m + (int)n) return none;
  AdapterMatch r;
  r.found = true;
  r.read_start = best_origin >= 0 ? (size_t)best_origin : 0;
  r.read_stop = best_query_stop;
  r.adapter_stop = best_ref_stop;
  r.matches = best_matches;
  r.errors = best_cost;
  return r;
}

TrimSpec parse_trim_spec(const char* adapter) {
  TrimSpec t;
  if (!adapter || !*adapter || std::strcmp(adapter, "none") == 0) return t;
  std::string a(adapter);
  if (a[0] == '+') {  // trim_file.py:34-35: UnconditionalCutter(int(adapter))
    char* end = nullptr;
    long v = std::strtol(a.c_str(), &end, 10);
    if (!end || *end) throw std::runtime_error("-ad " + a + ": not an integer");
    t.cut = (int)v;
    return t;
  }
  size_t at = 0;
  while (at <= a.size()) {
    size_t comma = a.find(',', at);
    if (comma == std::string::npos) comma = a.size();
    std::string one = a.substr(at, comma - at);
    for (char& ch : one) ch = (char)std::toupper((unsigned char)ch);
    if (!one.empty()) t.adapters.push_back(one);
    at = comma + 1;
  }
  return t;
}

size_t apply_trim_spec(const TrimSpec& t, std::string& read) {
  if (t.cut > 0) {
    read.erase(0, std::min(read.size(), (size_t)t.cut));
  } else if (t.cut < 0) {
    const size_t drop = std::min(read.size(), (size_t)(-t.cut));
    read.resize(read.size() - drop);
  }
  if (!t.adapters.empty()) {
    std::string upper(read);
    for (char& ch : upper) ch = (char)std::toupper((unsigned char)ch);
    AdapterMatch best;
    for (const std::string& a : t.adapters) {
      AdapterMatch mt = locate_adapter_3p(a, upper.data(), upper.size(), 0.12, std::min<int>(3, (int)a.size()));
      if (mt.found && (!best.found || mt.matches > best.matches)) best = mt;
    }
    if (best.found) read.resize(best.read_start);
  }
  return read.size();
}

void load_fastq(const std::string& path, int qual_cutoff, int min_len, const char* adapter, FastqData& out) {
  out = FastqData();
  const TrimSpec spec = parse_trim_spec(adapter);
  const bool modify = spec.cut != 0 || !spec.adapters.empty();
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
    if (modify) {
      seq.resize(stop);
      stop = apply_trim_spec(spec, seq);
    }
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
