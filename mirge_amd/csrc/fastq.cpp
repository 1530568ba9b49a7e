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
#include <atomic>
#include <cctype>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace mrg {

namespace {

// Records straight out of the inflate buffer: four memchr per record, no per-line copies.
struct RecordReader {
  gzFile f;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool at_eof = false;
  explicit RecordReader(const std::string& path) : buf(8u << 20) {
    f = gzopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    gzbuffer(f, 1 << 20);
  }
  ~RecordReader() {
    if (f) gzclose(f);
  }
  // keep the unread tail, append more input; false when nothing more arrives
  bool refill() {
    if (at_eof) return false;
    if (pos > 0) {
      std::memmove(buf.data(), buf.data() + pos, end - pos);
      end -= pos;
      pos = 0;
    }
    if (end == buf.size()) buf.resize(buf.size() * 2);  // a line longer than the buffer
    const int got = gzread(f, buf.data() + end, (unsigned)std::min<size_t>(buf.size() - end, 1u << 30));
    if (got < 0) throw std::runtime_error("read error (corrupt gzip?)");
    if (got == 0) {
      at_eof = true;
      return false;
    }
    end += (size_t)got;
    return true;
  }
  // One line [*s, *s + *n) without its terminator (CR stripped); false at end of input.
  // `from` is an offset from pos so that earlier lines of the same record stay valid: callers
  // re-derive their pointers after a refill (it may move the buffer).
  bool line_at(size_t from, size_t* start, size_t* len, size_t* next) {
    for (;;) {
      const char* base = buf.data() + pos;
      const size_t avail = end - pos;
      if (from < avail) {
        const char* nl = (const char*)std::memchr(base + from, '\n', avail - from);
        if (nl) {
          size_t n = (size_t)(nl - (base + from));
          *next = from + n + 1;
          if (n && base[from + n - 1] == '\r') --n;
          *start = from;
          *len = n;
          return true;
        }
      }
      if (!refill()) {
        if (from >= end - pos) return false;
        size_t n = (end - pos) - from;  // last line without a newline
        *next = from + n;
        if (n && buf[pos + from + n - 1] == '\r') --n;
        *start = from;
        *len = n;
        return true;
      }
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

namespace {

// A block of consecutive records.  The reader fills seq/qual/off, a worker trims every record and
// leaves only what survives (bases back to back + lengths).
struct Batch {
  std::vector<char> seq, qual;
  std::vector<uint32_t> off;  // record i = [off[i], off[i+1])
  std::string kept;
  std::vector<uint8_t> kept_len;
  std::vector<std::string> long_reads;  // trimmed reads beyond the packed limit (kept, not packed)
  uint32_t max_len = 0;
  bool has_n = false;
  std::string error;
};

void trim_batch(Batch& b, const TrimSpec& spec, int qual_cutoff, int base, int min_len) {
  const bool modify = spec.cut != 0 || !spec.adapters.empty();
  const size_t n = b.off.size() - 1;
  b.kept.reserve(b.seq.size() / 2);
  b.kept_len.reserve(n);
  std::string tmp;
  for (size_t i = 0; i < n; ++i) {
    const char* sq = b.seq.data() + b.off[i];
    const size_t len = b.off[i + 1] - b.off[i];
    size_t stop = quality_trim_3p(b.qual.data() + b.off[i], len, qual_cutoff, base);
    if (modify) {
      tmp.assign(sq, stop);
      stop = apply_trim_spec(spec, tmp);
      sq = tmp.data();
    }
    if ((int)stop < min_len) continue;
    if (stop > 32 * 4) {
      // longer than four packed words (e.g. `-ad none` on a 151-cycle run): the reference accepts
      // any length, so the read is kept -- counted, listed as unannotated -- but not packed
      b.long_reads.emplace_back(sq, stop);
      continue;
    }
    b.kept.append(sq, stop);
    b.kept_len.push_back((uint8_t)stop);
    if (stop > b.max_len) b.max_len = (uint32_t)stop;
  }
  std::vector<char>().swap(b.seq);
  std::vector<char>().swap(b.qual);
  std::vector<uint32_t>().swap(b.off);
}

}  // namespace

// Reader (this thread: inflate + split into records) -> workers (trim) -> parallel 2-bit packing.
// The reference does the same with cutadapt worker processes (trim_file.py:24-66, `-cpu`).
void load_fastq(const std::string& path, int qual_cutoff, int min_len, const char* adapter, int threads,
                FastqData& out) {
  out = FastqData();
  const TrimSpec spec = parse_trim_spec(adapter);
  if (threads <= 0) threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  constexpr size_t kBatchRecords = 1 << 15;

  std::deque<std::unique_ptr<Batch>> batches;  // in file order; stable addresses
  std::mutex mu;
  std::condition_variable cv;
  size_t next_job = 0;
  bool eof = false;
  int base = 33;
  auto worker = [&]() {
    for (;;) {
      Batch* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return next_job < batches.size() || eof; });
        if (next_job >= batches.size()) return;
        job = batches[next_job++].get();
      }
      try {
        trim_batch(*job, spec, qual_cutoff, base, min_len);
      } catch (const std::exception& e) {
        job->error = e.what();
      }
    }
  };
  std::vector<std::thread> pool;
  struct Joiner {
    std::vector<std::thread>& p;
    std::mutex& m;
    std::condition_variable& c;
    bool& eof;
    ~Joiner() {
      {
        std::lock_guard<std::mutex> lk(m);
        eof = true;
      }
      c.notify_all();
      for (auto& t : p)
        if (t.joinable()) t.join();
    }
  } joiner{pool, mu, cv, eof};

  RecordReader rd(path);
  bool any64 = false;
  std::unique_ptr<Batch> cur(new Batch());
  cur->off.push_back(0);
  auto flush = [&]() {
    if (cur->off.size() == 1) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      batches.push_back(std::move(cur));
    }
    cv.notify_one();
    cur.reset(new Batch());
    cur->off.push_back(0);
  };
  for (;;) {
    size_t st[4], ln[4], nx = 0;
    if (!rd.line_at(0, &st[0], &ln[0], &nx)) break;
    if (ln[0] == 0) {  // blank line between records
      rd.pos += nx;
      continue;
    }
    bool whole = true;
    for (int k = 1; k < 4 && whole; ++k) whole = rd.line_at(nx, &st[k], &ln[k], &nx);
    const char* rec = rd.buf.data() + rd.pos;  // valid now: no refill after the last line_at
    if (rec[st[0]] != '@') throw std::runtime_error(path + ": record " + std::to_string(out.n_total + 1) + " does not start with '@'");
    if (!whole) throw std::runtime_error(path + ": truncated record " + std::to_string(out.n_total + 1));
    if (ln[1] != ln[3])
      throw std::runtime_error(path + ": sequence and quality lengths differ in record " + std::to_string(out.n_total + 1));
    const char* seq = rec + st[1];
    const char* qual = rec + st[3];
    // trim_file.py:104-106 sniffs the first 1000 records for a quality character > 'J' (74);
    // the trimming workers are created while the first record is being read (:107-110), so
    // only that record decides the base they trim with
    if (out.n_total < 1000) {
      bool hi = false;
      for (size_t i = 0; i < ln[3]; ++i) hi |= (unsigned char)qual[i] > 74;
      if (hi) any64 = true;
      if (out.n_total == 0) {
        base = hi ? 64 : 33;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
      }
    }
    ++out.n_total;
    cur->seq.insert(cur->seq.end(), seq, seq + ln[1]);
    cur->qual.insert(cur->qual.end(), qual, qual + ln[3]);
    cur->off.push_back((uint32_t)cur->seq.size());
    rd.pos += nx;
    if (cur->off.size() > kBatchRecords) flush();
  }
  flush();
  {
    std::lock_guard<std::mutex> lk(mu);
    eof = true;
  }
  cv.notify_all();
  for (auto& t : pool) t.join();
  pool.clear();

  out.phred = any64 ? 64 : 33;
  std::vector<uint64_t> first(batches.size() + 1, 0);
  for (size_t b = 0; b < batches.size(); ++b) {
    if (!batches[b]->error.empty()) throw std::runtime_error(path + ": " + batches[b]->error);
    first[b + 1] = first[b] + batches[b]->kept_len.size();
    for (auto& lr : batches[b]->long_reads) out.long_reads.push_back(std::move(lr));
    out.max_len = std::max(out.max_len, batches[b]->max_len);
  }
  out.n_kept = first.back();
  out.words_per_read = out.max_len <= 32 ? 1 : (out.max_len <= 64 ? 2 : 4);
  const uint32_t W = out.words_per_read;
  const uint64_t n = out.n_kept;
  out.words.assign((size_t)W * n, 0);
  out.nmask.assign((size_t)W * n, 0);
  out.lens.resize(n);
  std::atomic<size_t> next_pack{0};
  auto packer = [&]() {
    for (;;) {
      const size_t b = next_pack.fetch_add(1);
      if (b >= batches.size()) return;
      Batch& bt = *batches[b];
      const char* p = bt.kept.data();
      uint64_t r = first[b];
      for (uint8_t len : bt.kept_len) {
        out.lens[r] = len;
        for (uint32_t w = 0; w < W && 32u * w < len; ++w) {
          uint64_t word = 0, mask = 0;
          const uint32_t nb = std::min<uint32_t>(32, len - 32 * w);
          for (uint32_t i = 0; i < nb; ++i) {
            const int c = code_of(p[32 * w + i]);
            if (c < 0) mask |= 1ull << (2 * i);
            else word |= (uint64_t)c << (2 * i);
          }
          out.words[(size_t)w * n + r] = word;
          out.nmask[(size_t)w * n + r] = mask;
          if (mask) bt.has_n = true;
        }
        p += len;
        ++r;
      }
      std::string().swap(bt.kept);
    }
  };
  for (int t = 0; t < threads; ++t) pool.emplace_back(packer);
  for (auto& t : pool) t.join();
  pool.clear();
  for (auto& b : batches) out.has_n |= b->has_n;
}

}  // namespace mrg
