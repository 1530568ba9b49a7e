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
#include "pgzip.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace mrg {

namespace {

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
  // the block's text (whole records), split by a worker: the reader inflates / reads straight into it (no zero fill,
  // no second copy: at 4-5 GB/s of text the reader thread is what bounds a plain or parallel-gzip file)
  struct Raw {
    std::unique_ptr<char[]> p;
    size_t n = 0, cap = 0;
    char* data() { return p.get(); }
    const char* data() const { return p.get(); }
    size_t size() const { return n; }
    void reserve(size_t want, size_t keep) {
      if (want <= cap) return;
      std::unique_ptr<char[]> q(new char[want]);
      if (keep) std::memcpy(q.get(), p.get(), keep);
      p = std::move(q);
      cap = want;
    }
    void release() {
      p.reset();
      n = cap = 0;
    }
  } raw;
  uint64_t n_records = 0;
  uint64_t bad_record = 0;    // 1-based number (inside the block) of an ill-formed record
  int bad_kind = 0;           // 1 no '@', 2 truncated, 3 sequence / quality lengths differ
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
    if (stop > 255) {
      // longer than the packed form holds (eight words, one length byte): the reference accepts any
      // length, so the read is kept -- counted, listed as unannotated -- but not packed
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

namespace {

// Start of the last record header in p[0, len) that has two more line starts after it inside the
// block (so that "the line after next starts with '+'" can be tested), or 0 when there is none.
// A header line starts with '@' and is followed, two lines on, by the '+' line; a quality line
// that starts with '@' is followed two lines on by a sequence or a header line, never by '+'.
size_t last_record_start(const char* p, size_t len) {
  std::vector<size_t> starts;  // starts of the last lines, the last (possibly partial) line first
  size_t end = len;
  while (starts.size() < 64) {
    const void* nl = end ? memrchr(p, '\n', end) : nullptr;  // last newline in [0, end)
    if (!nl) {
      starts.push_back(0);
      break;
    }
    const size_t at = (size_t)((const char*)nl - p);
    starts.push_back(at + 1);
    end = at;
  }
  for (size_t i = 2; i < starts.size(); ++i) {
    const size_t s0 = starts[i], s2 = starts[i - 2];
    if (s0 > 0 && s0 < len && p[s0] == '@' && s2 < len && p[s2] == '+') return s0;
  }
  return 0;
}

// raw text of whole records -> seq / qual / off of the batch; an ill-formed record stops the
// block (b.bad_record = its 1-based number inside the block, b.bad_kind says what was wrong)
void split_block(Batch& b, bool* any_hi_first1000) {
  const char* p = b.raw.data();
  const size_t len = b.raw.size();
  size_t pos = 0;
  auto line = [&](size_t from, size_t* st, size_t* ln, size_t* next) -> bool {
    if (from >= len) return false;
    const char* nl = (const char*)std::memchr(p + from, '\n', len - from);
    size_t n = nl ? (size_t)(nl - (p + from)) : len - from;
    *next = from + n + (nl ? 1 : 0);
    if (n && p[from + n - 1] == '\r') --n;
    *st = from;
    *ln = n;
    return true;
  };
  b.off.push_back(0);
  b.seq.reserve(len / 3);
  b.qual.reserve(len / 3);
  while (pos < len) {
    size_t st[4], ln[4], nx = 0;
    if (!line(pos, &st[0], &ln[0], &nx)) break;
    if (ln[0] == 0) {  // blank line between records
      pos = nx;
      continue;
    }
    bool whole = true;
    for (int k = 1; k < 4 && whole; ++k) whole = line(nx, &st[k], &ln[k], &nx);
    if (p[st[0]] != '@') {
      b.bad_record = b.n_records + 1;
      b.bad_kind = 1;
      return;
    }
    if (!whole) {
      b.bad_record = b.n_records + 1;
      b.bad_kind = 2;
      return;
    }
    if (ln[1] != ln[3]) {
      b.bad_record = b.n_records + 1;
      b.bad_kind = 3;
      return;
    }
    if (any_hi_first1000 && b.n_records < 1000) {
      const char* q = p + st[3];
      for (size_t i = 0; i < ln[3]; ++i)
        if ((unsigned char)q[i] > 74) *any_hi_first1000 = true;
    }
    ++b.n_records;
    b.seq.insert(b.seq.end(), p + st[1], p + st[1] + ln[1]);
    b.qual.insert(b.qual.end(), p + st[3], p + st[3] + ln[3]);
    b.off.push_back((uint32_t)b.seq.size());
    pos = nx;
  }
  b.raw.release();
}

}  // namespace

size_t fastq_block_cut(const char* buf, size_t len, bool at_eof) {
  if (at_eof) return len;
  return last_record_start(buf, len);
}

// Reader (this thread: inflate, cut the text at record boundaries) -> workers (split a block into
// records, trim them) -> parallel 2-bit packing.  The per-record work is in the workers: the
// reader only finds the last record header of each ~4 MB block.
// The reference does the same with cutadapt worker processes (trim_file.py:24-66, `-cpu`).
namespace {

// the base the quality characters of a file are read with: its FIRST record decides (trim_file.py:104-110: the trimming
// workers are created while the first record is being read)
int sniff_base(const char* p, size_t len) {
  size_t pos = 0, lines = 0, qs = 0, qe = 0;
  while (pos < len && lines < 4) {
    const char* nl = (const char*)std::memchr(p + pos, '\n', len - pos);
    const size_t e = nl ? (size_t)(nl - p) : len;
    if (lines == 0 && e == pos) {  // blank lines before the first record
      pos = e + 1;
      continue;
    }
    if (lines == 3) {
      qs = pos;
      qe = e;
    }
    ++lines;
    pos = e + 1;
  }
  bool hi = false;
  for (size_t i = qs; i < qe; ++i) hi |= (unsigned char)p[i] > 74;
  return hi ? 64 : 33;
}

// A plain FASTQ file cut for `n_parts` readers: the first record start at or behind byte `from` -- a line that starts
// with '@' whose line after next starts with '+' (a quality line may start with '@' too: two lines on it has a
// sequence, never a '+') -- or `size` when there is none.  Every reader finds the same cuts.
uint64_t record_start_at_or_after(int fd, uint64_t from, uint64_t size) {
  if (from == 0) return 0;
  if (from >= size) return size;
  std::vector<char> buf;
  for (size_t window = 1u << 20;; window *= 4) {
    const size_t want = (size_t)std::min<uint64_t>(window, size - from);
    buf.resize(want);
    size_t got = 0;
    while (got < want) {
      const ssize_t r = pread(fd, buf.data() + got, want - got, (off_t)(from + got));
      if (r <= 0) break;
      got += (size_t)r;
    }
    // line starts behind the first newline (the line `from` falls into belongs to the reader in front)
    std::vector<size_t> starts;
    for (size_t i = 0; i < got; ++i)
      if (buf[i] == '\n' && i + 1 < got) starts.push_back(i + 1);
    for (size_t i = 0; i + 2 < starts.size(); ++i)
      if (buf[starts[i]] == '@' && buf[starts[i + 2]] == '+') return from + starts[i];
    if (from + got >= size) return size;
    if (window > (1u << 28)) throw std::runtime_error("no FASTQ record boundary within 256 MB");
  }
}

}  // namespace

void load_fastq(const std::string& path, int qual_cutoff, int min_len, const char* adapter, int threads,
                FastqData& out, int part, int n_parts) {
  out = FastqData();
  const TrimSpec spec = parse_trim_spec(adapter);
  if (threads <= 0) threads = (int)std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
  constexpr size_t kBlockBytes = 4u << 20;
  if (n_parts < 1 || part < 0 || part >= n_parts) throw std::runtime_error("load_fastq: part out of range");
  // One file read by several readers (`--gpus N` on a single sample): a PLAIN file is cut into n_parts byte ranges at
  // record starts, every reader takes its own; a GZIP file cannot be entered in the middle, so every reader inflates
  // all of it and takes every n_parts-th block of records (the trimming, the adapter search and the packing -- what
  // the worker threads do -- are shared out, the inflate is not).
  bool plain_range = false;
  int range_fd = -1;
  uint64_t range_pos = 0, range_end = 0;
  int forced_base = 0;
  struct FdCloser {
    int& fd;
    ~FdCloser() {
      if (fd >= 0) close(fd);
    }
  } fd_closer{range_fd};
  if (n_parts > 1) {
    range_fd = open(path.c_str(), O_RDONLY);
    if (range_fd < 0) throw std::runtime_error("cannot open " + path);
    unsigned char magic[2] = {0, 0};
    const ssize_t mg = pread(range_fd, magic, 2, 0);
    if (!(mg == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
      struct stat st;
      if (fstat(range_fd, &st) != 0) throw std::runtime_error("cannot stat " + path);
      const uint64_t size = (uint64_t)st.st_size;
      range_pos = record_start_at_or_after(range_fd, size / (uint64_t)n_parts * (uint64_t)part, size);
      range_end = part + 1 == n_parts ? size : record_start_at_or_after(range_fd, size / (uint64_t)n_parts * (uint64_t)(part + 1), size);
      plain_range = true;
      if (part > 0) {  // (the file's first record decides the quality base for every part)
        std::vector<char> head((size_t)std::min<uint64_t>(size, 1u << 16));
        const ssize_t r = pread(range_fd, head.data(), head.size(), 0);
        forced_base = sniff_base(head.data(), r > 0 ? (size_t)r : 0);
      }
    }
  }
  uint64_t block_no = 0;  // blocks of records handed out so far (gzip parts: block k belongs to part k % n_parts)

  std::deque<std::unique_ptr<Batch>> batches;  // in file order; stable addresses
  std::mutex mu;
  std::condition_variable cv;
  size_t next_job = 0;
  bool eof = false;
  int base = 33;
  bool any64 = false;  // written by the worker of block 0 only
  const bool file_head = part == 0;  // this reader's first block is the file's first block (the 1000-record sniff looks there)
  auto worker = [&]() {
    for (;;) {
      Batch* job = nullptr;
      bool first_block = false;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return next_job < batches.size() || eof; });
        if (next_job >= batches.size()) return;
        first_block = next_job == 0 && file_head;
        job = batches[next_job++].get();
      }
      try {
        split_block(*job, first_block ? &any64 : nullptr);
        if (!job->bad_record) trim_batch(*job, spec, qual_cutoff, base, min_len);
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

  // (gzip samples: inflated by pgzip.cpp's workers; plain text and one thread: zlib's reader.  The inflate gets HALF the
  // budget, at most 32: its reader thread saturates around there (scripts/pgzip_scale.sh), the trim / split pool below
  // has the same `threads`, and 2 x threads busy workers with 2 x threads + 4 look-ahead chunks of ~21 MB each were
  // 2.7 GB of transient buffers at 64)
  std::unique_ptr<GzipReader> gz;
  if (!plain_range) gz.reset(new GzipReader(path, threads <= 1 ? threads : std::min(32, std::max(2, threads / 2))));
  auto src_read = [&](char* dst, size_t n) -> size_t {
    if (!plain_range) return gz->read(dst, n);
    size_t got = 0;
    n = (size_t)std::min<uint64_t>(n, range_end - range_pos);
    while (got < n) {
      const ssize_t r = pread(range_fd, dst + got, n - got, (off_t)(range_pos + got));
      if (r < 0) throw std::runtime_error("read error on " + path);
      if (r == 0) break;
      got += (size_t)r;
    }
    range_pos += got;
    return got;
  };
  // the batch being filled: the reader reads straight into its buffer; what lies behind the last record boundary
  // starts the next batch
  std::unique_ptr<Batch> cur(new Batch());
  cur->raw.reserve(kBlockBytes + (1u << 20), 0);
  size_t have = 0;
  bool at_eof = false, started = false;
  auto hand_over = [&](size_t n_bytes) {
    std::unique_ptr<Batch> next(new Batch());
    const size_t tail = have - n_bytes;
    next->raw.reserve(std::max<size_t>(kBlockBytes + (1u << 20), tail + kBlockBytes), 0);
    if (tail) std::memcpy(next->raw.data(), cur->raw.data() + n_bytes, tail);
    cur->raw.n = n_bytes;
    std::unique_ptr<Batch> bt = std::move(cur);
    cur = std::move(next);
    have = tail;
    if (!started) {
      // trim_file.py:104-106 sniffs the first 1000 records for a quality character > 'J' (74); the
      // trimming workers are created while the first record is being read (:107-110), so only
      // that record decides the base they trim with
      base = forced_base ? forced_base : sniff_base(bt->raw.data(), bt->raw.size());
      for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
      started = true;
    }
    const bool mine = plain_range || n_parts == 1 || (block_no % (uint64_t)n_parts) == (uint64_t)part;
    ++block_no;
    if (!mine) return;  // (a gzip file's block of another part: inflated here too, trimmed and packed there)
    {
      std::lock_guard<std::mutex> lk(mu);
      batches.push_back(std::move(bt));
    }
    cv.notify_one();
  };
  auto read_more = [&](size_t until) {  // fill the current batch up to `until` bytes (or the end of the file)
    while (!at_eof && have < until) {
      if (cur->raw.cap < until) cur->raw.reserve(until + (1u << 20), have);
      const size_t got = src_read(cur->raw.data() + have, std::min<size_t>(cur->raw.cap - have, 1u << 30));
      if (got == 0) at_eof = true;
      have += got;
    }
  };
  for (;;) {
    // the first block must hold the 1000 records the quality sniff looks at
    const size_t want = started ? kBlockBytes : std::max<size_t>(kBlockBytes, 1u << 20);
    read_more(want);
    if (at_eof) {
      if (have) hand_over(have);
      break;
    }
    size_t cut = last_record_start(cur->raw.data(), have);
    if (!started) {
      // (at least 1000 records in the first block: count the lines before the cut)
      size_t lines = 0;
      for (const char* q = cur->raw.data(); cut && lines < 4004;) {
        q = (const char*)std::memchr(q, '\n', cur->raw.data() + cut - q);
        if (!q) break;
        ++lines;
        ++q;
      }
      if (cut && lines < 4004) cut = 0;
    }
    if (cut == 0) {  // no boundary yet (a record longer than the block, or a short first block): read on
      read_more(have + kBlockBytes);
      if (at_eof) {
        if (have) hand_over(have);
        break;
      }
      continue;
    }
    hand_over(cut);
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    eof = true;
  }
  cv.notify_all();
  for (auto& t : pool) t.join();
  pool.clear();

  for (size_t b = 0; b < batches.size(); ++b) {
    const Batch& bt = *batches[b];
    if (bt.bad_record) {
      const std::string rec = std::to_string(out.n_total + bt.bad_record);
      if (bt.bad_kind == 1) throw std::runtime_error(path + ": record " + rec + " does not start with '@'");
      if (bt.bad_kind == 2) throw std::runtime_error(path + ": truncated record " + rec);
      throw std::runtime_error(path + ": sequence and quality lengths differ in record " + rec);
    }
    out.n_total += bt.n_records;
  }

  out.phred = file_head ? (any64 ? 64 : 33) : 0;  // (0: this part does not hold the records the sniff reports on)
  std::vector<uint64_t> first(batches.size() + 1, 0);
  for (size_t b = 0; b < batches.size(); ++b) {
    if (!batches[b]->error.empty()) throw std::runtime_error(path + ": " + batches[b]->error);
    first[b + 1] = first[b] + batches[b]->kept_len.size();
    for (auto& lr : batches[b]->long_reads) out.long_reads.push_back(std::move(lr));
    out.max_len = std::max(out.max_len, batches[b]->max_len);
  }
  out.n_kept = first.back();
  out.words_per_read = out.max_len <= 32 ? 1 : (out.max_len <= 64 ? 2 : (out.max_len <= 128 ? 4 : 8));
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
