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
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <exception>
#include <thread>
#include <vector>

#include "tables.hpp"

namespace mrg {

namespace {

struct TableArgs {
  bool mapped;
  const uint64_t* reads;
  uint32_t W;
  uint64_t stride;
  const uint8_t* lens;
  const uint64_t* nmask;
  const int8_t* pass_id;
  const int32_t* ref_id;
  const uint32_t* quant;
  uint32_t n_samples, n_slots;
  const char* const* names;
  const uint64_t* names_off;
};

// rows [lo, hi) of the table as text, appended to `out`; returns the rows written, ~0 = an entry index out of range
uint64_t format_rows(const TableArgs& a, uint64_t lo, uint64_t hi, std::string& out) {
  static const char kBase[4] = {'A', 'C', 'G', 'T'};
  char num[24];
  uint64_t rows = 0;
  for (uint64_t r = lo; r < hi; ++r) {
    const int pass = a.pass_id[r];
    if (a.mapped != (pass >= 0)) continue;
    const uint32_t L = a.lens[r];
    for (uint32_t i = 0; i < L; ++i) {
      const uint64_t w = a.reads[(uint64_t)(i >> 5) * a.stride + r];
      const bool is_n = a.nmask && ((a.nmask[(uint64_t)(i >> 5) * a.stride + r] >> ((i & 31) * 2)) & 1ull);
      out.push_back(is_n ? 'N' : kBase[(w >> ((i & 31) * 2)) & 3ull]);
    }
    out.push_back(',');
    out.push_back(a.mapped ? '1' : '0');
    for (uint32_t s = 0; s < a.n_slots; ++s) {
      out.push_back(',');
      if (a.mapped && (int)s == pass) {
        const uint64_t k = a.names_off[s] + (uint64_t)a.ref_id[r];
        if (k >= a.names_off[s + 1]) return ~0ull;
        out.append(a.names[k]);
      }
    }
    for (uint32_t s = 0; s < a.n_samples; ++s) {
      out.push_back(',');
      const int len = std::snprintf(num, sizeof num, "%u", a.quant[r * a.n_samples + s]);
      out.append(num, (size_t)len);
    }
    out.push_back('\n');
    ++rows;
  }
  return rows;
}

}  // namespace

// The rows are formatted by several threads, a block of 2^18 rows each.  Round 4 wrote the blocks of a round in order from
// ONE thread while the others waited (1 GB of unmapped.csv for 27 M reads: 1.9-2.9 s, most of it that copy into the page
// cache); round 6: a worker formats its block, learns where the block starts from the worker of the block in front (a
// chain of SIZES only: it is passed on as soon as a block is formatted, not when it is written), and writes it there itself
// with pwrite -- formatting and the writes of up to 16 blocks overlap.  Same bytes for every thread count
// (tests/test_report_tables.py; MIRGE_AMD_TABLE_THREADS overrides the count: hardware threads, at most 16).
uint64_t write_read_table(const char* path, bool mapped, const char* header, bool append, const uint64_t* reads,
                          uint32_t W, uint64_t stride, const uint8_t* lens, const uint64_t* nmask, uint64_t n,
                          const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant, uint32_t n_samples,
                          uint32_t n_slots, const char* const* names, const uint64_t* names_off) {
  const int fd = ::open(path, append ? (O_WRONLY | O_CREAT) : (O_WRONLY | O_CREAT | O_TRUNC), 0644);
  if (fd < 0) throw std::runtime_error(std::string("cannot open ") + path);
  struct Closer {
    int fd;
    ~Closer() {
      if (fd >= 0) ::close(fd);
    }
  } closer{fd};
  auto write_at = [&](const char* data, size_t len, uint64_t at) {
    while (len) {
      const ssize_t w = ::pwrite(fd, data, len, (off_t)at);
      if (w < 0) {
        if (errno == EINTR) continue;
        throw std::runtime_error(std::string("write to ") + path + " failed");
      }
      data += w;
      len -= (size_t)w;
      at += (uint64_t)w;
    }
  };
  uint64_t start = 0;
  if (append) {
    const off_t end = ::lseek(fd, 0, SEEK_END);
    if (end < 0) throw std::runtime_error(std::string("cannot seek in ") + path);
    start = (uint64_t)end;
  } else if (header) {
    write_at(header, std::strlen(header), 0);
    start = std::strlen(header);
  }
  const TableArgs a{mapped, reads, W, stride, lens, nmask, pass_id, ref_id, quant, n_samples, n_slots, names, names_off};
  unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (const char* e = std::getenv("MIRGE_AMD_TABLE_THREADS")) n_threads = (unsigned)std::max(1, std::atoi(e));
  constexpr uint64_t kBlockRows = 1u << 18;
  const uint64_t n_blocks = (n + kBlockRows - 1) / kBlockRows;
  if (n_blocks <= 1) n_threads = 1;
  n_threads = (unsigned)std::min<uint64_t>(n_threads, std::max<uint64_t>(n_blocks, 1));
  // begin[b] = file offset of block b, known once block b - 1 is formatted (begin[0] = start)
  std::vector<std::atomic<uint64_t>> begin(n_blocks + 1);
  for (auto& x : begin) x.store(~0ull, std::memory_order_relaxed);
  begin[0].store(start, std::memory_order_release);
  std::atomic<uint64_t> next{0}, rows{0};
  std::atomic<bool> stop{false};
  std::vector<std::exception_ptr> failed(n_threads);
  auto worker = [&](unsigned t) {
    std::string text;
    try {
      for (;;) {
        const uint64_t b = next.fetch_add(1, std::memory_order_relaxed);
        if (b >= n_blocks || stop.load(std::memory_order_relaxed)) break;
        const uint64_t lo = b * kBlockRows, hi = std::min(n, lo + kBlockRows);
        text.clear();
        const uint64_t got = format_rows(a, lo, hi, text);
        if (got == ~0ull) throw std::runtime_error("write_read_table: entry index out of range for its pass");
        uint64_t at;
        while ((at = begin[b].load(std::memory_order_acquire)) == ~0ull) {
          if (stop.load(std::memory_order_relaxed)) return;
          std::this_thread::yield();
        }
        begin[b + 1].store(at + text.size(), std::memory_order_release);   // (the block behind may go on before this one is written)
        if (!text.empty()) write_at(text.data(), text.size(), at);
        rows.fetch_add(got, std::memory_order_relaxed);
      }
    } catch (...) {
      // (a worker that throws -- std::bad_alloc from a growing row buffer, a full disk -- must not take the process down: the
      // exception is kept, the others stop at their next block, every thread is joined, and it is rethrown below, where the
      // C-ABI turns it into an error code)
      failed[t] = std::current_exception();
      stop.store(true, std::memory_order_relaxed);
    }
  };
  if (n_threads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> pool;
    struct Joiner {
      std::vector<std::thread>& p;
      ~Joiner() {
        for (auto& th : p)
          if (th.joinable()) th.join();
      }
    };
    {
      Joiner joiner{pool};
      try {
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(worker, t);
      } catch (...) {
        stop.store(true, std::memory_order_relaxed);
        throw;
      }
    }
  }
  for (unsigned t = 0; t < n_threads; ++t)
    if (failed[t]) std::rethrow_exception(failed[t]);
  const int fd_close = closer.fd;
  closer.fd = -1;
  if (::close(fd_close) != 0) throw std::runtime_error(std::string("cannot close ") + path);
  return rows.load();
}

// ---------------------------------------------------------------------------
// isomirs.csv + isomirs.samples.csv (writeDataToCSV.py:1090-1170; the grouping of :588-606) straight from the columnar
// arrays -- what report.write_isomir_tables(columnar.isomir_dic(columnar.read_subset(...))) writes, without the 10^5..10^6
// Python dict records in between (1.5 s of the command line's 3 s on a 32 M-read sample).  The reference's arithmetic in
// the reference's order, so that the text comes out byte for byte: str(float) of Python 2 is "%.12g" (+ ".0" for a whole
// number), math.log(x, 2) is log(x) / log(2), calcEntropy adds -1 * f * log2(f) over the entries > 1 in list order.
namespace {

void py2_float(double x, std::string& out) {
  char buf[40];
  if (x != x) {
    out += "nan";
    return;
  }
  if (x > 1.7976931348623157e308 || x < -1.7976931348623157e308) {
    out += x > 0 ? "inf" : "-inf";
    return;
  }
  const int len = std::snprintf(buf, sizeof buf, "%.12g", x);
  out.append(buf, (size_t)len);
  if (!std::memchr(buf, '.', (size_t)len) && !std::memchr(buf, 'e', (size_t)len) && !std::memchr(buf, 'n', (size_t)len)) out += ".0";
}

double calc_entropy(const uint64_t* v, size_t n) {
  uint64_t total = 0;
  for (size_t i = 0; i < n; ++i) total += v[i];
  double h = 0;
  for (size_t i = 0; i < n; ++i)
    if (v[i] > 1) {
      const double f = (double)v[i] / (double)total;
      h = h + (-1.0 * f) * (std::log(f) / std::log(2.0));
    }
  return h;
}

}  // namespace

uint64_t write_isomir_tables(const char* isomirs_path, const char* samples_path, const char* header1, const char* header2,
                             const uint64_t* reads, uint32_t W, uint64_t stride, const uint8_t* lens, const uint64_t* nmask,
                             uint64_t n, const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant, uint32_t S,
                             int32_t canon_pass, int32_t isomir_pass, const int32_t* group_of_entry, uint64_t n_entries,
                             const char* const* group_names, uint32_t n_groups, const double* filtered) {
  static const char kBase[4] = {'A', 'C', 'G', 'T'};
  // groups in the order their first read (exact-miRNA or isomiR pass) appears; the isomiR reads of each, in read order
  std::vector<uint32_t> order;
  std::vector<uint8_t> seen(n_groups, 0);
  std::vector<std::vector<uint64_t>> iso(n_groups);
  std::vector<uint64_t> canon((size_t)n_groups * S, 0);
  for (uint64_t r = 0; r < n; ++r) {
    const int p = pass_id[r];
    if (p != canon_pass && p != isomir_pass) continue;
    const int32_t e = ref_id[r];
    if (e < 0 || (uint64_t)e >= n_entries) throw std::runtime_error("write_isomir_tables: entry index out of range");
    const int32_t g = group_of_entry[e];
    if (g < 0 || (uint32_t)g >= n_groups) throw std::runtime_error("write_isomir_tables: group index out of range");
    if (!seen[g]) {
      seen[g] = 1;
      order.push_back((uint32_t)g);
    }
    if (p == isomir_pass) iso[g].push_back(r);
    else
      for (uint32_t s = 0; s < S; ++s) canon[(size_t)g * S + s] += quant[r * S + s];
  }
  FILE* f1 = std::fopen(isomirs_path, "wb");
  if (!f1) throw std::runtime_error(std::string("cannot open ") + isomirs_path);
  FILE* f2 = std::fopen(samples_path, "wb");
  if (!f2) {
    std::fclose(f1);
    throw std::runtime_error(std::string("cannot open ") + samples_path);
  }
  struct Files {
    FILE *a, *b;
    ~Files() {
      if (a) std::fclose(a);
      if (b) std::fclose(b);
    }
  } files{f1, f2};
  std::fputs(header1, f1);
  std::fputs(header2, f2);
  std::string out1, out2;
  std::vector<uint64_t> vals, counts(S);
  uint64_t rows = 0;
  for (uint32_t g : order) {
    const char* name = group_names[g];
    for (uint64_t r : iso[g]) {
      out1 += name;
      out1.push_back(',');
      const uint32_t L = lens[r];
      for (uint32_t i = 0; i < L; ++i) {
        const uint64_t w = reads[(uint64_t)(i >> 5) * stride + r];
        const bool is_n = nmask && ((nmask[(uint64_t)(i >> 5) * stride + r] >> ((i & 31) * 2)) & 1ull);
        out1.push_back(is_n ? 'N' : kBase[(w >> ((i & 31) * 2)) & 3ull]);
      }
      for (uint32_t s = 0; s < S; ++s) {
        counts[s] = quant[r * S + s];
        out1.push_back(',');
        py2_float((double)counts[s] * 1000000.0 / filtered[s], out1);
      }
      out1.push_back(',');
      if (S == 1) {
        out1 += "NA";
      } else {
        const double hmax = std::log((double)S) / std::log(2.0);
        if (hmax == 0) out1 += "NA";
        else py2_float(calc_entropy(counts.data(), S) / hmax, out1);
      }
      out1.push_back('\n');
      ++rows;
      if (out1.size() > (4u << 20)) {
        if (std::fwrite(out1.data(), 1, out1.size(), f1) != out1.size()) throw std::runtime_error("short write to isomirs.csv");
        out1.clear();
      }
    }
    // isomirs.samples.csv: the reference appends to ONE row list across the samples and writes it after each sample
    // that has isomiRs (sic)
    if (!iso[g].empty()) {
      std::string row = name;
      for (uint32_t lane = 0; lane < S; ++lane) {
        const double factor = 1000000.0 / filtered[lane];
        vals.clear();
        uint64_t top = 0, sum = 0;
        for (uint64_t r : iso[g]) {
          const uint64_t v = quant[r * S + lane];
          vals.push_back(v);
          top = std::max(top, v);
          sum += v;
        }
        const double top_rpm = (double)top * factor, iso_sum = (double)sum * factor;
        vals.push_back(canon[(size_t)g * S + lane]);
        const double h_all = calc_entropy(vals.data(), vals.size());
        const double canon_rpm = (double)canon[(size_t)g * S + lane] * factor;
        const size_t nv = vals.size();
        row.push_back(',');
        if (nv > 1) py2_float(h_all / (std::log((double)nv) / std::log(2.0)), row);
        else row += "NA";
        const double combined = canon_rpm + iso_sum;
        row.push_back(',');
        if (combined > 0) py2_float(100.0 * canon_rpm / combined, row);
        else row += "NA";
        row.push_back(',');
        py2_float(canon_rpm, row);
        row.push_back(',');
        py2_float(top_rpm, row);
        out2 += row;
        out2.push_back('\n');
      }
    }
  }
  if (!out1.empty() && std::fwrite(out1.data(), 1, out1.size(), f1) != out1.size()) throw std::runtime_error("short write to isomirs.csv");
  if (!out2.empty() && std::fwrite(out2.data(), 1, out2.size(), f2) != out2.size()) throw std::runtime_error("short write to isomirs.samples.csv");
  FILE* a = files.a;
  FILE* b = files.b;
  files.a = files.b = nullptr;
  const int ca = std::fclose(a), cb = std::fclose(b);
  if (ca != 0 || cb != 0) throw std::runtime_error("cannot close the isomiR tables");
  return rows;
}

}  // namespace mrg
