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
#include <algorithm>
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

// The rows are formatted by several threads, a block of rows each, and written in order: at 10^7-10^8 rows one thread
// formatting a gigabyte of text was most of the command line's "Summarizing" phase (MIRGE_AMD_TABLE_THREADS overrides
// the thread count: hardware threads, at most 16).
uint64_t write_read_table(const char* path, bool mapped, const char* header, bool append, const uint64_t* reads,
                          uint32_t W, uint64_t stride, const uint8_t* lens, const uint64_t* nmask, uint64_t n,
                          const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant, uint32_t n_samples,
                          uint32_t n_slots, const char* const* names, const uint64_t* names_off) {
  FILE* f = std::fopen(path, append ? "ab" : "wb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  std::vector<char> buf(8u << 20);
  std::setvbuf(f, buf.data(), _IOFBF, buf.size());
  if (header && !append) std::fputs(header, f);
  const TableArgs a{mapped, reads, W, stride, lens, nmask, pass_id, ref_id, quant, n_samples, n_slots, names, names_off};
  unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (const char* e = std::getenv("MIRGE_AMD_TABLE_THREADS")) n_threads = (unsigned)std::max(1, std::atoi(e));
  constexpr uint64_t kBlockRows = 1u << 18;
  if (n <= kBlockRows) n_threads = 1;
  std::vector<std::string> text(n_threads);
  std::vector<uint64_t> got(n_threads, 0);
  uint64_t rows = 0;
  // (a worker that throws -- std::bad_alloc from a growing row buffer -- must not take the process down: the exception is
  // kept, every thread is joined -- also when starting one fails --, and it is rethrown here, where the C-ABI turns it
  // into an error code)
  std::vector<std::exception_ptr> failed(n_threads);
  struct Joiner {
    std::vector<std::thread>& p;
    ~Joiner() {
      for (auto& th : p)
        if (th.joinable()) th.join();
    }
  };
  for (uint64_t base = 0; base < n; base += (uint64_t)n_threads * kBlockRows) {
    std::vector<std::thread> pool;
    {
      Joiner joiner{pool};
      for (unsigned t = 0; t < n_threads; ++t) {
        const uint64_t lo = std::min(n, base + (uint64_t)t * kBlockRows), hi = std::min(n, lo + kBlockRows);
        text[t].clear();
        got[t] = 0;
        if (lo >= hi) continue;
        if (n_threads == 1) got[t] = format_rows(a, lo, hi, text[t]);
        else
          pool.emplace_back([&a, &text, &got, &failed, t, lo, hi] {
            try {
              got[t] = format_rows(a, lo, hi, text[t]);
            } catch (...) {
              failed[t] = std::current_exception();
            }
          });
      }
    }
    for (unsigned t = 0; t < n_threads; ++t)
      if (failed[t]) {
        std::fclose(f);
        std::rethrow_exception(failed[t]);
      }
    for (unsigned t = 0; t < n_threads; ++t) {
      if (got[t] == ~0ull) {
        std::fclose(f);
        throw std::runtime_error("write_read_table: entry index out of range for its pass");
      }
      if (!text[t].empty() && std::fwrite(text[t].data(), 1, text[t].size(), f) != text[t].size()) {
        std::fclose(f);
        throw std::runtime_error(std::string("short write to ") + path);
      }
      rows += got[t];
    }
  }
  if (std::fclose(f) != 0) throw std::runtime_error(std::string("cannot close ") + path);
  return rows;
}

}  // namespace mrg
