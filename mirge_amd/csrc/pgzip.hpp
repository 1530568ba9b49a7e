// Parallel inflate of ONE gzip stream (internal header); see pgzip.cpp.
//
// Role in the reference: miRge2.0 takes `.fastq.gz` samples (parseArgument.py:32, __main__.py:289-314) and
// hands them to cutadapt's reader, one `gzip` stream inflated by one process.  A deflate stream has no
// index, so a second thread cannot know where a block starts or what the 32 KB window before it holds;
// this reader finds block starts by trial (a dynamic-Huffman header that decodes to a complete code is
// one in ~10^12 at a random bit position), inflates every chunk of the file with the unknown window kept
// symbolic (16-bit symbols: a literal, or "byte k of the window in front of this chunk"), and resolves
// the symbols once the chunk in front is done.  The output is the byte stream `gzread` would return.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

namespace mrg {

class GzipReader {
 public:
  // threads <= 1, a plain (not gzip) file or a file of less than two chunks: zlib's gzread.
  // chunk_bytes: compressed bytes per chunk (0 = default 1 MiB, less for small files; tests use small ones).
  GzipReader(const std::string& path, int threads, size_t chunk_bytes = 0);
  ~GzipReader();
  GzipReader(const GzipReader&) = delete;
  GzipReader& operator=(const GzipReader&) = delete;
  // Next bytes of the inflated stream, in order; 0 = end of file.  Throws std::runtime_error.
  size_t read(char* dst, size_t n);
  bool parallel() const;                // false: the zlib path is serving this file
  uint64_t chunks_merged() const;       // chunk starts that turned out not to be block starts (diagnostics)

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace mrg
