// Inflate rate of csrc/pgzip.cpp alone (no FASTQ parsing): ./pgzip_rate file.gz threads [request_bytes]
//   g++ -O2 -std=c++17 -I mirge_amd/csrc scripts/micro/pgzip_rate.cpp mirge_amd/csrc/pgzip.cpp -o /tmp/pgzip_rate -lz -pthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pgzip.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int threads = atoi(argv[2]);
  const size_t req = argc > 3 ? (size_t)atol(argv[3]) : (64u << 20);
  const auto t0 = std::chrono::steady_clock::now();
  mrg::GzipReader r(argv[1], threads);
  std::vector<char> buf(req);
  size_t total = 0;
  for (;;) {
    const size_t got = r.read(buf.data(), buf.size());
    if (!got) break;
    total += got;
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("threads %d request %zu: %zu bytes in %.3f s = %.2f GB/s (parallel %d, merged %llu)\n", threads, req, total, s, total / s / 1e9,
         (int)r.parallel(), (unsigned long long)r.chunks_merged());
  return 0;
}
