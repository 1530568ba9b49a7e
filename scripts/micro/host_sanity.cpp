// Host-side threaded code under the sanitizers (the GPU side has none on this pool): index build + parallel dictionary fill,
// the FASTQ loader (plain and gzip, several threads), the parallel table writer.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined  (or -fsanitize=thread)  -I mirge_amd/csrc scripts/micro/host_sanity.cpp \
//       mirge_amd/csrc/{fm_index,dict_index,fastq,pgzip,tables}.cpp -o /tmp/host_sanity -lz -pthread -mpclmul -msse4.1
//   /tmp/host_sanity file.fastq file.fastq.gz
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "dict_index.hpp"
#include "fastq.hpp"
#include "fm_index.hpp"
#include "tables.hpp"

int main(int argc, char** argv) {
  std::mt19937_64 rng(7);
  // ---- an index of 5 Mbp (beyond kDictSmallBases: the parallel fill) with repeats and N runs ----
  std::vector<std::string> names, seqs;
  std::string unit;
  for (int i = 0; i < 300; ++i) unit.push_back("ACGT"[rng() & 3]);
  for (int e = 0; e < 40; ++e) {
    std::string s;
    for (int i = 0; i < 125000; ++i) s.push_back("ACGT"[rng() & 3]);
    s.replace(1000 + 37 * e, unit.size(), unit);
    if (e % 7 == 0) s.replace(60000, 5, "NNNNN");
    names.push_back("e" + std::to_string(e));
    seqs.push_back(s);
  }
  mrg::FmIndex ix;
  mrg::build_index(names, seqs, ix);
  mrg::ExactDict d1, d8;
  mrg::build_exact_dict(ix, 16, d1, 1);
  mrg::build_exact_dict(ix, 16, d8, 8);
  printf("index %u bases; dictionary: %llu keys (1 thread), %llu keys (8 threads), slots 2^%u\n", ix.n, (unsigned long long)d1.n_keys,
         (unsigned long long)d8.n_keys, d8.log2_slots);
  if (d1.n_keys != d8.n_keys || d1.log2_slots != d8.log2_slots) return 1;
  // ---- the FASTQ loader ----
  for (int a = 1; a < argc; ++a)
    for (int threads : {1, 6}) {
      mrg::FastqData fq;
      mrg::load_fastq(argv[a], 10, 16, "TGGAATTCTCGGGTGCCAAGG", threads, fq);
      printf("%s, %d threads: %llu records, %llu kept, %u words\n", argv[a], threads, (unsigned long long)fq.n_total,
             (unsigned long long)fq.n_kept, fq.words_per_read);
    }
  // ---- the table writer: 600 000 rows, two words per read ----
  const uint64_t n = 600000;
  std::vector<uint64_t> reads(2 * n), nm(2 * n, 0);
  std::vector<uint8_t> lens(n);
  std::vector<int8_t> pass(n);
  std::vector<int32_t> ref(n);
  std::vector<uint32_t> quant(2 * n);
  for (uint64_t i = 0; i < n; ++i) {
    reads[i] = rng();
    reads[n + i] = rng();
    lens[i] = (uint8_t)(16 + rng() % 40);
    pass[i] = (int8_t)((int)(rng() % 4) - 1);
    ref[i] = pass[i] >= 0 ? (int32_t)(rng() % 5) : -1;
    quant[2 * i] = (uint32_t)(rng() % 100);
    quant[2 * i + 1] = (uint32_t)(rng() % 100);
  }
  std::vector<std::string> nm_s;
  for (int i = 0; i < 15; ++i) nm_s.push_back("name" + std::to_string(i));
  std::vector<const char*> name_ptr;
  for (auto& s : nm_s) name_ptr.push_back(s.c_str());
  const uint64_t off[4] = {0, 5, 10, 15};
  const uint64_t rows = mrg::write_read_table("/tmp/host_sanity_mapped.csv", true, "h\n", false, reads.data(), 2, n, lens.data(), nm.data(), n,
                                              pass.data(), ref.data(), quant.data(), 2, 3, name_ptr.data(), off);
  printf("table: %llu rows\n", (unsigned long long)rows);
  return 0;
}
