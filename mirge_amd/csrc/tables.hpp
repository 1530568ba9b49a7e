// Streaming writers of mapped.csv / unmapped.csv from columnar arrays (internal header).
#pragma once
#include <cstdint>

namespace mrg {

// Returns the number of rows written; throws std::runtime_error on I/O errors.
uint64_t write_read_table(const char* path, bool mapped, const char* header, bool append, const uint64_t* reads,
                          uint32_t W, uint64_t stride, const uint8_t* lens, const uint64_t* nmask, uint64_t n,
                          const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant, uint32_t n_samples,
                          uint32_t n_slots, const char* const* names, const uint64_t* names_off);

// isomirs.csv + isomirs.samples.csv (writeDataToCSV.py:1090-1170) from the arrays: reads claimed by canon_pass / isomir_pass
// grouped by group_of_entry[ref] (the miRNA name with its SNP suffix stripped), groups in order of first appearance.
// Returns the rows of isomirs.csv; throws std::runtime_error.
uint64_t write_isomir_tables(const char* isomirs_path, const char* samples_path, const char* header1, const char* header2,
                             const uint64_t* reads, uint32_t W, uint64_t stride, const uint8_t* lens, const uint64_t* nmask,
                             uint64_t n, const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant, uint32_t S,
                             int32_t canon_pass, int32_t isomir_pass, const int32_t* group_of_entry, uint64_t n_entries,
                             const char* const* group_names, uint32_t n_groups, const double* filtered);

}  // namespace mrg
