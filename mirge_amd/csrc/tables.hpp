// Streaming writers of mapped.csv / unmapped.csv from columnar arrays (internal header).
#pragma once
#include <cstdint>

namespace mrg {

// Returns the number of rows written; throws std::runtime_error on I/O errors.
uint64_t write_read_table(const char* path, bool mapped, const char* header, bool append, const uint64_t* reads,
                          uint32_t W, uint64_t stride, const uint8_t* lens, const uint64_t* nmask, uint64_t n,
                          const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant, uint32_t n_samples,
                          uint32_t n_slots, const char* const* names, const uint64_t* names_off);

}  // namespace mrg
