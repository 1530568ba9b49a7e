// Device-wide primitives of this library's own (gfx950, wave64): prefix sums and a stable LSD radix sort.
// Internal header (collapse.hip, pairs.hip, ingest.hip); everything is asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrg {
namespace prims {

// Bytes of device scratch a scan of n elements needs (block sums of every level).
size_t scan_temp_bytes(uint64_t n);
// out[i] = sum of in[0 .. i) (exclusive) or in[0 .. i] (inclusive), uint32 in, uint32 or uint64 sums; in == out is
// allowed.  tmp: scan_temp_bytes(n) bytes.
hipError_t exclusive_sum_u32(const uint32_t* in, uint32_t* out, uint64_t n, void* tmp, hipStream_t stream);
hipError_t inclusive_sum_u32(const uint32_t* in, uint32_t* out, uint64_t n, void* tmp, hipStream_t stream);
hipError_t exclusive_sum_u32_to_u64(const uint32_t* in, uint64_t* out, uint64_t n, void* tmp, hipStream_t stream);

// Stable least-significant-digit radix sort of (key, value) pairs over key bits [0, bits), eight bits per pass:
// a pass = per-tile digit histogram, one prefix sum over (digit, tile), stable scatter (ranks inside a tile by
// wave ballots).  Ping-pongs between the two buffer pairs; *result_in_second says where the sorted pairs are.
// vals may be null (keys only).  tmp: radix_temp_bytes(n) bytes.
size_t radix_temp_bytes(uint64_t n);
hipError_t radix_sort_pairs_u64(uint64_t* keys0, uint64_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, uint32_t bits, void* tmp,
                                hipStream_t stream, bool* result_in_second);
hipError_t radix_sort_pairs_u32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, uint32_t bits, void* tmp,
                                hipStream_t stream, bool* result_in_second);

}  // namespace prims
}  // namespace mrg
