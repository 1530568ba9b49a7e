// Kernel parameter blocks and launch entry points shared by kernels.hip and
// capi.hip (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrg {

// Match workgroups are 1024 threads: 16 waves share one staged copy of the
// library (or, for HBM-served libraries, of its superblock table) and one
// survivor ring; two workgroups per CU give the full 32 waves.
template <bool LDSI>
struct MatchBlock {
  static constexpr uint32_t kThreads = 1024u;
};
constexpr uint32_t kTallyThreads = 1024u;
// Survivor staging ring per match workgroup (a power of two >= 2 * threads) +
// control words (2 x kMaxWaves per-wave counts, 1 reserved global base, padding).
constexpr uint32_t kStageCapMax = 4096u;
constexpr uint32_t kStageCapMin = 2048u;
constexpr uint32_t kMaxWaves = 16u;
constexpr uint32_t stage_bytes(uint32_t cap) { return (cap + 2u * kMaxWaves + 4u) * 4u; }

struct MatchParams {
  // library (device pointers)
  const uint32_t* blocks;  // 16 B per 32 BWT symbols
  const uint32_t* super;   // 16 B per 65536 BWT symbols
  const uint32_t* text;
  const uint64_t* sa;      // 8 B rows: pos | before<<32 | after<<40 | seg<<48
  const uint32_t* ftab;    // k-mer jump table: lo, hi per k-mer
  uint32_t ftab_k;         // 0 = do not use it
  const uint32_t* seg_start;
  const uint32_t* seg_ref;
  const uint32_t* seg_off;
  const uint32_t* chunk_seg;
  uint32_t n, nblk, nsup, primary, text_words;
  uint32_t simple_segs;  // every entry is one N-free segment: seg == entry, offset 0
  uint32_t stage_cap;    // survivor ring entries (power of two)
  // reads
  const uint64_t* reads;
  const uint8_t* lens;
  const uint64_t* nmask;  // may be null
  uint32_t n_total;       // SoA stride and identity list length
  const uint32_t* idx_in; // null = identity list of n_total reads
  const uint32_t* n_in;
  uint32_t* idx_out;      // null on the last pass
  uint32_t* n_out;
  // outputs
  int8_t* pass_id;
  int32_t* ref_id;
  int32_t* pos;
  uint8_t* mm;
  uint64_t* counters;  // processed, aligned, steps, candidates, lookups of this pass
  // policy
  int32_t seed_len, max_mm_seed, max_mm_total, trim5, trim3, min_len, max_len, poly_t;
  int32_t pass_index;
  uint32_t wstop;
};

struct TallyParams {
  const int8_t* pass_id;
  const int32_t* ref_id;
  const uint32_t* quant;
  uint64_t n;
  uint32_t n_samples, n_mirna, n_pass;
  int32_t canon_pass, isomir_pass;
  uint64_t* counts;
};

// lds_mode: 0 = index in HBM/L2, 1 = occ blocks in LDS, 2 = occ blocks + text in LDS
hipError_t launch_match(const MatchParams& p, uint32_t words_per_read, int lds_mode,
                        uint32_t grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t launch_tally(const TallyParams& p, bool lds_hist, uint32_t grid,
                        uint32_t lds_bytes, hipStream_t stream);
hipError_t launch_export_pass_counts(const uint64_t* stats, uint32_t n_pass, uint64_t* out,
                                     hipStream_t stream);

}  // namespace mrg
