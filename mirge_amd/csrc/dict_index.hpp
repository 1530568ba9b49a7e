// Exact-match dictionary of one library (internal header; host build, uploaded verbatim).
//
// Role in the reference: the `-n 0` / `-v 0` bowtie runs (runAnnotationPipeline.py:577, :598, :688)
// on reads no longer than the seed (28 nt) ask one question: "does this read occur, letter for
// letter, inside an entry of the library" -- a dictionary lookup, not a search.  The FM index
// answers it in three dependent memory trips (jump table, suffix-array row, text window); this
// table answers it in ONE 16-byte load for nearly every read, hit or miss.
//
// Open addressing over 16-byte slots, keyed by the first `key_bases` bases at a text position:
//   win   the 32 text bases starting at the position (2 bits per base, first base lowest; bases past
//         the end of the text read as A), so the rest of the read is compared from the slot itself
//   ref   library entry of the position
//   meta  bits 0-5   bases from the position to the end of its N-free segment (clamped to 63)
//         bits 6-9   chain: how many slots after its HOME slot a key homed HERE may sit in
//                    (15 = more than the table can chain: the reader falls back to the FM index)
//         bit  10    slot occupied
//         bits 11-31 offset of the position inside its entry
// Positions are inserted in text order and a lookup walks home, home + 1, ... home + chain: the
// first slot that matches is the lowest (entry, offset) -- the tie rule of mrg_cascade_run.  A
// position whose window and segment room equal those of an earlier one can never win and is left
// out (paralogous miRNA entries with identical mature sequences).
#pragma once
#include <cstdint>
#include <vector>

#include "fm_index.hpp"

namespace mrg {

struct DictSlot {
  uint64_t win;
  uint32_t ref;
  uint32_t meta;
};
static_assert(sizeof(DictSlot) == 16, "dictionary slot must be 16 bytes");

constexpr uint32_t kDictAfterMask = 63u, kDictChainShift = 6u, kDictChainMask = 15u, kDictOccBit = 1u << 10, kDictOffShift = 11u;
constexpr uint32_t kDictChainOverflow = 15u;
constexpr uint32_t kDictSmallBases = 1u << 22;  // "small" libraries (serial build; their entries are also kept for seed units)
constexpr uint32_t kDictMaxBases = 1u << 30;    // libraries up to this size can have a dictionary (16 B x 2..4 slots per base of HBM)
constexpr uint32_t kDictMaxOffset = 1u << 21;   // entry offsets must fit 21 bits
constexpr uint32_t kDictHashMul = 0x9E3779B1u;  // slot = (key * mul) >> (32 - log2_slots)

struct ExactDict {
  uint32_t key_bases = 0;   // 0 = not built
  uint32_t log2_slots = 0;
  uint64_t n_keys = 0;      // positions stored
  uint64_t n_overflow = 0;  // home slots whose chain overflowed (their keys are served by the FM index)
  std::vector<DictSlot> slots;
};

// Throws std::runtime_error when the library cannot have one (too large, entry offsets too wide).
// Libraries beyond kDictSmallBases are filled by `threads` workers (0 = one per hardware thread), each
// owning a range of home slots; the layout then depends on the worker count, what a lookup finds does not.
void build_exact_dict(const FmIndex& ix, uint32_t key_bases, ExactDict& out, uint32_t threads = 0);
// Bytes of the slot array build_exact_dict would start with (before any doubling), 0 = no dictionary possible.
uint64_t exact_dict_bytes(const FmIndex& ix, uint32_t key_bases);

}  // namespace mrg
