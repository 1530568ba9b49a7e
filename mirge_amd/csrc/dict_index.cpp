// Host build of the exact-match dictionary (dict_index.hpp).
#include "dict_index.hpp"

#include <algorithm>
#include <stdexcept>

namespace mrg {

namespace {

// 32 bases starting at text position q (first base in the low two bits); bases past the end of the
// text read as A (the text array is zero padded)
inline uint64_t text_window(const FmIndex& ix, uint64_t q) {
  const size_t w = q >> 4;
  const uint32_t sh = (uint32_t)(q & 15) * 2;
  const size_t nw = ix.text.size();
  const uint64_t t0 = w < nw ? ix.text[w] : 0, t1 = w + 1 < nw ? ix.text[w + 1] : 0, t2 = w + 2 < nw ? ix.text[w + 2] : 0;
  const uint64_t lo = t0 | (t1 << 32);
  uint64_t win = sh ? (lo >> sh) | (t2 << (64 - sh)) : lo;
  if (q + 32 > ix.n) {
    const uint64_t have = q < ix.n ? ix.n - q : 0;
    win &= have ? ((have >= 32 ? ~0ull : ((1ull << (2 * have)) - 1ull))) : 0ull;
  }
  return win;
}

inline uint32_t key_of(uint64_t win, uint32_t key_bases) {
  return (uint32_t)(key_bases >= 16 ? win : (win & ((1ull << (2 * key_bases)) - 1ull)));
}

inline uint32_t home_of(uint32_t key, uint32_t log2_slots) { return (uint32_t)(key * kDictHashMul) >> (32u - log2_slots); }

// one attempt at a given table size; returns the number of overflowed home slots
uint64_t fill(const FmIndex& ix, uint32_t key_bases, uint32_t log2_slots, ExactDict& out) {
  const uint32_t n_slots = 1u << log2_slots, smask = n_slots - 1u;
  out.slots.assign(n_slots, DictSlot{0, 0, 0});
  out.n_keys = 0;
  uint64_t overflow = 0;
  const uint32_t nseg = (uint32_t)ix.seg_ref.size();
  for (uint32_t sg = 0; sg < nseg; ++sg) {
    const uint32_t s0 = ix.seg_start[sg], s1 = ix.seg_start[sg + 1];
    for (uint32_t p = s0; p + key_bases <= s1; ++p) {
      const uint64_t win = text_window(ix, p);
      const uint32_t after = std::min<uint32_t>(63u, s1 - p), room = std::min<uint32_t>(32u, after);
      const uint32_t home = home_of(key_of(win, key_bases), log2_slots);
      DictSlot& hs = out.slots[home];
      if (((hs.meta >> kDictChainShift) & kDictChainMask) == kDictChainOverflow) continue;  // this home is served by the FM index
      bool placed = false, dominated = false;
      uint32_t d = 0;
      for (; d < kDictChainOverflow; ++d) {
        DictSlot& s = out.slots[(home + d) & smask];
        if (!(s.meta & kDictOccBit)) {
          const uint32_t keep = s.meta & (kDictChainMask << kDictChainShift);  // this slot's own chain field
          s.win = win;
          s.ref = ix.seg_ref[sg];
          s.meta = keep | after | kDictOccBit | ((ix.seg_off[sg] + (p - s0)) << kDictOffShift);
          placed = true;
          break;
        }
        if (s.win == win && std::min<uint32_t>(32u, s.meta & kDictAfterMask) >= room) {
          dominated = true;  // an earlier position matches whatever this one could
          break;
        }
      }
      if (dominated) continue;
      if (!placed) {
        hs.meta = (hs.meta & ~(kDictChainMask << kDictChainShift)) | (kDictChainOverflow << kDictChainShift);
        ++overflow;
        continue;
      }
      const uint32_t chain = (hs.meta >> kDictChainShift) & kDictChainMask;
      if (d > chain) hs.meta = (hs.meta & ~(kDictChainMask << kDictChainShift)) | (d << kDictChainShift);
      ++out.n_keys;
    }
  }
  return overflow;
}

}  // namespace

void build_exact_dict(const FmIndex& ix, uint32_t key_bases, ExactDict& out) {
  out = ExactDict();
  if (key_bases < 8 || key_bases > 16) throw std::runtime_error("exact dictionary: key length must be 8..16 bases");
  if (ix.n > kDictMaxBases) throw std::runtime_error("exact dictionary: library too large");
  for (uint32_t v : ix.ref_len)
    if (v >= kDictMaxOffset) throw std::runtime_error("exact dictionary: entry too long");
  uint64_t n_pos = 0;
  for (size_t sg = 0; sg + 1 < ix.seg_start.size(); ++sg) {
    const uint32_t len = ix.seg_start[sg + 1] - ix.seg_start[sg];
    if (len >= key_bases) n_pos += len - key_bases + 1;
  }
  uint32_t log2_slots = 10;
  while ((1ull << log2_slots) < 2 * n_pos) ++log2_slots;
  // a repeat-rich library chains more than a random one: up to two doublings before positions
  // are handed to the FM index
  for (int attempt = 0;; ++attempt) {
    const uint64_t overflow = fill(ix, key_bases, log2_slots, out);
    out.n_overflow = overflow;
    if (overflow == 0 || attempt == 2 || log2_slots >= 26) break;
    ++log2_slots;
  }
  out.key_bases = key_bases;
  out.log2_slots = log2_slots;
}

}  // namespace mrg
