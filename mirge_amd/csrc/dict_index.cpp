// Host build of the exact-match dictionary (dict_index.hpp).
#include "dict_index.hpp"

#include <algorithm>
#include <stdexcept>
#include <exception>
#include <thread>

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

// Inserts text position p of segment sg (home slot `home`); returns 1 when the home's chain overflowed
// by this insertion, counts a stored key in n_keys.
inline uint32_t insert_position(const FmIndex& ix, ExactDict& out, uint32_t smask, uint32_t sg, uint32_t s0, uint32_t s1, uint32_t p,
                                uint64_t win, uint32_t home, uint64_t& n_keys);

// one attempt at a given table size; returns the number of overflowed home slots.
// threads > 1: worker t owns the home slots [t, t + 1) * n_slots / threads and inserts, in text order, the
// positions homed there whose chain cannot leave the range; the positions homed within a chain's reach of a
// range's end are inserted afterwards, serially.  Positions with the same key share a home, so they stay
// in text order among themselves either way -- which is all the "first match = lowest (entry, offset)"
// rule needs.
uint64_t fill(const FmIndex& ix, uint32_t key_bases, uint32_t log2_slots, ExactDict& out, uint32_t threads) {
  const uint32_t n_slots = 1u << log2_slots, smask = n_slots - 1u;
  out.slots.assign(n_slots, DictSlot{0, 0, 0});
  out.n_keys = 0;
  const uint32_t nseg = (uint32_t)ix.seg_ref.size();
  if (threads > 1 && n_slots / threads > 4u * kDictChainOverflow) {
    struct Late {
      uint32_t sg, p;
    };
    std::vector<std::vector<Late>> late(threads);
    std::vector<uint64_t> keys(threads, 0), over(threads, 0);
    auto work = [&](uint32_t t) {
      const uint64_t lo = (uint64_t)n_slots * t / threads, hi = (uint64_t)n_slots * (t + 1) / threads;
      for (uint32_t sg = 0; sg < nseg; ++sg) {
        const uint32_t s0 = ix.seg_start[sg], s1 = ix.seg_start[sg + 1];
        for (uint32_t p = s0; p + key_bases <= s1; ++p) {
          const uint64_t win = text_window(ix, p);
          const uint32_t home = home_of(key_of(win, key_bases), log2_slots);
          if (home < lo || home >= hi) continue;
          if ((uint64_t)home + kDictChainOverflow >= hi) {
            late[t].push_back(Late{sg, p});
            continue;
          }
          over[t] += insert_position(ix, out, smask, sg, s0, s1, p, win, home, keys[t]);
        }
      }
    };
    // (a worker that throws -- std::bad_alloc from late[t].push_back -- is caught where it runs and rethrown here, after
    // every thread is joined, also when starting one fails)
    std::vector<std::exception_ptr> failed(threads);
    auto guarded = [&](uint32_t t) {
      try {
        work(t);
      } catch (...) {
        failed[t] = std::current_exception();
      }
    };
    {
      std::vector<std::thread> pool;
      struct Joiner {
        std::vector<std::thread>& p;
        ~Joiner() {
          for (auto& th : p)
            if (th.joinable()) th.join();
        }
      } joiner{pool};
      for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(guarded, t);
      guarded(0);
    }
    for (uint32_t t = 0; t < threads; ++t)
      if (failed[t]) std::rethrow_exception(failed[t]);
    uint64_t overflow = 0;
    for (uint32_t t = 0; t < threads; ++t) {
      overflow += over[t];
      out.n_keys += keys[t];
    }
    for (uint32_t t = 0; t < threads; ++t)
      for (const Late& e : late[t]) {
        const uint32_t s0 = ix.seg_start[e.sg], s1 = ix.seg_start[e.sg + 1];
        const uint64_t win = text_window(ix, e.p);
        overflow += insert_position(ix, out, smask, e.sg, s0, s1, e.p, win, home_of(key_of(win, key_bases), log2_slots), out.n_keys);
      }
    return overflow;
  }
  uint64_t overflow = 0;
  for (uint32_t sg = 0; sg < nseg; ++sg) {
    const uint32_t s0 = ix.seg_start[sg], s1 = ix.seg_start[sg + 1];
    for (uint32_t p = s0; p + key_bases <= s1; ++p) {
      const uint64_t win = text_window(ix, p);
      overflow += insert_position(ix, out, smask, sg, s0, s1, p, win, home_of(key_of(win, key_bases), log2_slots), out.n_keys);
    }
  }
  return overflow;
}

inline uint32_t insert_position(const FmIndex& ix, ExactDict& out, uint32_t smask, uint32_t sg, uint32_t s0, uint32_t s1, uint32_t p,
                                uint64_t win, uint32_t home, uint64_t& n_keys) {
  const uint32_t after = std::min<uint32_t>(63u, s1 - p), room = std::min<uint32_t>(32u, after);
  DictSlot& hs = out.slots[home];
  if (((hs.meta >> kDictChainShift) & kDictChainMask) == kDictChainOverflow) return 0u;  // this home is served by the FM index
  bool placed = false;
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
    // an earlier position matches whatever this one could
    if (s.win == win && std::min<uint32_t>(32u, s.meta & kDictAfterMask) >= room) return 0u;
  }
  if (!placed) {
    hs.meta = (hs.meta & ~(kDictChainMask << kDictChainShift)) | (kDictChainOverflow << kDictChainShift);
    return 1u;
  }
  const uint32_t chain = (hs.meta >> kDictChainShift) & kDictChainMask;
  if (d > chain) hs.meta = (hs.meta & ~(kDictChainMask << kDictChainShift)) | (d << kDictChainShift);
  ++n_keys;
  return 0u;
}

}  // namespace

namespace {
// slots the first attempt uses: twice the positions, rounded up to a power of two (0 = the library cannot have a dictionary)
uint32_t first_log2_slots(const FmIndex& ix, uint32_t key_bases) {
  if (key_bases < 8 || key_bases > 16 || ix.n > kDictMaxBases) return 0u;
  for (uint32_t v : ix.ref_len)
    if (v >= kDictMaxOffset) return 0u;
  uint64_t n_pos = 0;
  for (size_t sg = 0; sg + 1 < ix.seg_start.size(); ++sg) {
    const uint32_t len = ix.seg_start[sg + 1] - ix.seg_start[sg];
    if (len >= key_bases) n_pos += len - key_bases + 1;
  }
  uint32_t log2_slots = 10;
  while ((1ull << log2_slots) < 2 * n_pos) ++log2_slots;
  return log2_slots <= 31u ? log2_slots : 0u;
}
}  // namespace

uint64_t exact_dict_bytes(const FmIndex& ix, uint32_t key_bases) {
  const uint32_t l2 = first_log2_slots(ix, key_bases);
  return l2 ? (sizeof(DictSlot) << l2) : 0ull;
}

void build_exact_dict(const FmIndex& ix, uint32_t key_bases, ExactDict& out, uint32_t threads) {
  out = ExactDict();
  if (key_bases < 8 || key_bases > 16) throw std::runtime_error("exact dictionary: key length must be 8..16 bases");
  if (ix.n > kDictMaxBases) throw std::runtime_error("exact dictionary: library too large");
  for (uint32_t v : ix.ref_len)
    if (v >= kDictMaxOffset) throw std::runtime_error("exact dictionary: entry too long");
  uint32_t log2_slots = first_log2_slots(ix, key_bases);
  if (!log2_slots) throw std::runtime_error("exact dictionary: library too large");
  const bool big = ix.n > kDictSmallBases;
  if (!big) threads = 1;
  else if (!threads) threads = std::min<uint32_t>(std::max<uint32_t>(std::thread::hardware_concurrency(), 1u), 64u);
  // a repeat-rich library chains more than a random one: up to two doublings before positions
  // are handed to the FM index (a large library's table is gigabytes already: it is not doubled)
  for (int attempt = 0;; ++attempt) {
    const uint64_t overflow = fill(ix, key_bases, log2_slots, out, threads);
    out.n_overflow = overflow;
    if (overflow == 0 || attempt == 2 || log2_slots >= 26 || big) break;
    ++log2_slots;
  }
  out.key_bases = key_bases;
  out.log2_slots = log2_slots;
}

}  // namespace mrg
