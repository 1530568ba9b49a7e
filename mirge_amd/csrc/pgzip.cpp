// Parallel inflate of one gzip stream (pgzip.hpp).
//
// What it replaces in the reference: the single `gzip` stream behind cutadapt's reader for `.fastq.gz`
// samples (parseArgument.py:32, __main__.py:289-314, trim_file.py:89-134): one inflate thread fed every
// trimming worker, 10 M reads/s whatever the core count.
//
// How (the published two-stage scheme of pugz / rapidgzip, restated; deflate itself is RFC 1951):
//   1. the compressed file is cut into chunks of `chunk_bytes`;
//   2. FIND: in every chunk but the first, the first bit position that starts a dynamic-Huffman block --
//      BFINAL = 0, BTYPE = 2, HLIT / HDIST in range, a COMPLETE code-length code, literal/length and
//      distance codes that are complete too and contain the end-of-block symbol, and a few hundred
//      symbols that decode to text-like literals and legal distances;
//   3. DECODE: every chunk is inflated from its start to the next chunk's start with a symbolic window:
//      the output is 16-bit symbols, a literal byte or 0x8000 | k = "byte k of the 32 KB in front of
//      this chunk" (copies copy symbols, so a reference to a reference resolves by itself);
//   4. WINDOW / RESOLVE: once the chunk in front has its last 32 KB as bytes, a chunk's symbols become
//      bytes (the last 32 KB first, so the chain along the file costs 32 KB per chunk);
//   5. the reader hands the chunks out in order and checks every member's CRC-32 and length.
// A start that FIND got wrong (the decoder of the chunk in front runs past it without landing on it) is
// dropped and its chunk merged into the one in front; a file the scheme cannot handle at all (no block
// start found anywhere: stored or fixed-Huffman blocks only) degenerates to one thread, never to a wrong
// byte.  Any inconsistency (bad code, distance beyond the window, CRC) is an error, as with gzread.
#include "pgzip.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

namespace mrg {

namespace {

constexpr uint32_t kWindow = 32768u;
constexpr uint64_t kNoStart = ~0ull;

// ------------------------------------------------------------------ bits
struct BitSrc {
  const uint8_t* data = nullptr;
  uint64_t n_bits = 0;  // size of the file in bits
  // >= 57 bits from bit position pos, LSB first (zeros past the end of the file)
  inline uint64_t peek(uint64_t pos) const {
    const uint64_t byte = pos >> 3;
    uint64_t v = 0;
    const uint64_t n_bytes = n_bits >> 3;
    if (byte + 8 <= n_bytes) std::memcpy(&v, data + byte, 8);
    else if (byte < n_bytes) std::memcpy(&v, data + byte, (size_t)(n_bytes - byte));
    return v >> (pos & 7);
  }
};

// ------------------------------------------------------------------ Huffman tables
// Two-level lookup.  Entry: bit 31 = link to a second-level table (bits 4..30 its offset, bits 0..3 its
// index width), else bits 4..30 = symbol, bits 0..3 = code length (0 = no such code).
struct Huff {
  static constexpr uint32_t kPrim = 10;
  std::vector<uint32_t> tab;  // primary (1 << kPrim) then second-level tables
  uint32_t max_len = 0;
  // lens[n]: code length per symbol (0 = unused).  Returns: 0 complete, 1 incomplete, -1 over-subscribed / empty.
  int build(const uint8_t* lens, uint32_t n) {
    uint32_t count[16] = {0};
    for (uint32_t i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    max_len = 0;
    int64_t left = 1;
    uint32_t total = 0;
    for (uint32_t l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= count[l];
      if (left < 0) return -1;
      if (count[l]) max_len = l;
      total += count[l];
    }
    if (!total) return -1;
    uint32_t next[16];
    {
      uint32_t code = 0;
      for (uint32_t l = 1; l <= 15; ++l) {
        code = (code + count[l - 1]) << 1;
        next[l] = code;
      }
    }
    tab.assign(1u << kPrim, 0u);
    // widest code behind every primary index that needs a second level
    uint8_t sub_bits[1u << kPrim];
    if (max_len > kPrim) std::memset(sub_bits, 0, sizeof sub_bits);
    std::vector<uint32_t> rev(n, 0);
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t l = lens[s];
      if (!l) continue;
      uint32_t c = next[l]++, r = 0;
      for (uint32_t b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
      rev[s] = r;
      if (l > kPrim) {
        uint8_t& sb = sub_bits[r & ((1u << kPrim) - 1u)];
        sb = std::max<uint8_t>(sb, (uint8_t)(l - kPrim));
      }
    }
    if (max_len > kPrim)
      for (uint32_t i = 0; i < (1u << kPrim); ++i)
        if (sub_bits[i]) {
          tab[i] = 0x80000000u | ((uint32_t)tab.size() << 4) | sub_bits[i];
          tab.resize(tab.size() + (1u << sub_bits[i]), 0u);
        }
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t l = lens[s];
      if (!l) continue;
      const uint32_t r = rev[s], e = (s << 4) | l;
      if (l <= kPrim) {
        for (uint32_t i = r; i < (1u << kPrim); i += 1u << l) tab[i] = e;
      } else {
        const uint32_t link = tab[r & ((1u << kPrim) - 1u)];
        const uint32_t off = (link >> 4) & 0x7FFFFFFu, width = link & 15u;
        for (uint32_t i = r >> kPrim; i < (1u << width); i += 1u << (l - kPrim)) tab[off + i] = e;
      }
    }
    return left ? 1 : 0;
  }
  // -> symbol, consumes `len` bits (0 = invalid code)
  inline uint32_t decode(uint64_t bits, uint32_t& len) const {
    uint32_t e = tab[bits & ((1u << kPrim) - 1u)];
    if (e & 0x80000000u) e = tab[((e >> 4) & 0x7FFFFFFu) + ((bits >> kPrim) & ((1u << (e & 15u)) - 1u))];
    len = e & 15u;
    return e >> 4;
  }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BlockCodes {
  Huff lit, dist;
  bool dist_empty = false;  // no distance code at all (a block of literals only)
};

// The header of a dynamic-Huffman block whose 3 header bits start at `pos`; on success pos is behind the
// code lengths.  strict (FIND): every code must be complete.  Returns false on anything zlib rejects.
bool read_dynamic_header(const BitSrc& src, uint64_t& pos, BlockCodes& bc, bool strict) {
  if (pos + 17 > src.n_bits) return false;
  uint64_t bits = src.peek(pos);
  const uint32_t hlit = (uint32_t)(bits & 31u) + 257u, hdist = (uint32_t)((bits >> 5) & 31u) + 1u, hclen = (uint32_t)((bits >> 10) & 15u) + 4u;
  if (hlit > 286u || hdist > 30u) return false;
  pos += 14;
  if (pos + 3ull * hclen > src.n_bits) return false;
  uint8_t cl[19] = {0};
  bits = src.peek(pos);
  for (uint32_t i = 0; i < hclen; ++i) cl[kClOrder[i]] = (uint8_t)((bits >> (3u * i)) & 7u);
  pos += 3ull * hclen;
  Huff clh;
  if (clh.build(cl, 19) != 0) return false;  // (zlib: the code-length code must be complete)
  uint8_t lens[286 + 30];
  uint32_t i = 0;
  const uint32_t n = hlit + hdist;
  while (i < n) {
    if (pos + 14 > src.n_bits) return false;
    bits = src.peek(pos);
    uint32_t len;
    const uint32_t sym = clh.decode(bits, len);
    if (!len) return false;
    pos += len;
    bits >>= len;
    if (sym < 16u) {
      lens[i++] = (uint8_t)sym;
      continue;
    }
    uint32_t rep, val = 0;
    if (sym == 16u) {
      if (!i) return false;
      val = lens[i - 1];
      rep = 3u + (uint32_t)(bits & 3u);
      pos += 2;
    } else if (sym == 17u) {
      rep = 3u + (uint32_t)(bits & 7u);
      pos += 3;
    } else {
      rep = 11u + (uint32_t)(bits & 127u);
      pos += 7;
    }
    if (i + rep > n) return false;
    while (rep--) lens[i++] = (uint8_t)val;
  }
  if (!lens[256]) return false;  // no end-of-block code
  const int rl = bc.lit.build(lens, hlit);
  if (rl < 0 || (rl > 0 && (strict || bc.lit.max_len != 1))) return false;
  bool any_dist = false;
  for (uint32_t d = 0; d < hdist; ++d) any_dist |= lens[hlit + d] != 0;
  bc.dist_empty = !any_dist;
  if (any_dist) {
    const int rd = bc.dist.build(lens + hlit, hdist);
    if (rd < 0 || (rd > 0 && bc.dist.max_len != 1)) return false;
  }
  return true;
}

const BlockCodes& fixed_codes() {
  static const BlockCodes* fc = [] {
    BlockCodes* b = new BlockCodes();
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    b->lit.build(l, 288);
    uint8_t d[30];
    for (int i = 0; i < 30; ++i) d[i] = 5;
    b->dist.build(d, 30);
    return b;
  }();
  return *fc;
}

// gzip member header at byte offset `off`: returns the offset of the deflate data, 0 = not a gzip header
size_t parse_gzip_header(const uint8_t* p, size_t n, size_t off) {
  if (off + 18 > n || p[off] != 0x1f || p[off + 1] != 0x8b || p[off + 2] != 8) return 0;
  const uint8_t flg = p[off + 3];
  size_t q = off + 10;
  if (flg & 4) {  // FEXTRA
    if (q + 2 > n) return 0;
    q += 2 + (size_t)(p[q] | (p[q + 1] << 8));
  }
  for (int f = 0; f < 2; ++f)
    if (flg & (f ? 16 : 8)) {  // FNAME, FCOMMENT
      while (q < n && p[q]) ++q;
      ++q;
    }
  if (flg & 2) q += 2;  // FHCRC
  return q < n ? q : 0;
}

// ------------------------------------------------------------------ symbols -> bytes, CRC-32
// The resolve stage was a third of the reader's CPU time (2 ns per byte: a branch per symbol, then zlib's table-driven
// crc32 over the bytes).  Markers only sit near a chunk's front, so 16 symbols at a time are packed with one SSE2
// instruction when none of them is a marker; the CRC-32 (gzip's polynomial) is folded 64 bytes at a time with carry-less
// multiplication (the published PCLMULQDQ scheme: Gopal et al., "Fast CRC Computation for Generic Polynomials Using
// PCLMULQDQ"), checked against zlib's crc32 once at start-up and left to zlib when the CPU lacks the instruction.
#if defined(__x86_64__)
#include <immintrin.h>
#define MRG_X86 1
#else
#define MRG_X86 0
#endif

// bytes[i] = sym[i] < 0x8000 ? sym[i] : window[sym[i] & 0x7FFF]; returns false when a marker meets no window
bool resolve_symbols(const uint16_t* sym, size_t n, const uint8_t* window, uint8_t* bytes) {
  size_t i = 0;
#if MRG_X86
  for (; i + 16 <= n; i += 16) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(sym + i));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(sym + i + 8));
    if ((_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA) == 0) {  // no symbol has its top bit set: sixteen literals
      _mm_storeu_si128(reinterpret_cast<__m128i*>(bytes + i), _mm_packus_epi16(a, b));
      continue;
    }
    for (size_t j = i; j < i + 16; ++j) {
      const uint16_t v = sym[j];
      if (v & 0x8000u) {
        if (!window) return false;
        bytes[j] = window[v & 0x7FFFu];
      } else {
        bytes[j] = (uint8_t)v;
      }
    }
  }
#endif
  for (; i < n; ++i) {
    const uint16_t v = sym[i];
    if (v & 0x8000u) {
      if (!window) return false;
      bytes[i] = window[v & 0x7FFFu];
    } else {
      bytes[i] = (uint8_t)v;
    }
  }
  return true;
}

#if MRG_X86
// CRC-32 (reflected, polynomial 0xEDB88320) of p[0, len), len a multiple of 16 and >= 64, continuing from `crc`
// (the raw register: pre- and post-inversion are the caller's)
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul_blocks(const uint8_t* p, size_t len, uint32_t crc) {
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
  const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
  const __m128i k5k0 = _mm_set_epi64x(0x0000000000ll, 0x0163cd6124ll);
  const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x00));
  x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x10));
  x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x20));
  x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  x0 = k1k2;
  p += 64;
  len -= 64;
  while (len >= 64) {  // four lanes of 16 bytes, folded 64 bytes ahead
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
    x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
    x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x00));
    y6 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x10));
    y7 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x20));
    y8 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    p += 64;
    len -= 64;
  }
  x0 = k3k4;  // the four lanes into one
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (len >= 16) {  // single blocks
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    p += 16;
    len -= 16;
  }
  // 128 -> 64 bits
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = k5k0;
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  // Barrett reduction 64 -> 32 bits
  x0 = poly;
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// zlib's crc32(crc, p, len) (same value), through carry-less multiplication where the CPU has it and the start-up
// check against zlib agreed
uint32_t fast_crc32(uint32_t crc, const uint8_t* p, size_t len) {
#if MRG_X86
  static const bool use_clmul = [] {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    uint8_t t[1024 + 7];
    for (size_t i = 0; i < sizeof t; ++i) t[i] = (uint8_t)(i * 131u + (i >> 3) * 7u + 5u);
    for (size_t n : {64u, 80u, 128u, 1008u, 1024u}) {
      const uint32_t want = (uint32_t)crc32(0x1234567ul, t + 3, (uInt)n);
      const uint32_t got = ~crc32_clmul_blocks(t + 3, n, ~0x1234567u);
      if (want != got) return false;
    }
    return true;
  }();
  if (use_clmul && len >= 64) {
    const size_t body = len & ~(size_t)15;
    crc = ~crc32_clmul_blocks(p, body, ~crc);
    p += body;
    len -= body;
  }
#endif
  while (len) {
    const size_t m_ = std::min<size_t>(len, 1u << 30);
    crc = (uint32_t)crc32(crc, p, (uInt)m_);
    p += m_;
    len -= m_;
  }
  return crc;
}

// ------------------------------------------------------------------ buffers
// A chunk's symbols (16 MB and more) and bytes live in raw buffers that are recycled: std::vector would zero-fill
// every growth and hand the pages back on release, and dozens of threads faulting fresh pages in and out of one
// address space serialise in the kernel (the first version fell from 22 to 17 M reads/s between 32 and 128 threads).
struct RawBuf {
  void* p = nullptr;
  size_t cap = 0;  // bytes
  void ensure(size_t bytes, size_t keep) {  // at least `bytes`; the first `keep` bytes survive
    if (bytes <= cap) return;
    size_t want = std::max(bytes, cap + cap / 2);
    want = (want + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    void* q = std::malloc(want);
    if (!q) throw std::bad_alloc();
    if (keep) std::memcpy(q, p, keep);
    std::free(p);
    p = q;
    cap = want;
  }
  void release() {
    std::free(p);
    p = nullptr;
    cap = 0;
  }
};
struct BufPool {
  std::mutex mu;
  std::vector<RawBuf> spare;
  RawBuf take() {
    std::lock_guard<std::mutex> lk(mu);
    if (spare.empty()) return RawBuf();
    RawBuf b = spare.back();
    spare.pop_back();
    return b;
  }
  void give(RawBuf& b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(mu);
    spare.push_back(b);
    b = RawBuf();
  }
  ~BufPool() {
    for (RawBuf& b : spare) b.release();
  }
};

// ------------------------------------------------------------------ one chunk
struct MemberEnd {
  uint64_t out_off;  // symbols of this chunk that belong to the member that ends here
  uint32_t crc, isize;
};

struct Chunk {
  uint64_t nominal_bit = 0;    // where its search starts
  uint64_t start_bit = kNoStart;
  uint64_t end_bit = 0;        // where its decoder stopped
  bool find_done = false, busy = false /* a find or a decode is running */, decode_done = false, windowing = false, window_done = false,
       resolving = false, resolve_done = false;
  bool dropped = false;        // start was not a block start, or none found: the chunk in front covers it
  bool at_eof = false;         // its decoder reached the end of the file
  bool clean_start = false;    // starts a member: nothing in front of it can be referenced
  RawBuf sym;        // n_sym 16-bit symbols
  size_t n_sym = 0;
  size_t sym_hint = 0;  // bytes the symbol buffer starts with (a chunk of FASTQ text inflates 4-6 x)
  RawBuf bytes;      // n_sym bytes once resolved
  std::vector<uint8_t> window;  // the last 32 KB of the stream up to and including this chunk
  std::vector<MemberEnd> ends;
  std::vector<uint32_t> seg_crc;  // crc of the bytes between member ends (ends.size() + 1 segments)
  std::string error;
};

// Inflates from `pos` (a block header) until a block boundary that is one of the later chunks' starts
// (`stop_at(bit)`: 0 = go on, 1 = stop here) or the end of the file.
template <class StopAt>
void decode_chunk(const BitSrc& src, Chunk& c, StopAt&& stop_at) {
  uint64_t pos = c.start_bit;
  RawBuf& ob = c.sym;
  ob.ensure(c.sym_hint ? c.sym_hint : (8u << 20), 0);
  uint16_t* out = static_cast<uint16_t*>(ob.p);
  size_t n_out = 0;
  size_t member_base = 0;  // symbols before the current member (within this chunk): references cannot go in front of it
  bool member_clean = c.clean_start;
  BlockCodes dyn;
  auto need = [&](size_t extra) {
    if (ob.cap < (n_out + extra) * 2) {
      ob.ensure((n_out + extra) * 2, n_out * 2);
      out = static_cast<uint16_t*>(ob.p);
    }
  };
  for (;;) {
    if (pos + 3 > src.n_bits) throw std::runtime_error("unexpected end of the deflate stream");
    uint64_t bits = src.peek(pos);
    const bool final_block = bits & 1u;
    const uint32_t type = (uint32_t)(bits >> 1) & 3u;
    pos += 3;
    if (type == 3u) throw std::runtime_error("invalid deflate block type");
    if (type == 0u) {
      pos = (pos + 7) & ~7ull;
      if (pos + 32 > src.n_bits) throw std::runtime_error("truncated stored block");
      const uint64_t b = pos >> 3;
      const uint32_t len = src.data[b] | (src.data[b + 1] << 8), nlen = src.data[b + 2] | (src.data[b + 3] << 8);
      if ((len ^ nlen) != 0xFFFFu) throw std::runtime_error("corrupt stored block");
      if (pos + 32 + 8ull * len > src.n_bits) throw std::runtime_error("truncated stored block");
      need(len);
      for (uint32_t i = 0; i < len; ++i) out[n_out + i] = src.data[b + 4 + i];
      n_out += len;
      pos += 32 + 8ull * len;
    } else {
      const BlockCodes* bc = &fixed_codes();
      if (type == 2u) {
        if (!read_dynamic_header(src, pos, dyn, false)) throw std::runtime_error("corrupt dynamic-Huffman header");
        bc = &dyn;
      }
      for (;;) {
        need(258 + 8);
        if (pos > src.n_bits) throw std::runtime_error("unexpected end of the deflate stream");
        bits = src.peek(pos);
        uint32_t len;
        uint32_t sym = bc->lit.decode(bits, len);
        if (!len) throw std::runtime_error("invalid literal/length code");
        pos += len;
        if (sym < 256u) {
          out[n_out++] = (uint16_t)sym;
          // (a second literal from the bits in hand: most of a FASTQ stream is literals)
          bits >>= len;
          sym = bc->lit.decode(bits, len);
          if (len && sym < 256u) {
            out[n_out++] = (uint16_t)sym;
            pos += len;
          }
          continue;
        }
        if (sym == 256u) break;
        sym -= 257u;
        if (sym >= 29u) throw std::runtime_error("invalid length symbol");
        bits >>= len;
        uint32_t length = kLenBase[sym] + (uint32_t)(bits & ((1u << kLenExtra[sym]) - 1u));
        pos += kLenExtra[sym];
        bits >>= kLenExtra[sym];
        if (bc->dist_empty) throw std::runtime_error("distance code in a block without distance codes");
        uint32_t dl;
        const uint32_t ds = bc->dist.decode(bits, dl);
        if (!dl || ds >= 30u) throw std::runtime_error("invalid distance code");
        bits >>= dl;
        const uint32_t dist = kDistBase[ds] + (uint32_t)(bits & ((1u << kDistExtra[ds]) - 1u));
        pos += dl + kDistExtra[ds];
        if (dist > n_out - member_base) {
          // reaches in front of this chunk's part of the member: legal only into the window of a member that
          // began before the chunk
          if (member_clean) throw std::runtime_error("distance beyond the start of the gzip member");
          if (dist - n_out > kWindow) throw std::runtime_error("distance beyond the deflate window");
        }
        // copy, symbol by symbol where the source lies in front of the chunk
        size_t i = 0;
        if (dist > n_out) {
          const size_t front = std::min<size_t>(length, dist - n_out);
          const uint32_t w0 = kWindow - (uint32_t)(dist - n_out);
          for (; i < front; ++i) out[n_out + i] = (uint16_t)(0x8000u | (w0 + (uint32_t)i));
        }
        const uint16_t* s = out + n_out - dist;
        uint16_t* d = out + n_out;
        for (; i < length; ++i) d[i] = s[i];
        n_out += length;
      }
    }
    if (final_block) {
      // gzip trailer, then maybe another member
      pos = (pos + 7) & ~7ull;
      if (pos + 64 > src.n_bits) throw std::runtime_error("truncated gzip trailer");
      const uint64_t b = pos >> 3;
      MemberEnd me;
      me.out_off = n_out;
      std::memcpy(&me.crc, src.data + b, 4);
      std::memcpy(&me.isize, src.data + b + 4, 4);
      c.ends.push_back(me);
      pos += 64;
      const size_t next = parse_gzip_header(src.data, (size_t)(src.n_bits >> 3), (size_t)(pos >> 3));
      if (!next) {  // end of the file (whatever follows a complete member is ignored, as gzread does)
        c.at_eof = true;
        break;
      }
      pos = (uint64_t)next << 3;
      member_base = n_out;
      member_clean = true;
    }
    if (pos >= src.n_bits) throw std::runtime_error("unexpected end of the deflate stream");
    if (stop_at(pos)) break;
  }
  c.n_sym = n_out;
  c.end_bit = pos;
}

// A gzip member header at a byte position of [from, to) (bit positions) whose deflate data starts with a plausible block:
// a file of many members (bgzip / BGZF: 64 KB members of ONE final block each; `cat a.gz b.gz`) has no non-final
// dynamic block to find, but every member is a chunk start -- and a CLEAN one: nothing in front of it can be referenced.
// Returns the bit position of the member's first block header (what the decoder in front stops at), kNoStart = none.
// Like every found start it only counts once the decoder of the chunk in front lands exactly on it.
uint64_t find_member_start(const BitSrc& src, uint64_t from, uint64_t to) {
  const size_t n = (size_t)(src.n_bits >> 3);
  size_t p = (size_t)((from + 7) >> 3);
  const size_t end = std::min<size_t>((size_t)(to >> 3), n > 32 ? n - 32 : 0);
  while (p < end) {
    const uint8_t* hit = static_cast<const uint8_t*>(std::memchr(src.data + p, 0x1f, end - p));
    if (!hit) break;
    p = (size_t)(hit - src.data);
    if (src.data[p + 1] == 0x8b && src.data[p + 2] == 8 && (src.data[p + 3] & 0xE0) == 0) {
      const size_t data0 = parse_gzip_header(src.data, n, p);
      if (data0 && (uint64_t)data0 * 8 + 64 < src.n_bits) {
        uint64_t pos = (uint64_t)data0 << 3;
        const uint64_t bits = src.peek(pos);
        const uint32_t type = (uint32_t)(bits >> 1) & 3u;
        bool ok = false;
        if (type == 0u) {  // stored: LEN and its complement
          const size_t b = (size_t)((pos + 3 + 7) >> 3);
          ok = b + 4 <= n && ((src.data[b] | (src.data[b + 1] << 8)) ^ (src.data[b + 2] | (src.data[b + 3] << 8))) == 0xFFFF;
        } else if (type == 1u) {
          ok = true;  // (fixed codes: nothing to check in the header; the landing rule decides)
        } else if (type == 2u) {
          BlockCodes bc;
          uint64_t q = pos + 3;
          ok = read_dynamic_header(src, q, bc, true);
        }
        if (ok) return pos;
      }
    }
    ++p;
  }
  return kNoStart;
}

// The first plausible dynamic-block start in [from, to) (bit positions), kNoStart = none.
uint64_t find_block_start(const BitSrc& src, uint64_t from, uint64_t to) {
  BlockCodes bc;
  to = std::min(to, src.n_bits > 64 ? src.n_bits - 64 : 0);
  for (uint64_t pos = from; pos < to; ++pos) {
    const uint64_t bits = src.peek(pos);
    // BFINAL = 0, BTYPE = 10b, HLIT <= 29, HDIST <= 29
    if ((bits & 7u) != 4u) continue;
    if (((bits >> 3) & 31u) > 29u || ((bits >> 8) & 31u) > 29u) continue;
    {
      // the code-length code must be complete (Kraft sum of its up to 19 three-bit lengths): decided from the bits in
      // hand, before any table is built -- one position in nine gets this far, one in a few hundred beyond
      const uint32_t hclen = (uint32_t)((bits >> 13) & 15u) + 4u;
      uint64_t cb = src.peek(pos + 17);
      uint32_t kraft = 0;
      for (uint32_t i = 0; i < hclen; ++i) {
        const uint32_t l = (uint32_t)(cb >> (3u * i)) & 7u;
        kraft += l ? (128u >> l) : 0u;
      }
      if (kraft != 128u) continue;
    }
    uint64_t p = pos + 3;
    if (!read_dynamic_header(src, p, bc, true)) continue;
    // a few hundred symbols: text-like literals, legal lengths and distances
    bool ok = true;
    uint32_t n_sym = 0, produced = 0;
    for (; ok && n_sym < 512u; ++n_sym) {
      if (p + 48 > src.n_bits) break;
      uint64_t b = src.peek(p);
      uint32_t len;
      uint32_t sym = bc.lit.decode(b, len);
      if (!len) {
        ok = false;
        break;
      }
      p += len;
      if (sym < 256u) {
        ok = sym == 9u || sym == 10u || sym == 13u || (sym >= 32u && sym < 127u);
        ++produced;
        continue;
      }
      if (sym == 256u) break;
      sym -= 257u;
      if (sym >= 29u || bc.dist_empty) {
        ok = false;
        break;
      }
      b >>= len;
      const uint32_t length = kLenBase[sym] + (uint32_t)(b & ((1u << kLenExtra[sym]) - 1u));
      p += kLenExtra[sym];
      b >>= kLenExtra[sym];
      uint32_t dl;
      const uint32_t ds = bc.dist.decode(b, dl);
      if (!dl || ds >= 30u) {
        ok = false;
        break;
      }
      b >>= dl;
      const uint32_t dist = kDistBase[ds] + (uint32_t)(b & ((1u << kDistExtra[ds]) - 1u));
      p += dl + kDistExtra[ds];
      if (dist > produced + kWindow) ok = false;
      produced += length;
    }
    if (ok) return pos;
  }
  return kNoStart;
}

}  // namespace

// ------------------------------------------------------------------ the reader
struct GzipReader::Impl {
  // zlib path
  gzFile gz = nullptr;
  // parallel path
  int fd = -1;
  const uint8_t* map = nullptr;
  size_t map_len = 0;
  BitSrc src;
  std::vector<Chunk> chunks;
  BufPool bufs, byte_bufs;  // (symbol buffers are seven times the byte buffers: one pool each, or every take reallocates)
  size_t lookahead = 0;
  std::mutex mu;
  std::condition_variable cv_work, cv_ready;
  std::vector<std::thread> pool;
  bool stop = false;
  size_t base = 0;      // first chunk the reader has not finished handing out
  size_t cur = 0;       // chunk being handed out
  size_t cur_off = 0;   // bytes of it already handed out
  size_t cur_seg = 0;   // crc segments of it already folded
  bool done = false;
  uint64_t merged = 0;
  // the member being read: running crc / length
  uint32_t m_crc = 0;
  uint64_t m_len = 0;
  std::string error;

  ~Impl() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : pool)
      if (t.joinable()) t.join();
    {
      std::lock_guard<std::mutex> lk(copy_mu);
      copy_stop = true;
    }
    copy_cv.notify_all();
    for (auto& t : copiers)
      if (t.joinable()) t.join();
    if (std::getenv("MIRGE_AMD_GZ_PROFILE"))
      std::fprintf(stderr, "[pgzip] reader: %.3f s waiting, %.3f s copying; workers (summed): find %.3f s, decode %.3f s, window %.3f s, resolve %.3f s\n",
                   t_wait, t_copy, us_find.load() / 1e6, us_decode.load() / 1e6, us_window.load() / 1e6, us_resolve.load() / 1e6);
    for (Chunk& c : chunks) {
      c.sym.release();
      c.bytes.release();
    }
    if (gz) gzclose(gz);
    if (map) munmap(const_cast<uint8_t*>(map), map_len);
    if (fd >= 0) close(fd);
  }

  // the chunk in front of k that is not dropped (k itself when k == 0); requires the finds in between to be done
  // (callers hold mu)
  bool prev_live(size_t k, size_t& prev) const {
    for (size_t j = k; j-- > 0;) {
      if (!chunks[j].find_done) return false;
      if (!chunks[j].dropped) {
        prev = j;
        return true;
      }
    }
    return false;
  }

  // ---- copy helpers: the reader's memcpy out of a chunk, split four ways ----
  static constexpr int kCopiers = 3;
  struct CopyJob {
    char* dst = nullptr;
    const uint8_t* src = nullptr;
    size_t len = 0;
    uint64_t posted = 0;             // generation the reader posted
    std::atomic<uint64_t> done{0};   // generation the helper finished
  } copy_jobs[kCopiers];
  std::mutex copy_mu;
  std::condition_variable copy_cv;
  std::vector<std::thread> copiers;
  bool copy_stop = false;
  void copier(int id) {
    uint64_t seen = 0;
    for (;;) {
      CopyJob& j = copy_jobs[id];
      {
        std::unique_lock<std::mutex> lk(copy_mu);
        copy_cv.wait(lk, [&] { return copy_stop || j.posted != seen; });
        if (copy_stop) return;
        seen = j.posted;
      }
      std::memcpy(j.dst, j.src, j.len);
      j.done.store(seen, std::memory_order_release);
    }
  }
  void copy_out(char* dst, const uint8_t* src, size_t len) {
    if (len < (256u << 10) || copiers.empty()) {
      std::memcpy(dst, src, len);
      return;
    }
    const size_t step = (len / (kCopiers + 1)) & ~(size_t)4095;
    uint64_t gen;
    {
      std::lock_guard<std::mutex> lk(copy_mu);
      gen = copy_jobs[0].posted + 1;
      for (int q = 0; q < kCopiers; ++q) {
        copy_jobs[q].dst = dst + (size_t)q * step;
        copy_jobs[q].src = src + (size_t)q * step;
        copy_jobs[q].len = step;
        copy_jobs[q].posted = gen;
      }
    }
    copy_cv.notify_all();
    std::memcpy(dst + (size_t)kCopiers * step, src + (size_t)kCopiers * step, len - (size_t)kCopiers * step);
    for (int q = 0; q < kCopiers; ++q)
      while (copy_jobs[q].done.load(std::memory_order_acquire) != gen) std::this_thread::yield();
  }

  double t_wait = 0, t_copy = 0;  // reader: seconds waiting for chunks / copying them out (MIRGE_AMD_GZ_PROFILE prints them)
  std::atomic<uint64_t> us_find{0}, us_decode{0}, us_window{0}, us_resolve{0};
  size_t next_find = 1;  // chunks below it have their find done or running
  size_t win_cur = 0, res_cur = 0, dec_cur = 0;  // cursors of the chain, the resolves and the decodes (worker())

  // runs chunk k's find here unless it is done or another thread has it (then waits for that one); lk held on entry and exit
  void ensure_find(std::unique_lock<std::mutex>& lk, size_t k) {
    Chunk& c = chunks[k];
    if (c.find_done) return;
    if (c.busy) {
      cv_ready.wait(lk, [&] { return c.find_done || stop; });
      return;
    }
    c.busy = true;
    lk.unlock();
    const uint64_t to = k + 1 < chunks.size() ? chunks[k + 1].nominal_bit : src.n_bits;
    // a member start in range bounds the search for a block start (whichever comes first is the chunk's start)
    const uint64_t member = find_member_start(src, c.nominal_bit, to);
    uint64_t found = find_block_start(src, c.nominal_bit, member == kNoStart ? to : std::min(to, member));
    bool clean = false;
    if (found == kNoStart && member != kNoStart) {
      found = member;
      clean = true;
    }
    lk.lock();
    c.busy = false;
    c.clean_start = clean;
    c.start_bit = found;
    if (found == kNoStart) c.dropped = true;
    c.find_done = true;
    cv_work.notify_all();
    cv_ready.notify_all();
  }

  void worker() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      if (stop) return;
      // ---- the most urgent runnable task; cursors keep a pick O(1): with a scan of the look-ahead window under
      // the one mutex, 64 workers woken by every state change spent more time in the lock than a chunk takes
      // to decode (the reader stopped scaling at 16 threads, 3 GB/s) ----
      const size_t hi = std::min(chunks.size(), base + lookahead);
      int what = 0;  // 1 find, 2 decode, 3 window, 4 resolve
      size_t k = 0;
      if (win_cur < base) win_cur = base;
      if (res_cur < base) res_cur = base;
      if (dec_cur < base) dec_cur = base;
      // the chain first (everything behind waits for it): the first live chunk without a window
      while (win_cur < chunks.size() && ((chunks[win_cur].find_done && chunks[win_cur].dropped) || chunks[win_cur].window_done)) ++win_cur;
      if (win_cur < hi) {
        Chunk& c = chunks[win_cur];
        size_t pv = 0;
        if (c.decode_done && c.error.empty() && !c.windowing &&
            (win_cur == 0 || c.clean_start || (prev_live(win_cur, pv) && chunks[pv].window_done))) {
          what = 3;
          k = win_cur;
        }
      }
      // resolve: every decoded chunk up to the chain's position has its window in front of it
      if (!what) {
        while (res_cur < chunks.size() && (chunks[res_cur].dropped || chunks[res_cur].resolving || chunks[res_cur].resolve_done) &&
               (chunks[res_cur].find_done || chunks[res_cur].resolve_done))
          ++res_cur;
        for (size_t j = res_cur; j < hi && j <= win_cur; ++j) {
          Chunk& c = chunks[j];
          if (c.dropped || !c.decode_done || !c.error.empty() || c.resolving || c.resolve_done) continue;
          size_t pv = 0;
          if (j == 0 || c.clean_start || (prev_live(j, pv) && chunks[pv].window_done)) {
            what = 4;
            k = j;
            break;
          }
        }
      }
      while (!what && next_find < chunks.size() && (chunks[next_find].find_done || chunks[next_find].busy)) ++next_find;
      if (!what && next_find < chunks.size() && next_find < base + 4 * lookahead) {
        what = 1;
        k = next_find++;
      }
      if (!what) {
        while (dec_cur < chunks.size() && ((chunks[dec_cur].find_done && chunks[dec_cur].dropped) || chunks[dec_cur].decode_done ||
                                            (chunks[dec_cur].find_done && chunks[dec_cur].busy)))
          ++dec_cur;
        for (size_t j = dec_cur; j < hi; ++j) {
          Chunk& c = chunks[j];
          if (!c.find_done) break;  // (finds complete in order of their start: nothing behind is ready either)
          if (c.dropped || c.busy || c.decode_done) continue;
          // the next chunks' finds should be known: the decoder stops at the first live start behind it
          bool known = true;
          for (size_t q = j + 1; q < std::min(chunks.size(), j + 3); ++q) known &= chunks[q].find_done;
          if (known) {
            what = 2;
            k = j;
          }
          break;
        }
      }
      if (!what) {
        cv_work.wait(lk);
        continue;
      }
      Chunk& c = chunks[k];
      const auto tt0 = std::chrono::steady_clock::now();
      struct Tick {
        std::atomic<uint64_t>* to;
        std::chrono::steady_clock::time_point t0;
        ~Tick() { *to += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
      } tick{what == 1 ? &us_find : what == 2 ? &us_decode : what == 3 ? &us_window : &us_resolve, tt0};
      if (what == 1) {
        ensure_find(lk, k);
      } else if (what == 2) {
        c.busy = true;
        lk.unlock();
        c.sym = bufs.take();
        std::string err;
        size_t swallowed_to = k;  // chunks (k, swallowed_to] turned out to be covered by this one
        try {
          size_t target = k + 1;
          decode_chunk(src, c, [&](uint64_t bit) -> int {
            // the next live start at or behind `bit`
            while (target < chunks.size()) {
              bool dropped_;
              uint64_t st;
              {
                std::unique_lock<std::mutex> l2(mu);
                ensure_find(l2, target);
                if (stop) throw std::runtime_error("reader closed");
                dropped_ = chunks[target].dropped;
                st = chunks[target].start_bit;
              }
              if (!dropped_) {
                if (st == bit) return 1;
                if (st > bit) return 0;
                swallowed_to = target;  // ran past it without landing on it: it was not a block start
              }
              ++target;
            }
            return 0;
          });
        } catch (const std::exception& e) {
          err = e.what();
        }
        lk.lock();
        // (a chunk that was itself swallowed meanwhile -- its start was no block start: the decoder in front ran past
        // it -- has nothing to say about the chunks behind it: that decoder may have stopped at one of them)
        if (err.empty() && !c.dropped) {
          for (size_t q = k + 1; q < chunks.size() && (q <= swallowed_to || c.at_eof); ++q) {
            if (!chunks[q].dropped && q <= swallowed_to) ++merged;
            chunks[q].dropped = true;
            chunks[q].find_done = true;
          }
        }
        c.error = err;
        c.busy = false;
        c.decode_done = true;
        cv_work.notify_all();
        cv_ready.notify_all();
      } else if (what == 3) {
        // the last 32 KB of the stream up to the end of this chunk, as bytes
        size_t pv = 0;
        const bool has_prev = k != 0 && !c.clean_start && prev_live(k, pv);
        const uint8_t* pw = has_prev ? chunks[pv].window.data() : nullptr;
        if (has_prev && chunks[pv].end_bit != c.start_bit) {
          c.error = "internal: chunk chain broken";
          c.window_done = true;
          cv_ready.notify_all();
          continue;
        }
        c.windowing = true;
        lk.unlock();
        std::vector<uint8_t> win(kWindow, 0);
        std::string err;
        const size_t n = c.n_sym;
        const uint16_t* sym = static_cast<const uint16_t*>(c.sym.p);
        const size_t take = std::min<size_t>(n, kWindow);
        // (window index w of THIS chunk's references = byte w of the previous window)
        if (take < kWindow && pw) std::memcpy(win.data(), pw + take, kWindow - take);
        for (size_t i = 0; i < take; ++i) {
          const uint16_t v = sym[n - take + i];
          if (v & 0x8000u) {
            if (!pw) {
              err = "distance beyond the start of the gzip stream";
              break;
            }
            win[kWindow - take + i] = pw[v & 0x7FFFu];
          } else {
            win[kWindow - take + i] = (uint8_t)v;
          }
        }
        lk.lock();
        c.window = std::move(win);
        if (!err.empty() && c.error.empty()) c.error = err;
        c.windowing = false;
        c.window_done = true;
        cv_work.notify_all();
        cv_ready.notify_all();
      } else {
        size_t pv = 0;
        const bool has_prev = k != 0 && !c.clean_start && prev_live(k, pv);
        const uint8_t* pw = has_prev ? chunks[pv].window.data() : nullptr;
        c.resolving = true;
        lk.unlock();
        std::string err;
        const size_t n = c.n_sym;
        const uint16_t* sym = static_cast<const uint16_t*>(c.sym.p);
        RawBuf bb = byte_bufs.take();
        bb.ensure(n + 64, 0);
        uint8_t* bytes = static_cast<uint8_t*>(bb.p);
        if (!resolve_symbols(sym, n, pw, bytes)) err = "distance beyond the start of the gzip stream";
        std::vector<uint32_t> seg_crc;
        size_t seg_from = 0;
        for (size_t e = 0; e <= c.ends.size(); ++e) {
          const size_t to = e < c.ends.size() ? (size_t)c.ends[e].out_off : n;
          seg_crc.push_back(fast_crc32((uint32_t)crc32(0L, Z_NULL, 0), bytes + seg_from, to - seg_from));
          seg_from = to;
        }
        lk.lock();
        // (the symbols stay until the chunk's window has been cut from them)
        c.bytes = bb;
        c.seg_crc = std::move(seg_crc);
        if (!err.empty() && c.error.empty()) c.error = err;
        c.resolving = false;
        c.resolve_done = true;
        cv_ready.notify_all();
        cv_work.notify_all();
      }
    }
  }
};

GzipReader::GzipReader(const std::string& path, int threads, size_t chunk_bytes) : impl_(new Impl()) {
  Impl& m = *impl_;
  if (threads > 64) threads = 64;  // (more workers only add memory: 3 x threads chunks of ~18 MB are in flight)
  if (!chunk_bytes) {
    chunk_bytes = 1u << 20;
    // (tests cut small files into many chunks)
    if (const char* e = std::getenv("MIRGE_AMD_GZ_CHUNK")) {
      const long v = std::atol(e);
      if (v >= 4096) chunk_bytes = (size_t)v;
    }
  }
  bool par = threads > 1;
  if (par) {
    m.fd = open(path.c_str(), O_RDONLY);
    if (m.fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    if (fstat(m.fd, &st) != 0) throw std::runtime_error("cannot stat " + path);
    m.map_len = (size_t)st.st_size;
    par = m.map_len >= 2 * chunk_bytes + 64;
    if (par) {
      void* p = mmap(nullptr, m.map_len, PROT_READ, MAP_PRIVATE, m.fd, 0);
      if (p == MAP_FAILED) throw std::runtime_error("cannot map " + path);
      m.map = (const uint8_t*)p;
      (void)madvise(p, m.map_len, MADV_SEQUENTIAL);
      const size_t data0 = parse_gzip_header(m.map, m.map_len, 0);
      if (!data0) {
        par = false;
        munmap(p, m.map_len);
        m.map = nullptr;
      } else {
        m.src.data = m.map;
        m.src.n_bits = (uint64_t)m.map_len * 8;
        // (a file of few chunks: smaller ones, so that every thread has some)
        while (chunk_bytes > (256u << 10) && (m.map_len - data0) / chunk_bytes < 4u * (size_t)threads) chunk_bytes /= 2;
        const size_t n_chunks = (m.map_len - data0 + chunk_bytes - 1) / chunk_bytes;
        m.chunks.resize(std::max<size_t>(n_chunks, 1));
        for (size_t k = 0; k < m.chunks.size(); ++k) {
          m.chunks[k].nominal_bit = (uint64_t)(data0 + k * chunk_bytes) * 8;
          m.chunks[k].sym_hint = chunk_bytes * 14;
        }
        m.chunks[0].start_bit = (uint64_t)data0 * 8;
        m.chunks[0].find_done = true;
        m.chunks[0].clean_start = true;
        m.lookahead = std::max<size_t>(8, 2 * (size_t)threads + 4);
        for (int t = 0; t < threads; ++t) m.pool.emplace_back([&m] { m.worker(); });
        if (threads >= 8)
          for (int q = 0; q < Impl::kCopiers; ++q) m.copiers.emplace_back([&m, q] { m.copier(q); });
      }
    }
    if (!par) {
      close(m.fd);
      m.fd = -1;
    }
  }
  if (!par) {
    m.gz = gzopen(path.c_str(), "rb");
    if (!m.gz) throw std::runtime_error("cannot open " + path);
    gzbuffer(m.gz, 1 << 20);
  }
}

GzipReader::~GzipReader() {}

bool GzipReader::parallel() const { return impl_->gz == nullptr; }
uint64_t GzipReader::chunks_merged() const {
  std::lock_guard<std::mutex> lk(impl_->mu);
  return impl_->merged;
}

size_t GzipReader::read(char* dst, size_t n) {
  Impl& m = *impl_;
  if (m.gz) {
    size_t total = 0;
    while (total < n) {
      const int got = gzread(m.gz, dst + total, (unsigned)std::min<size_t>(n - total, 1u << 30));
      if (got < 0) throw std::runtime_error("read error (corrupt gzip?)");
      if (got == 0) break;
      total += (size_t)got;
    }
    return total;
  }
  size_t total = 0;
  while (total < n && !m.done) {
    std::unique_lock<std::mutex> lk(m.mu);
    // the next live chunk
    while (m.cur < m.chunks.size() && m.chunks[m.cur].find_done && m.chunks[m.cur].dropped) {
      ++m.cur;
      m.base = m.cur;
      m.cv_work.notify_all();
    }
    if (m.cur >= m.chunks.size()) {
      m.done = true;
      break;
    }
    Chunk& c = m.chunks[m.cur];
    const auto tw0 = std::chrono::steady_clock::now();
    m.cv_ready.wait(lk, [&] { return (c.find_done && c.dropped) || (c.resolve_done && c.window_done) || (c.decode_done && !c.error.empty()); });
    m.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
    if (c.find_done && c.dropped) continue;  // (swallowed by the chunk in front while we waited)
    if (!c.error.empty()) throw std::runtime_error("corrupt gzip stream: " + c.error);
    lk.unlock();
    // hand out bytes, folding the member checks as their ends go by
    const size_t avail = c.n_sym - m.cur_off;
    const size_t take = std::min(avail, n - total);
    // (a chunk's bytes were written by another core, often on the other socket: one thread copies them out at
    // 3 GB/s, which capped the whole reader whatever the worker count; the copy helpers take three quarters of a piece)
    const auto tc0 = std::chrono::steady_clock::now();
    m.copy_out(dst + total, static_cast<const uint8_t*>(c.bytes.p) + m.cur_off, take);
    m.t_copy += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
    total += take;
    m.cur_off += take;
    if (m.cur_off == c.n_sym) {
      // the whole chunk is out: its crc segments
      size_t from = 0;
      for (size_t e = 0; e <= c.ends.size(); ++e) {
        const size_t to = e < c.ends.size() ? (size_t)c.ends[e].out_off : c.n_sym;
        m.m_crc = (uint32_t)crc32_combine(m.m_crc, c.seg_crc[e], (z_off_t)(to - from));
        m.m_len += to - from;
        if (e < c.ends.size()) {
          if (m.m_crc != c.ends[e].crc || (uint32_t)m.m_len != c.ends[e].isize)
            throw std::runtime_error("corrupt gzip stream: CRC or length mismatch");
          m.m_crc = 0;
          m.m_len = 0;
        }
        from = to;
      }
      const bool eof = c.at_eof;
      lk.lock();
      m.byte_bufs.give(c.bytes);
      // (the 32 KB windows stay: the chunks behind are cut from them)
      if (c.window_done) m.bufs.give(c.sym);
      ++m.cur;
      m.cur_off = 0;
      m.base = m.cur;
      if (eof) m.done = true;
      m.cv_work.notify_all();
    }
  }
  return total;
}

}  // namespace mrg
