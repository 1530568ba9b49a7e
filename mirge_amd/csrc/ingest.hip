// FASTQ text -> packed reads ON THE DEVICE: record splitting, 3' quality trimming, the `-ad +N` cutter,
// the minimum length and the 2-bit packing of trim_file (utils/trim_file.py:24-66, 89-134) and of the
// FASTQ loop of quantReads (utils/quantReads.py:4-24), for raw text blocks that the host only reads,
// cuts at a record boundary and uploads.
//
// Why: csrc/fastq.cpp does this on the host at 85 M reads/s on the GPU box's 256 cores
// (profiles/r03_ingest_scale.json) -- 1.2 s for what the cascade annotates in 6 ms.  Here a block of text
// is one upload and three kernels; the packed reads never exist on the host.
// Scope: strict four-line records ('\n' or "\r\n", last newline optional), `-ad none`, `-ad +N` and (round 4)
// adapter SEQUENCES (`-ad illumina`, the reference's own usage example, parseArgument.py:29: cutadapt's 3'
// search, one thread per read); anything else (blank lines between records) is reported (status != 0) and
// the caller takes the host parser, which also words the errors.
// Rules restated exactly as csrc/fastq.cpp / oracle/ingest.py state them:
//   quality  walk from the 3' end accumulating (cutoff - q), stop when the sum turns negative, cut at the
//            position of the maximum (cutadapt's / BWA's rule);
//   +N       N > 0 removes the first N bases, N < 0 the last -N, after the quality trim;
//   keep     trimmed length >= min_len; bases other than ACGT (any case) are N (code 0 + mask bit).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_util.hpp"
#include "kernels.hpp"
#include "prims.hpp"

namespace mrg {

namespace {

constexpr uint32_t kIngestThreads = 256;

// Positions of the newlines of a text block, in order: a workgroup takes a tile of 16 KB (64 consecutive bytes per
// thread), counts its newlines (COUNT: into counts[tile]), and -- behind one prefix sum over the tiles (prims.hip) --
// writes their positions from the tile's offset on, every thread from its own (workgroup prefix of the thread counts).
constexpr uint32_t kNlPerThread = 64u, kNlTile = kIngestThreads * kNlPerThread;
template <bool COUNT>
__global__ void __launch_bounds__(kIngestThreads) newline_kernel(const char* __restrict__ text, uint32_t n, uint32_t* __restrict__ counts,
                                                                 const uint32_t* __restrict__ offs, uint32_t* __restrict__ pos, uint32_t pos_cap) {
  __shared__ uint32_t wtot[kIngestThreads / 64u];
  const uint32_t base = blockIdx.x * kNlTile + threadIdx.x * kNlPerThread;
  uint64_t mask = 0;  // bit i: byte base + i is a newline
  if (base + kNlPerThread <= n) {
#pragma unroll
    for (uint32_t q = 0; q < kNlPerThread / 16u; ++q) {
      const uint4 v = *reinterpret_cast<const uint4*>(text + base + 16u * q);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k)
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b)
          if (((w[k] >> (8u * b)) & 255u) == (uint32_t)'\n') mask |= 1ull << (16u * q + 4u * k + b);
    }
  } else {
    for (uint32_t i = 0; i < kNlPerThread; ++i)
      if (base + i < n && text[base + i] == '\n') mask |= 1ull << i;
  }
  const uint32_t c = (uint32_t)__popcll(mask);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t incl = dev::wave_incl_scan(c);
  if (lane == 63u) wtot[wave] = incl;
  __syncthreads();
  uint32_t before = incl - c, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kIngestThreads / 64u; ++w) {
    before += w < wave ? wtot[w] : 0u;
    total += wtot[w];
  }
  if (COUNT) {
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
  } else {
    uint32_t at = offs[blockIdx.x] + before;
    while (mask) {
      const uint32_t i = (uint32_t)__ffsll((long long)mask) - 1u;
      mask &= mask - 1ull;
      if (at < pos_cap) pos[at] = base + i;
      ++at;
    }
  }
}

__device__ __forceinline__ char upper_char(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

// cutadapt's 3' adapter search (`-a ADAPTER`, error rate 0.12, minimum overlap 3: trim_file.py:30-41) on one
// read: csrc/fastq.cpp's locate_adapter_3p restated cell for cell -- an exact occurrence first, else the
// semi-global alignment (the adapter from its first base, free start in the read, the adapter may run off the
// read's 3' end) column by column with Ukkonen's band, the candidate rule "more matches, then fewer errors",
// and the last column's rows examined as the original examines them.  rd[0, n): the read (any case).
// Returns true and where the read is cut / how many bases matched.
__device__ bool locate_adapter_dev(const char* ad, int m, int k, const char* rd, int n, int& read_start, int& n_matches) {
  if (m == 0) return false;
  if (n >= m) {
    for (int p0 = 0; p0 + m <= n; ++p0) {
      int i = 0;
      while (i < m && upper_char(rd[p0 + i]) == ad[i]) ++i;
      if (i == m) {
        read_start = p0;
        n_matches = m;
        return true;
      }
    }
  }
  constexpr double kRate = 0.12;
  const int min_overlap = m < 3 ? m : 3;
  short cost[kAdapterMaxLen + 1], mat[kAdapterMaxLen + 1], org[kAdapterMaxLen + 1];
  for (int i = 0; i <= m; ++i) {
    cost[i] = (short)i;
    mat[i] = 0;
    org[i] = 0;
  }
  int best_cost = m + n, best_matches = 0, best_origin = 0;
  auto consider = [&](int i) -> bool {
    const int length = i + (org[i] < 0 ? (int)org[i] : 0);
    const int c = cost[i], mt = mat[i];
    if (length >= min_overlap && (double)c <= (double)length * kRate && (mt > best_matches || (mt == best_matches && c < best_cost))) {
      best_matches = mt;
      best_cost = c;
      best_origin = org[i];
      return true;
    }
    return false;
  };
  int last = m < k + 1 ? m : k + 1;
  bool exact_full = false;
  for (int j = 1; j <= n && !exact_full; ++j) {
    int d_cost = cost[0], d_mat = mat[0], d_org = org[0];
    org[0] = (short)j;
    const char c = upper_char(rd[j - 1]);
    for (int i = 1; i <= last; ++i) {
      int c_cost, c_mat, c_org;
      if (ad[i - 1] == c) {
        c_cost = d_cost;
        c_mat = d_mat + 1;
        c_org = d_org;
      } else {
        const int c_diag = d_cost + 1, c_del = cost[i] + 1, c_ins = cost[i - 1] + 1;
        if (c_diag <= c_del && c_diag <= c_ins) {
          c_cost = c_diag;
          c_mat = d_mat;
          c_org = d_org;
        } else if (c_ins <= c_del) {
          c_cost = c_ins;
          c_mat = mat[i - 1];
          c_org = org[i - 1];
        } else {
          c_cost = c_del;
          c_mat = mat[i];
          c_org = org[i];
        }
      }
      d_cost = cost[i];
      d_mat = mat[i];
      d_org = org[i];
      cost[i] = (short)c_cost;
      mat[i] = (short)c_mat;
      org[i] = (short)c_org;
    }
    while (last >= 0 && cost[last] > k) --last;
    if (last < m) {
      ++last;
    } else if (consider(m) && best_cost == 0 && best_matches == m) {
      exact_full = true;
    }
  }
  if (!exact_full)
    for (int i = 0; i <= m; ++i) consider(i);
  if (best_cost == m + n) return false;
  read_start = best_origin >= 0 ? best_origin : 0;
  n_matches = best_matches;
  return true;
}

// per record: where its trimmed read starts in the text and how long it is (0 = dropped)
__global__ void __launch_bounds__(kIngestThreads) ingest_trim_kernel(const char* __restrict__ text, uint64_t n_bytes,
                                                                      const uint32_t* __restrict__ nl, uint32_t n_nl, uint32_t n_records,
                                                                      int32_t phred, int32_t cutoff, int32_t min_len, int32_t cut,
                                                                      const AdapterSet ads, uint32_t max_packed, uint32_t* __restrict__ rec_start,
                                                                      uint32_t* __restrict__ rec_len, uint32_t* __restrict__ keep,
                                                                      uint32_t* __restrict__ info /* status, bad record, n_long, max_len */) {
  const uint32_t r = blockIdx.x * kIngestThreads + threadIdx.x;
  if (r >= n_records) return;
  auto line_end = [&](uint32_t i) -> uint32_t { return i < n_nl ? nl[i] : (uint32_t)n_bytes; };
  const uint32_t s0 = r ? nl[4 * r - 1] + 1 : 0u;
  const uint32_t e0 = line_end(4 * r), e1 = line_end(4 * r + 1), e2 = line_end(4 * r + 2);
  uint32_t e3 = line_end(4 * r + 3);
  const uint32_t s1 = e0 + 1, s2 = e1 + 1, s3 = e2 + 1;
  uint32_t q1 = e1;
  if (q1 > s1 && text[q1 - 1] == '\r') --q1;
  if (e3 > s3 && text[e3 - 1] == '\r') --e3;
  uint32_t len = 0, start = s1;
  uint32_t bad = 0;
  if (e0 <= s0 || text[s0] != '@') bad = 1;                      // not a header line (or a blank line)
  else if (s2 >= (uint32_t)n_bytes || text[s2] != '+') bad = 2;  // the third line is not the '+' line
  else if (q1 - s1 != e3 - s3) bad = 3;                          // sequence and quality lengths differ
  if (bad) {
    atomicCAS(&info[0], 0u, bad);
    atomicMin(&info[1], r);
  } else {
    const uint32_t L = q1 - s1;
    // 3' quality trim
    int32_t s = 0, best = 0;
    uint32_t stop = L;
    for (uint32_t i = L; i-- > 0;) {
      s += cutoff - ((int32_t)(uint8_t)text[s3 + i] - phred);
      if (s < 0) break;
      if (s > best) {
        best = s;
        stop = i;
      }
    }
    uint32_t a = 0, b = stop;
    if (cut > 0) a = min((uint32_t)cut, b);
    else if (cut < 0) b = b > (uint32_t)(-cut) ? b - (uint32_t)(-cut) : 0u;
    len = b - a;
    start = s1 + a;
    // adapter sequences (trim_file.py:34-41; fastq.cpp: apply_trim_spec): the adapter with the most matched bases
    // decides, the first one on ties; the read is cut where its match begins
    if (ads.n) {
      bool found = false;
      int best_start = 0, best_matches = 0;
      for (uint32_t q = 0; q < ads.n; ++q) {
        int rs = 0, nm = 0;
        if (locate_adapter_dev(ads.seq[q], (int)ads.len[q], (int)ads.k[q], text + start, (int)len, rs, nm) && (!found || nm > best_matches)) {
          found = true;
          best_start = rs;
          best_matches = nm;
        }
      }
      if (found) len = (uint32_t)best_start;
    }
  }
  uint32_t k = (!bad && (int32_t)len >= min_len) ? 1u : 0u;
  if (k && len > max_packed) {  // kept by the rules, too long for the words the caller offers
    atomicAdd(&info[2], 1u);
    k = 0u;
  }
  if (k) atomicMax(&info[3], len);
  rec_start[r] = start;
  rec_len[r] = k ? len : 0u;
  keep[r] = k;
}

__global__ void __launch_bounds__(kIngestThreads) ingest_pack_kernel(const char* __restrict__ text, const uint32_t* __restrict__ rec_start,
                                                                      const uint32_t* __restrict__ rec_len, const uint32_t* __restrict__ out_idx,
                                                                      uint32_t n_records, uint32_t W, uint64_t cap, uint64_t* __restrict__ words,
                                                                      uint8_t* __restrict__ lens, uint64_t* __restrict__ nmask,
                                                                      uint32_t* __restrict__ info /* [4] has_n */) {
  const uint32_t r = blockIdx.x * kIngestThreads + threadIdx.x;
  if (r >= n_records) return;
  const uint32_t len = rec_len[r];
  if (!len) return;
  const uint64_t o = out_idx[r];
  const char* p = text + rec_start[r];
  bool any_n = false;
  for (uint32_t w = 0; w < W; ++w) {
    uint64_t bits = 0, nm = 0;
    const uint32_t nb = len > 32u * w ? min(32u, len - 32u * w) : 0u;
    for (uint32_t i = 0; i < nb; ++i) {
      const char c = p[32u * w + i];
      uint64_t code = 0;
      bool is_n = false;
      switch (c) {
        case 'A': case 'a': code = 0; break;
        case 'C': case 'c': code = 1; break;
        case 'G': case 'g': code = 2; break;
        case 'T': case 't': code = 3; break;
        default: is_n = true; break;
      }
      bits |= code << (2u * i);
      if (is_n) nm |= 1ull << (2u * i);
    }
    words[(uint64_t)w * cap + o] = bits;
    if (nmask) nmask[(uint64_t)w * cap + o] = nm;
    any_n |= nm != 0ull;
  }
  lens[o] = (uint8_t)len;
  if (any_n) atomicOr(&info[4], 1u);
}

#define CK(expr)                     \
  do {                               \
    hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return e_; \
  } while (0)

struct Scratch {
  void* p = nullptr;
  ~Scratch() { (void)hipFree(p); }
};

}  // namespace

hipError_t fastq_parse_device(const char* d_text, uint64_t n_bytes, int32_t phred, int32_t cutoff, int32_t min_len, int32_t cut,
                              const AdapterSet& ads, uint32_t W, uint64_t cap, uint64_t* d_words, uint8_t* d_lens, uint64_t* d_nmask,
                              uint64_t* h_info, hipStream_t stream) {
  // h_info: records, kept (packed), too long, max_len, has_n, status (0 ok; 1-3 ill-formed record, 4 line count not a
  // multiple of four, 5 more kept reads than `cap`), first bad record
  for (int i = 0; i < 7; ++i) h_info[i] = 0;
  if (n_bytes == 0) return hipSuccess;
  if (n_bytes >= 0x7fffffffull) return hipErrorInvalidValue;
  const uint32_t n = (uint32_t)n_bytes;
  // ---- positions of the newlines ----
  Scratch s_nl, s_cnt, s_tmp, s_rec, s_info;
  CK(hipMalloc(&s_nl.p, (size_t)(n / 2 + 16) * 4));  // (a line has at least one byte besides its '\n'... blank lines are errors anyway)
  CK(hipMalloc(&s_cnt.p, 8));
  const uint32_t n_tiles = (n + kNlTile - 1) / kNlTile;
  CK(hipMalloc(&s_tmp.p, (size_t)(n_tiles + 1) * 4));
  uint32_t* tile_cnt = (uint32_t*)s_tmp.p;
  Scratch s_scan0;
  CK(hipMalloc(&s_scan0.p, prims::scan_temp_bytes(n_tiles + 1)));
  CK(hipMemsetAsync(tile_cnt + n_tiles, 0, 4, stream));
  const uint32_t pos_cap = n / 2 + 16;
  hipLaunchKernelGGL(newline_kernel<true>, dim3(n_tiles), dim3(kIngestThreads), 0, stream, d_text, n, tile_cnt, (const uint32_t*)nullptr,
                     (uint32_t*)nullptr, 0u);
  CK(hipGetLastError());
  CK(prims::exclusive_sum_u32(tile_cnt, tile_cnt, n_tiles + 1, s_scan0.p, stream));   // (the entry behind the last tile: the total)
  uint32_t n_nl = 0;
  char last = 0;
  CK(hipMemcpyAsync(&n_nl, tile_cnt + n_tiles, 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&last, d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  // (a text of nothing but newlines would overflow the position array: blank lines are not four-line records anyway)
  if ((uint64_t)n_nl > (uint64_t)n / 2 + 8) {
    h_info[5] = 1;
    return hipSuccess;
  }
  hipLaunchKernelGGL(newline_kernel<false>, dim3(n_tiles), dim3(kIngestThreads), 0, stream, d_text, n, (uint32_t*)nullptr, tile_cnt,
                     (uint32_t*)s_nl.p, pos_cap);
  CK(hipGetLastError());
  const uint32_t n_lines = n_nl + (last != '\n' ? 1u : 0u);
  if (n_lines % 4u) {
    h_info[5] = 4;
    return hipSuccess;
  }
  const uint32_t n_records = n_lines / 4u;
  h_info[0] = n_records;
  if (!n_records) return hipSuccess;
  // ---- per record: trimmed extent, keep flag ----
  CK(hipMalloc(&s_rec.p, (size_t)n_records * 16));
  uint32_t* rec_start = (uint32_t*)s_rec.p;
  uint32_t* rec_len = rec_start + n_records;
  uint32_t* keep = rec_len + n_records;
  uint32_t* out_idx = keep + n_records;
  CK(hipMalloc(&s_info.p, 32));
  {
    const uint32_t init[8] = {0u, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u};
    CK(hipMemcpyAsync(s_info.p, init, 32, hipMemcpyHostToDevice, stream));
  }
  const uint32_t grid = (n_records + kIngestThreads - 1) / kIngestThreads;
  hipLaunchKernelGGL(ingest_trim_kernel, dim3(grid), dim3(kIngestThreads), 0, stream, d_text, n_bytes, (const uint32_t*)s_nl.p, n_nl,
                     n_records, phred, cutoff, min_len, cut, ads, 32u * W < 255u ? 32u * W : 255u /* one length byte */, rec_start, rec_len, keep, (uint32_t*)s_info.p);
  CK(hipGetLastError());
  // ---- output positions: exclusive prefix of the keep flags ----
  Scratch s_scan;
  CK(hipMalloc(&s_scan.p, prims::scan_temp_bytes(n_records)));
  CK(prims::exclusive_sum_u32(keep, out_idx, n_records, s_scan.p, stream));
  uint32_t last_idx = 0, last_keep = 0, info[8];
  CK(hipMemcpyAsync(&last_idx, out_idx + n_records - 1, 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&last_keep, keep + n_records - 1, 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(info, s_info.p, 32, hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  const uint64_t n_kept = (uint64_t)last_idx + last_keep;
  h_info[2] = info[2];
  h_info[3] = info[3];
  if (info[0]) {
    h_info[5] = info[0];
    h_info[6] = (uint64_t)info[1] + 1;  // 1-based record number inside the block
    return hipSuccess;
  }
  if (n_kept > cap) {
    h_info[5] = 5;
    return hipSuccess;
  }
  h_info[1] = n_kept;
  if (n_kept) {
    hipLaunchKernelGGL(ingest_pack_kernel, dim3(grid), dim3(kIngestThreads), 0, stream, d_text, rec_start, rec_len, out_idx, n_records, W, cap,
                       d_words, d_lens, d_nmask, (uint32_t*)s_info.p);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(info, s_info.p, 32, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    h_info[4] = info[4];
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// The compact wire form of a collapsed read set that lives on the HOST (include/mirge_amd.h:
// mrg_expand_compact): reads grouped by length as a bit stream of 2 L bits each, one-byte counts with an
// escape list -- 6.5 bytes per 22-nt read over PCIe instead of 13 -- widened here into the arrays the
// cascade and the tally take.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kExpandThreads = 256;

// One read per thread: read j of a run of L-base reads sits at bit 2 L j of the run's words.
__global__ void __launch_bounds__(kExpandThreads) expand_bits_kernel(const uint64_t* __restrict__ in, CompactRuns runs, uint32_t n,
                                                                      uint64_t* __restrict__ words, uint8_t* __restrict__ lens) {
  __shared__ uint32_t s_end[kCompactMaxRuns];
  __shared__ uint32_t s_len[kCompactMaxRuns];
  __shared__ uint32_t s_base[kCompactMaxRuns];
  if (threadIdx.x < kCompactMaxRuns) {
    s_end[threadIdx.x] = threadIdx.x < runs.n ? runs.end[threadIdx.x] : 0xFFFFFFFFu;
    s_len[threadIdx.x] = threadIdx.x < runs.n ? runs.len[threadIdx.x] : 0u;
    s_base[threadIdx.x] = threadIdx.x < runs.n ? runs.base[threadIdx.x] : 0u;
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * kExpandThreads + threadIdx.x;
  if (i >= n) return;
  // the run of read i: the first one that ends behind it
  uint32_t lo = 0, hi = runs.n - 1u;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (s_end[mid] > i) hi = mid;
    else lo = mid + 1u;
  }
  const uint32_t L = s_len[lo], start = lo ? s_end[lo - 1u] : 0u;
  const uint64_t bit = (uint64_t)(i - start) * (2u * L);
  const uint64_t* src = in + s_base[lo] + (bit >> 6);
  const uint32_t sh = (uint32_t)(bit & 63u);
  uint64_t w = src[0] >> sh;
  if (sh + 2u * L > 64u) w |= src[1] << (64u - sh);
  words[i] = L >= 32u ? w : (w & ((1ull << (2u * L)) - 1ull));
  lens[i] = (uint8_t)L;
}

// one-byte counts -> 32-bit counts, four per thread (255 = escaped: overwritten by apply_escapes_kernel)
__global__ void __launch_bounds__(kExpandThreads) widen_counts_kernel(const uint8_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * kExpandThreads + threadIdx.x) * 4ull;
  if (i0 >= n) return;
  if (i0 + 4ull <= n) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(in + i0);
    *reinterpret_cast<uint4*>(out + i0) = make_uint4(v & 255u, (v >> 8) & 255u, (v >> 16) & 255u, v >> 24);
  } else {
    for (uint64_t i = i0; i < n; ++i) out[i] = in[i];
  }
}

__global__ void __launch_bounds__(kExpandThreads) apply_escapes_kernel(const uint32_t* __restrict__ esc, uint64_t n_esc, uint64_t n,
                                                                        uint32_t* __restrict__ out) {
  const uint64_t k = (uint64_t)blockIdx.x * kExpandThreads + threadIdx.x;
  if (k >= n_esc) return;
  const uint32_t i = esc[2 * k];
  if (i < n) out[i] = esc[2 * k + 1];
}

}  // namespace

hipError_t expand_compact(const uint64_t* d_bits, const CompactRuns& runs, const uint8_t* d_quant8, const uint32_t* d_esc, uint64_t n_esc,
                          uint64_t n, uint32_t n_samples, uint64_t* d_reads, uint8_t* d_lens, uint32_t* d_quant, hipStream_t stream) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(expand_bits_kernel, dim3((uint32_t)((n + kExpandThreads - 1) / kExpandThreads)), dim3(kExpandThreads), 0, stream, d_bits,
                     runs, (uint32_t)n, d_reads, d_lens);
  CK(hipGetLastError());
  if (d_quant8) {
    const uint64_t m = n * n_samples, mq = (m + 3) / 4;
    hipLaunchKernelGGL(widen_counts_kernel, dim3((uint32_t)((mq + kExpandThreads - 1) / kExpandThreads)), dim3(kExpandThreads), 0, stream,
                       d_quant8, m, d_quant);
    CK(hipGetLastError());
    if (n_esc) {
      hipLaunchKernelGGL(apply_escapes_kernel, dim3((uint32_t)((n_esc + kExpandThreads - 1) / kExpandThreads)), dim3(kExpandThreads), 0,
                         stream, d_esc, n_esc, m, d_quant);
      CK(hipGetLastError());
    }
  }
  return hipSuccess;
}

}  // namespace mrg
