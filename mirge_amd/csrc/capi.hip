// C-ABI of include/mirge_amd.h: handle management, HBM residency of the
// libraries, the cascade driver (one match launch per bowtie command line of
// RAP:577-599 / RAP:688, survivor lists kept on device) and the tally launch.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library is bound at run time (see RcclApi)

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mirge_amd.h"
#include "dict_index.hpp"
#include "pgzip.hpp"
#include "fastq.hpp"
#include "fm_index.hpp"
#include "kernels.hpp"
#include "tables.hpp"

struct mrg_index {
  mutable mrg::FmIndex ix;
  // A large library's jump tables and row context are built when somebody needs them on the HOST (mrg_index_get_view, a
  // context without `device_tables`): fm_index.hpp, derive_tables.
  mutable std::mutex derive_mutex;
  const mrg::FmIndex& derived() const {
    std::lock_guard<std::mutex> g(derive_mutex);
    mrg::derive_tables(ix);
    return ix;
  }
  // exact-match dictionaries by key length, built on first request (mrg_index_get_dict,
  // mrg_ctx_add_library) and kept for the life of the index
  mutable std::map<uint32_t, mrg::ExactDict> dicts;
  mutable std::mutex dict_mutex;
  const mrg::ExactDict& dict(uint32_t key_bases) const {
    std::lock_guard<std::mutex> g(dict_mutex);
    auto it = dicts.find(key_bases);
    if (it == dicts.end()) {
      mrg::ExactDict d;
      mrg::build_exact_dict(ix, key_bases, d);
      it = dicts.emplace(key_bases, std::move(d)).first;
    }
    return it->second;
  }
  void drop_dict(uint32_t key_bases) const {  // (a large library's slot array is gigabytes: not kept on the host once uploaded)
    std::lock_guard<std::mutex> g(dict_mutex);
    dicts.erase(key_bases);
  }
};
struct mrg_fastq {
  mrg::FastqData d;
};
struct mrg_gz {
  std::unique_ptr<mrg::GzipReader> rd;
};

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(MRG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                             \
  } while (0)

struct DevLib {
  uint32_t *blocks = nullptr, *super = nullptr, *text = nullptr, *seg_start = nullptr,
           *seg_ref = nullptr, *seg_off = nullptr, *chunk_seg = nullptr;
  uint64_t* sa = nullptr;
  uint32_t* ctx = nullptr;
  uint32_t* sa16 = nullptr;  // wide rows of a large library (fm_index.hpp: fill_wide_rows)
  bool tables_on_device = false;  // jump tables / row context / wide rows / buckets were filled by libtables.hip
  uint32_t* buckets = nullptr;  // seed buckets (fm_index.hpp: fill_seed_buckets), 128 B per k-mer of bucket_k bases
  uint32_t bucket_k = 0;
  uint32_t* pos_rows = nullptr;  // position lists of the k-mers whose bucket overflows (fm_index.hpp: seed_pos_lists), 16-byte rows
  uint64_t pos_rows_n = 0;
  // pair tables of a small library (fm_index.hpp: PairTables), for 2-mismatch passes
  uint32_t* pair_jump = nullptr;
  uint64_t* pair_rows = nullptr;
  uint32_t pair_row_off[3] = {0, 0, 0};
  uint32_t pair_anchor = 0;
  uint32_t* pair_jump_s = nullptr;        // second set: anchors of pair_anchor - 1 bases
  uint32_t pair_row_off_s[3] = {0, 0, 0};  // (its row lists follow the first set's in pair_rows)
  // pair tables of a large library (pairs.hip), built on the device the first time a one-mismatch
  // sub-pass of a fused launch meets reads with short seed regions: three anchors, gaps A and 2A
  uint32_t* bpair_jump = nullptr;
  uint64_t* bpair_rows = nullptr;
  uint32_t bpair_row_off[3] = {0, 0, 0};
  uint32_t bpair_anchor = 0;
  bool bpair_failed = false;  // not enough free HBM: the pigeonhole pieces stay
  // exact-match dictionary (dict_index.hpp) of a library of at most dict_max_bases bases
  mrg::DictSlot* dict_slots = nullptr;
  uint32_t dict_log2 = 0, dict_key = 0;
  uint64_t dict_n_keys = 0, dict_n_overflow = 0;  // positions stored / left to the FM index (their home's chain overflowed)
  uint32_t* kbits = nullptr;
  std::vector<uint32_t> kbits_host;  // host copy (32 KB): the per-round interleaved tables are built from it
  std::vector<std::string> host_seqs;  // entries of a library of at most kDictSmallBases bases (for seed units over several libraries)
  uint32_t* ftab = nullptr;
  mrg::JumpTables tabs = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  uint32_t n = 0, nblk = 0, nsup = 0, primary = 0, text_words = 0, n_seg = 0, n_ref = 0;
  uint32_t max_ref_len = 0;
  bool simple = false;  // every entry is exactly one N-free segment: segment id == entry id, offset 0
};

template <class T>
int upload(T** dst, const std::vector<T>& v, size_t pad_to_multiple = 1) {
  size_t n = v.size();
  size_t n_alloc = ((n + pad_to_multiple - 1) / pad_to_multiple) * pad_to_multiple;
  if (n_alloc == 0) n_alloc = pad_to_multiple;
  HIP_TRY(hipMalloc((void**)dst, n_alloc * sizeof(T)));
  HIP_TRY(hipMemset(*dst, 0, n_alloc * sizeof(T)));
  if (n) HIP_TRY(hipMemcpy(*dst, v.data(), n * sizeof(T), hipMemcpyHostToDevice));
  return MRG_OK;
}

constexpr uint32_t kStatsPerPass = 5;
constexpr uint32_t kWalkCap = 256;  // wave_seed_kernel: records a wave may leave behind its stream (in the cascade workspace)

void free_dev_lib(DevLib& l);

// What a seed_kernel unit searches: one library, or several small ones searched with the same
// policy as ONE index of their concatenation (entries of member j start at entry_lo[j]).  Built
// the first time a cascade plans such a unit, kept for the life of the context.
struct SeedLib {
  std::string key;     // member library ids, e.g. "2,4,5"
  DevLib lib;          // FM arrays (with 16-byte rows)
  bool owned = false;  // false: `lib` is a copy of the pointers of a library of the context
  std::vector<uint32_t> entry_lo;
  uint32_t* kbits = nullptr;  // presence bitmaps of the 8-, 9-, 10- and 11-mers (owned units of small libraries)
};

}  // namespace

namespace {
// RCCL is resolved with dlopen/dlsym at the first mrg_comm_* call: a process whose host side is
// PyTorch has torch's own librccl loaded already (that copy is reused, one RCCL per process), a
// torch-less host gets the system library.  Nothing links against it, so single-GPU users need no
// RCCL at all.
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return &api;
  tried = true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names)  // a copy that is already mapped (torch's) wins
    if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  for (size_t i = 0; !api.handle && i < sizeof names / sizeof *names; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!api.handle) {
    api.error = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?");
    return &api;
  }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
  api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
    api.error = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
    api.handle = nullptr;
  }
  return &api;
}
}  // namespace

struct mrg_ctx {
  int device = 0;
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  int n_cu = 0;
  uint64_t hbm_bytes = 0;
  std::string arch;
  int64_t lds_budget = 160 * 1024;
  int64_t wstop = 8;
  int64_t use_ftab = 1;
  int64_t force_lds_mode = -1;
  int64_t wide_rows = 64;
  int64_t hint_min_len = 0, hint_max_len = 255;  // length range of the reads of the NEXT run only
  uint64_t run_id = 0;
  // interleaved 9-mer tables of fused rounds, keyed by "lib ids | log2 | bits" (built on first use)
  std::vector<std::pair<std::string, uint32_t*>> round_tables;
  void* scratch = nullptr;  // context-owned device scratch (grown on demand; mrg_list_best_count)
  uint64_t scratch_bytes = 0;
  int64_t kmer_filter = 1;
  int64_t ctx_wide_rows = 32;
  int64_t prefer_two_blocks = 1;
  // 0 = one launch per pass; 1 = consecutive passes with at most one seed mismatch share a launch,
  // small (bitmap-filtered) and large libraries in separate groups; 2 = one group regardless of
  // library size; 3 = only the small-library runs are fused
  int64_t fuse = 1;
  int64_t round_large = 0;
  int64_t split_strata = 1;
  int64_t pair_big = 5;      // anchor length of the pair tables of large libraries (one-mismatch sub-passes of fused launches, reads with 3A..4A-1 seed bases); 0 = off
  int64_t pair_seeds = 1;    // 2-mismatch passes on small libraries search through anchor pairs (set before add_library)
  int64_t stratum_rows = 0;  // (measured slower than match_kernel: 2.43 vs 2.28 ms) 1: the last stratum of a split 2-mismatch pass runs stratum_kernel; 2: every strata launch does
  int64_t wide_rows_16 = 1;  // libraries of >= 2^20 bases get 16-byte rows with 32 bases of context
  int64_t dict = 1;          // one-word batches without N run the dictionary kernels (dict.hip) where a pass can
  int64_t dict_key = 16;     // key length of the exact-match dictionaries (set before add_library)
  // libraries up to this size get an exact-match dictionary (set before add_library).  The default covers the
  // small libraries; a host that will run passes WITHOUT seed mismatch on a large one (mRNA `-n 0`) raises it
  // for that library (16 B x 2..4 slots per base of HBM), as mirge_amd/engine.py does
  int64_t dict_max_bases = (int64_t)mrg::kDictSmallBases;
  int64_t split_mixed = 1;    // a batch with long reads / reads with N: its one-word N-free reads take the dictionary kernels
  int64_t split_min_len = 16;  // ... and so do not reads shorter than this (the reference's own minimum length, trim_file.py:33; shorter seed regions than 15 bases have no pair tables)
  int64_t stratum0_unit = 0;  // (measured slower, default off) the exact stratum of a 2-mismatch pass behind a seed launch rides in that launch
  int64_t walk_diag = 0;
  int64_t count_variants = 1;  // mrg_count_best, one seed mismatch: the jump-table variants kernel in front of the pigeonhole kernel
  int64_t long_lane = 0;   // round 6: 1 = the reads of 33..63 nt of a split batch ride the dictionary kernels too (their LONG instantiations; measured no faster than the FM kernels: off)
  int64_t walk_cap = 256;  // records a wave may leave behind its stream (<= 256; tests shrink it: beyond it a seed is verified row by row)
  int64_t pos_scan = 1;   // 0 at run time: the seed launches verify a wide interval row by row as before round 6
  int64_t pos_lists = 1;  // overflowing seed buckets get their rows in text order too (set before add_library)
  int64_t seed_buckets = 1;  // large libraries get seed buckets where they pay (set before add_library); 0 at run time: not used
  int64_t seed_wgs = 0;      // seed_kernel workgroups (256 threads) per CU; 0 = what the launch's instantiation keeps resident
  int64_t seed_units = 1;    // ... and their runs of passes with at most one seed mismatch go through seed_kernel
  int64_t pair_impl = 1;   // 1 = pair_wave_kernel for the anchor-pair search of one-word batches, 0 = stratum_kernel
  int64_t grid_pct = 100;  // share of the workgroups every cascade launch gets (see scale_grid)
  // "fused_step": a caller that follows EVERY cascade with a tally on the same stream (bench.py's step, the command line):
  // no per-pass events are recorded (mrg_pass_stats.ms = 0) and d_pass_counts is written by the tally launch
  int64_t fused_step = 0;
  uint64_t* pending_export_out = nullptr;
  hipStream_t pending_export_stream = nullptr;
  int64_t device_tables = 1;  // mrg_ctx_add_library: a large library's derived tables (dictionary, wide rows, seed buckets) are filled on the device
  int64_t collapse_fast = 1;  // mrg_collapse_run: batches that fit it take the duplication-aware path (0: always the general sort)
  int64_t seed_impl = -1;  // -1 = per launch (run_seed), 0 = seed_kernel (tiles), 1 = wave_seed_kernel, 2 = the same with more registers
  std::vector<DevLib> libs;
  std::vector<std::unique_ptr<SeedLib>> seed_libs;
  // last run
  hipStream_t last_stream = nullptr;
  uint64_t* last_stats_dev = nullptr;
  uint32_t last_n_pass = 0;
  uint32_t last_lds[MRG_MAX_PASSES] = {0};
  uint32_t last_mode[MRG_MAX_PASSES] = {0};
  uint32_t last_group[MRG_MAX_PASSES] = {0};
  uint32_t last_launches[MRG_MAX_PASSES] = {0};
  uint32_t last_split = 0;  // the last run split its batch into one-word reads and the rest
  uint32_t last_kbits_log2[MRG_MAX_PASSES] = {0};
  uint32_t last_pair_anchor[MRG_MAX_PASSES] = {0};
  uint32_t last_variant[MRG_MAX_PASSES] = {0};
  hipEvent_t ev[MRG_MAX_PASSES + 1] = {nullptr};
  hipEvent_t ev0[MRG_MAX_PASSES + 1] = {nullptr};  // the first cascade of a split batch
  // which event holds the time of pass boundary i: a boundary with no launch since the one before shares its event
  // (an event record costs ~5 us of an idle GPU: five of them less per step)
  uint8_t ev_ix[MRG_MAX_PASSES + 1] = {0}, ev0_ix[MRG_MAX_PASSES + 1] = {0};
  bool ev_ready = false;
  bool last_timed = true;  // the last run recorded its per-pass events
};

extern "C" {

int mrg_version(void) { return 100; }
const char* mrg_last_error(void) { return g_err.c_str(); }

// ----------------------------------------------------------------- index
int mrg_index_build(const char* const* names, const char* const* seqs, uint32_t n_ref,
                    mrg_index** out) {
  if (!out || (n_ref && (!names || !seqs))) return fail(MRG_ERR_ARG, "mrg_index_build: null argument");
  try {
    std::vector<std::string> nv(n_ref), sv(n_ref);
    for (uint32_t i = 0; i < n_ref; ++i) {
      nv[i] = names[i];
      sv[i] = seqs[i];
    }
    auto h = std::make_unique<mrg_index>();
    mrg::build_index(nv, sv, h->ix);
    *out = h.release();
    return MRG_OK;
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_index_build: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_FORMAT, "mrg_index_build: %s", e.what());
  }
}

int mrg_index_build_fasta(const char* fasta_path, mrg_index** out) {
  if (!fasta_path || !out) return fail(MRG_ERR_ARG, "mrg_index_build_fasta: null argument");
  try {
    std::vector<std::string> nv, sv;
    mrg::read_fasta(fasta_path, nv, sv);
    auto h = std::make_unique<mrg_index>();
    mrg::build_index(nv, sv, h->ix);
    *out = h.release();
    return MRG_OK;
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_index_build_fasta: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_index_build_fasta: %s", e.what());
  }
}

int mrg_index_build_ebwt(const char* prefix, mrg_index** out) {
  if (!prefix || !out) return fail(MRG_ERR_ARG, "mrg_index_build_ebwt: null argument");
  try {
    std::vector<std::string> nv, sv;
    mrg::read_ebwt(prefix, nv, sv);
    auto h = std::make_unique<mrg_index>();
    mrg::build_index(nv, sv, h->ix);
    *out = h.release();
    return MRG_OK;
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_index_build_ebwt: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_index_build_ebwt: %s", e.what());
  }
}

int mrg_index_save(const mrg_index* ix, const char* path) {
  if (!ix || !path) return fail(MRG_ERR_ARG, "mrg_index_save: null argument");
  try {
    mrg::save_index(ix->ix, path);
    return MRG_OK;
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_index_save: %s", e.what());
  }
}

int mrg_index_load(const char* path, mrg_index** out) {
  if (!path || !out) return fail(MRG_ERR_ARG, "mrg_index_load: null argument");
  try {
    auto h = std::make_unique<mrg_index>();
    mrg::load_index(path, h->ix);
    *out = h.release();
    return MRG_OK;
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_index_load: %s", e.what());
  }
}

void mrg_index_free(mrg_index* ix) { delete ix; }

int mrg_index_get_info(const mrg_index* h, mrg_index_info* info) {
  if (!h || !info) return fail(MRG_ERR_ARG, "mrg_index_get_info: null argument");
  const mrg::FmIndex& ix = h->ix;
  info->n_ref = (uint32_t)ix.names.size();
  info->n_seg = (uint32_t)ix.seg_ref.size();
  info->n_bases = ix.n;
  info->n_blocks = (uint32_t)ix.blocks.size();
  info->n_super = (uint32_t)(ix.super.size() / 4);
  info->primary = ix.primary;
  info->text_words = (uint32_t)ix.text.size();
  for (int t = 0; t < 4; ++t) info->ftab_ks[t] = ix.ftab_ks[t];
  for (int c = 0; c < 4; ++c) info->C[c] = ix.C[c];
  info->bytes_fm = (uint64_t)ix.blocks.size() * 16 + (uint64_t)ix.super.size() * 4;
  info->bytes_sa = (uint64_t)ix.sa.size() * 8;
  return MRG_OK;
}

int mrg_index_name(const mrg_index* h, uint32_t i, const char** name) {
  if (!h || !name) return fail(MRG_ERR_ARG, "mrg_index_name: null argument");
  if (i >= h->ix.names.size()) return fail(MRG_ERR_ARG, "mrg_index_name: entry %u out of range", i);
  *name = h->ix.names[i].c_str();
  return MRG_OK;
}

int mrg_index_seq(const mrg_index* h, uint32_t i, char* buf, uint32_t cap, uint32_t* len) {
  if (!h || !len) return fail(MRG_ERR_ARG, "mrg_index_seq: null argument");
  if (i >= h->ix.names.size()) return fail(MRG_ERR_ARG, "mrg_index_seq: entry %u out of range", i);
  *len = h->ix.ref_len[i];
  if (!buf) return MRG_OK;
  if (cap < *len + 1) return fail(MRG_ERR_ARG, "mrg_index_seq: buffer too small (%u < %u)", cap, *len + 1);
  std::string s = mrg::entry_sequence(h->ix, i);
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return MRG_OK;
}

int mrg_index_get_view(const mrg_index* h, mrg_index_view* v) {
  if (!h || !v) return fail(MRG_ERR_ARG, "mrg_index_get_view: null argument");
  const mrg::FmIndex& ix = h->derived();
  v->blocks = reinterpret_cast<const uint32_t*>(ix.blocks.data());
  v->super = ix.super.data();
  v->text = ix.text.data();
  v->sa = ix.sa.data();
  v->ftab = ix.ftab.data();
  v->ctx = ix.ctx.empty() ? nullptr : ix.ctx.data();
  v->kbits = ix.kbits.empty() ? nullptr : ix.kbits.data();
  v->seg_start = ix.seg_start.data();
  v->seg_ref = ix.seg_ref.data();
  v->seg_off = ix.seg_off.data();
  v->chunk_seg = ix.chunk_seg.data();
  return MRG_OK;
}

int mrg_index_get_dict(const mrg_index* h, uint32_t key_bases, mrg_dict_view* v) {
  if (!h || !v) return fail(MRG_ERR_ARG, "mrg_index_get_dict: null argument");
  try {
    const mrg::ExactDict& d = h->dict(key_bases);
    v->slots = reinterpret_cast<const uint64_t*>(d.slots.data());
    v->log2_slots = d.log2_slots;
    v->key_bases = d.key_bases;
    v->n_keys = d.n_keys;
    v->n_overflow = d.n_overflow;
    return MRG_OK;
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_index_get_dict: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_ARG, "mrg_index_get_dict: %s", e.what());
  }
}

// --------------------------------------------------------------- context
int mrg_ctx_create(int device, mrg_ctx** out) {
  if (!out) return fail(MRG_ERR_ARG, "mrg_ctx_create: null argument");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(MRG_ERR_NO_DEVICE,
                "mrg_ctx_create: no HIP device visible (%s); this engine has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= count)
    return fail(MRG_ERR_NO_DEVICE, "mrg_ctx_create: device %d not in [0,%d)", device, count);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  auto ctx = std::make_unique<mrg_ctx>();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  ctx->hbm_bytes = prop.totalGlobalMem;
  ctx->arch = prop.gcnArchName;
  if (ctx->arch.rfind("gfx950", 0) != 0)
    return fail(MRG_ERR_NO_DEVICE, "mrg_ctx_create: device %d is %s; kernels are built for gfx950 only",
                device, ctx->arch.c_str());
  ctx->lds_budget = (int64_t)prop.sharedMemPerBlock;  // 160 KiB on gfx950
  if (ctx->lds_budget > 160 * 1024) ctx->lds_budget = 160 * 1024;
  *out = ctx.release();
  return MRG_OK;
}

void mrg_ctx_destroy(mrg_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (DevLib& l : ctx->libs) free_dev_lib(l);
  for (auto& sl : ctx->seed_libs) {
    if (sl->owned) free_dev_lib(sl->lib);
    (void)hipFree(sl->kbits);
  }
  if (ctx->ev_ready)
  {
    for (auto& e : ctx->ev) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev0) (void)hipEventDestroy(e);
  }
  if (ctx->comm) (void)rccl_api()->CommDestroy(ctx->comm);
  (void)hipFree(ctx->scratch);
  for (auto& kv : ctx->round_tables) (void)hipFree(kv.second);
  delete ctx;
}

namespace {
void free_dev_lib(DevLib& l) {
  void* ptrs[] = {l.blocks, l.super, l.text, l.sa, l.ftab, l.ctx, l.sa16, l.buckets, l.pos_rows, l.kbits, l.pair_jump, l.pair_jump_s, l.pair_rows, l.dict_slots,
                  l.bpair_jump, l.bpair_rows, l.seg_start, l.seg_ref, l.seg_off, l.chunk_seg};
  for (void* p : ptrs) (void)hipFree(p);
}

// The FM arrays of one index into HBM.  wide_rows: also the 16-byte rows; pair_tables: also the
// anchor-pair tables of 2-mismatch passes.
int upload_index(mrg_ctx* ctx, const mrg::FmIndex& ix, DevLib& l, bool wide_rows, bool pair_tables) {
  mrg::StageTimer tm("upload_index");
  int rc;
  l.n = ix.n;
  l.nblk = (uint32_t)ix.blocks.size();
  l.primary = ix.primary;
  l.n_seg = (uint32_t)ix.seg_ref.size();
  l.n_ref = (uint32_t)ix.names.size();
  l.nsup = (uint32_t)(ix.super.size() / 4);
  // (not just "as many segments as entries": an all-N entry and one with an inner N run would
  // also give equal counts)
  l.simple = l.n_seg == l.n_ref;
  for (uint32_t sg = 0; l.simple && sg < l.n_seg; ++sg) l.simple = ix.seg_ref[sg] == sg && ix.seg_off[sg] == 0;
  for (uint32_t v : ix.ref_len) l.max_ref_len = std::max(l.max_ref_len, v);
  std::vector<uint32_t> blk(reinterpret_cast<const uint32_t*>(ix.blocks.data()),
                            reinterpret_cast<const uint32_t*>(ix.blocks.data()) + ix.blocks.size() * 4);
  if ((rc = upload(&l.blocks, blk, 4))) return rc;
  if ((rc = upload(&l.super, ix.super, 4))) return rc;
  // text is staged into LDS 16 B at a time: round its word count up to 4
  l.text_words = (uint32_t)((ix.text.size() + 3) / 4 * 4);
  if ((rc = upload(&l.text, ix.text, 4))) return rc;
  if ((rc = upload(&l.sa, ix.sa))) return rc;
  {
    // ascending k for the kernels; ix.ftab_ks is in storage order (largest first, 0 = absent)
    uint32_t off = 0, offs[4];
    for (int t = 0; t < 4; ++t) {
      offs[t] = off;
      if (ix.ftab_ks[t]) off += (1u << (2 * ix.ftab_ks[t])) + 1u;
    }
    for (int i = 0; i < 4; ++i) {
      int t = 3 - i;
      if (!ix.ftab_ks[t]) t = 1;  // no big table: the main one again
      l.tabs.k[i] = ix.ftab_ks[t];
      l.tabs.off[i] = offs[t];
    }
  }
  const bool on_device = ctx->device_tables && ix.n >= mrg::kLazyDeriveBases;
  if (!on_device && !ix.derived)
    return fail(MRG_ERR_ARG, "upload_index: the index's jump tables are not derived (derive_tables before the upload)");
  if (on_device) {
    // jump tables, row context and wide rows from the rows and the text just uploaded (libtables.hip)
    size_t total = 0;
    for (int t = 0; t < 4; ++t)
      if (ix.ftab_ks[t]) total += ((size_t)1 << (2 * ix.ftab_ks[t])) + 1;
    tm.lap("blocks, text, sa");
    void* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&l.ftab, total * 4));
    HIP_TRY(hipMalloc(&tmp, mrg::jump_tables_device_temp_bytes(ix.n + 1u)));
    hipError_t e = mrg::build_jump_tables_device(l.text, l.text_words, reinterpret_cast<const uint64_t*>(l.sa), ix.n, ix.ftab_ks, l.ftab, tmp, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(MRG_ERR_HIP, "mrg_ctx_add_library: jump tables on the device: %s", hipGetErrorString(e));
    HIP_TRY(hipMalloc((void**)&l.ctx, (size_t)(ix.n + 1u) * 4));
    HIP_TRY(mrg::build_row_context_device(l.text, l.text_words, reinterpret_cast<const uint64_t*>(l.sa), ix.n, l.ctx, nullptr));
    if (wide_rows) {
      HIP_TRY(hipMalloc((void**)&l.sa16, (size_t)(ix.n + 1u) * 16));
      HIP_TRY(mrg::build_wide_rows_device(l.text, l.text_words, reinterpret_cast<const uint64_t*>(l.sa), ix.n, l.sa16, nullptr));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    l.tables_on_device = true;
    tm.lap("jump tables, row context, wide rows (device)");
  } else {
    if ((rc = upload(&l.ftab, ix.ftab))) return rc;
    if (!ix.ctx.empty() && (rc = upload(&l.ctx, ix.ctx))) return rc;
    tm.lap("blocks, text, sa, jump tables, ctx");
  }
  if (wide_rows && !on_device) {
    // 16-byte rows for the fused launches: filled on the host in chunks, never kept there
    const size_t n_rows = ix.sa.size(), chunk = 1u << 24;
    HIP_TRY(hipMalloc((void**)&l.sa16, n_rows * 16));
    std::vector<uint32_t> buf;
    for (size_t lo = 0; lo < n_rows; lo += chunk) {
      const size_t hi = std::min(n_rows, lo + chunk);
      buf.resize((hi - lo) * 4);
      mrg::fill_wide_rows(ix, lo, hi, buf.data());
      HIP_TRY(hipMemcpy(l.sa16 + lo * 4, buf.data(), (hi - lo) * 16, hipMemcpyHostToDevice));
    }
  }
  tm.lap("wide rows");
  if (pair_tables && ix.n <= mrg::kPairMaxBases && ix.n >= 4u * mrg::kPairAnchor) {
    mrg::PairTables pt, pt_s;
    try {
      mrg::build_pair_tables(ix, mrg::kPairAnchor, pt);
      mrg::build_pair_tables(ix, mrg::kPairAnchor - 1u, pt_s);
    } catch (const std::exception& e) {
      return fail(MRG_ERR_ARG, "mrg_ctx_add_library: %s", e.what());
    }
    if ((rc = upload(&l.pair_jump, pt.jump))) return rc;
    if ((rc = upload(&l.pair_jump_s, pt_s.jump))) return rc;
    for (int t = 0; t < 3; ++t) {
      l.pair_row_off[t] = pt.row_off[t];
      l.pair_row_off_s[t] = pt.row_off[3] + pt_s.row_off[t];
    }
    pt.rows.insert(pt.rows.end(), pt_s.rows.begin(), pt_s.rows.end());
    if ((rc = upload(&l.pair_rows, pt.rows))) return rc;
    l.pair_anchor = pt.anchor;
  }
  tm.lap("pair tables");
  if (!ix.kbits.empty() && (rc = upload(&l.kbits, ix.kbits))) return rc;
  l.kbits_host = ix.kbits;  // the interleaved tables of fused rounds are built from it
  if ((rc = upload(&l.seg_start, ix.seg_start))) return rc;
  if ((rc = upload(&l.seg_ref, ix.seg_ref))) return rc;
  if ((rc = upload(&l.seg_off, ix.seg_off))) return rc;
  if ((rc = upload(&l.chunk_seg, ix.chunk_seg))) return rc;
  return MRG_OK;
}
}  // namespace

int mrg_ctx_add_library(mrg_ctx* ctx, const mrg_index* h, int32_t* lib_id) {
  if (!ctx || !h || !lib_id) return fail(MRG_ERR_ARG, "mrg_ctx_add_library: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const bool tables_dev = ctx->device_tables && h->ix.n >= mrg::kLazyDeriveBases;
  const mrg::FmIndex& ix = tables_dev ? h->ix : h->derived();
  DevLib l;
  struct Guard {  // a failed upload must not leak the arrays uploaded before it
    DevLib* l;
    ~Guard() {
      if (l) free_dev_lib(*l);
    }
  } guard{&l};
  mrg::StageTimer tm("add_library");
  int rc = upload_index(ctx, ix, l, ix.n >= mrg::kWideRowMinBases && ctx->wide_rows_16, ctx->pair_seeds != 0);
  if (rc) return rc;
  tm.lap("index arrays + wide rows + pair tables");
  // Exact-match dictionary: every library of at most dict_max_bases bases whose slot array (16 B x 2..4
  // slots per base: 8.6 GB for the 137 Mbp mRNA library) leaves 8 GB of this GPU's HBM free.  A pass
  // without seed mismatch on it -- mRNA `-n 0`, RAP:584/598 -- is then ONE 16-byte gather per read
  // instead of a jump-table line and a wide-row line.
  bool want_dict = ctx->dict && ix.n <= (uint64_t)ctx->dict_max_bases && ix.n <= mrg::kDictMaxBases && ix.n >= (uint32_t)ctx->dict_key;
  if (want_dict && ix.n > mrg::kDictSmallBases) {
    const uint64_t need = mrg::exact_dict_bytes(ix, (uint32_t)ctx->dict_key);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    want_dict = need != 0 && free_b > need + (8ull << 30);
  }
  if (want_dict && ctx->device_tables && ix.n > mrg::kDictSmallBases) {
    // a large library's dictionary is filled on the device, from the text and segment tables uploaded above
    // (dictbuild.hip): no host build, no 8.6 GB upload
    const uint64_t bytes = mrg::exact_dict_bytes(ix, (uint32_t)ctx->dict_key);
    uint32_t l2 = 0;
    while (bytes && (16ull << l2) < bytes) ++l2;
    if (bytes && mrg::exact_dict_device_ok(ix.n, (uint32_t)ctx->dict_key, l2)) {
      void* tmp = nullptr;
      HIP_TRY(hipMalloc((void**)&l.dict_slots, bytes));
      hipError_t e = hipMalloc(&tmp, mrg::exact_dict_device_temp_bytes(ix.n));
      uint64_t counts[2] = {0, 0};
      if (e == hipSuccess)
        e = mrg::build_exact_dict_device(l.text, ix.n, l.seg_start, l.seg_ref, l.seg_off, l.chunk_seg, (uint32_t)ctx->dict_key, l2, l.dict_slots, tmp,
                                         counts, nullptr);
      (void)hipFree(tmp);
      if (e != hipSuccess) return fail(MRG_ERR_HIP, "mrg_ctx_add_library: dictionary build on the device: %s", hipGetErrorString(e));
      l.dict_log2 = l2;
      l.dict_key = (uint32_t)ctx->dict_key;
      l.dict_n_keys = counts[0];
      l.dict_n_overflow = counts[1];
      want_dict = false;
    }
  }
  if (want_dict) {
    const mrg::ExactDict* ed = nullptr;
    try {
      ed = &h->dict((uint32_t)ctx->dict_key);
    } catch (const std::exception&) {
      ed = nullptr;  // entries too long for the slot format: the FM kernels serve this library
    }
    if (ed) {
      if ((rc = upload(&l.dict_slots, ed->slots))) return rc;
      l.dict_log2 = ed->log2_slots;
      l.dict_key = ed->key_bases;
      l.dict_n_keys = ed->n_keys;
      l.dict_n_overflow = ed->n_overflow;
      if (ix.n > mrg::kDictSmallBases) h->drop_dict((uint32_t)ctx->dict_key);
    }
  }
  tm.lap("dictionary");
  if (ctx->dict && ctx->seed_buckets && l.sa16) {
    // large library where a seed of 11 bases has a few rows: those rows in one line per seed
    const uint32_t bk = mrg::seed_bucket_k(ix);
    if (bk) {
      const uint64_t n_codes = 1ull << (2 * bk), chunk = 1ull << 18;
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      if (l.tables_on_device && free_b > n_codes * 128ull + (4ull << 30) && hipMalloc((void**)&l.buckets, n_codes * 128ull) == hipSuccess) {
        size_t tab_base = 0;
        for (int t = 0; t < 4 && ix.ftab_ks[t] != bk; ++t)
          if (ix.ftab_ks[t]) tab_base += ((size_t)1 << (2 * ix.ftab_ks[t])) + 1;
        HIP_TRY(mrg::build_seed_buckets_device(l.text, l.text_words, reinterpret_cast<const uint64_t*>(l.sa), ix.n, l.ftab + tab_base, bk, l.buckets, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        l.bucket_k = bk;
      } else if (!l.tables_on_device && free_b > n_codes * 128ull + (4ull << 30) && hipMalloc((void**)&l.buckets, n_codes * 128ull) == hipSuccess) {
        std::vector<uint32_t> buf(chunk * 32);
        for (uint64_t lo = 0; lo < n_codes; lo += chunk) {
          const uint64_t hi = std::min(n_codes, lo + chunk);
          mrg::fill_seed_buckets(ix, bk, lo, hi, buf.data());
          HIP_TRY(hipMemcpy(l.buckets + lo * 32, buf.data(), (hi - lo) * 128, hipMemcpyHostToDevice));
        }
        l.bucket_k = bk;
      } else {
        (void)hipGetLastError();
        l.buckets = nullptr;
      }
      if (l.buckets && l.bucket_k && ctx->pos_lists) {
        // the k-mers whose bucket overflows (repeats, poly-A, tandem motifs): their rows in text order (round 6)
        size_t tab_base = 0;
        for (int t = 0; t < 4 && ix.ftab_ks[t] != bk; ++t)
          if (ix.ftab_ks[t]) tab_base += ((size_t)1 << (2 * ix.ftab_ks[t])) + 1;
        hipError_t e = mrg::build_seed_pos_lists_device(l.text, l.text_words, reinterpret_cast<const uint64_t*>(l.sa), ix.n, l.ftab + tab_base, bk,
                                                        l.buckets, &l.pos_rows, &l.pos_rows_n, nullptr);
        if (e != hipSuccess) return fail(MRG_ERR_HIP, "mrg_ctx_add_library: position lists of the seed buckets: %s", hipGetErrorString(e));
      }
    }
  }
  // small libraries keep their entries on the host: passes that search several of them with one
  // policy get one index of their concatenation (seed_kernel units), built when a cascade first asks
  if (ctx->dict && ix.n <= mrg::kDictSmallBases) {
    l.host_seqs.resize(ix.names.size());
    for (uint32_t i = 0; i < l.host_seqs.size(); ++i) l.host_seqs[i] = mrg::entry_sequence(ix, i);
  }
  tm.lap("seed buckets + host entries");
  ctx->libs.push_back(l);
  guard.l = nullptr;
  *lib_id = (int32_t)ctx->libs.size() - 1;
  return MRG_OK;
}

int mrg_ctx_set_option(mrg_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return fail(MRG_ERR_ARG, "mrg_ctx_set_option: null argument");
  std::string k(key);
  if (k == "lds_budget") {
    if (value < 0 || value > 160 * 1024) return fail(MRG_ERR_ARG, "lds_budget must be in [0,163840]");
    ctx->lds_budget = value;
  } else if (k == "wstop") {
    if (value < 0) return fail(MRG_ERR_ARG, "wstop must be >= 0");
    ctx->wstop = value;
  } else if (k == "ctx_wide_rows") {
    if (value < 1) return fail(MRG_ERR_ARG, "ctx_wide_rows must be >= 1");
    ctx->ctx_wide_rows = value;
  } else if (k == "kmer_filter") {
    ctx->kmer_filter = value != 0;
  } else if (k == "hint_min_len") {
    ctx->hint_min_len = value;
  } else if (k == "hint_max_len") {
    ctx->hint_max_len = value;
  } else if (k == "wide_rows") {
    if (value < 1) return fail(MRG_ERR_ARG, "wide_rows must be >= 1");
    ctx->wide_rows = value;
  } else if (k == "prefer_two_blocks") {
    ctx->prefer_two_blocks = value != 0;
  } else if (k == "force_lds_mode") {
    if (value < -1 || value > 3) return fail(MRG_ERR_ARG, "force_lds_mode must be in [-1,3]");
    ctx->force_lds_mode = value;
  } else if (k == "ftab") {
    ctx->use_ftab = value != 0;
  } else if (k == "split_strata") {
    ctx->split_strata = value != 0;
  } else if (k == "pair_big") {
    if (value != 0 && (value < 4 || value > 7)) return fail(MRG_ERR_ARG, "pair_big must be 0 or in [4,7]");
    ctx->pair_big = value;
  } else if (k == "pair_seeds") {
    ctx->pair_seeds = value != 0;
  } else if (k == "stratum_rows") {
    if (value < 0 || value > 2) return fail(MRG_ERR_ARG, "stratum_rows must be in [0,2]");
    ctx->stratum_rows = value;
  } else if (k == "round_large") {
    ctx->round_large = value != 0;
  } else if (k == "wide_rows_16") {
    ctx->wide_rows_16 = value != 0;  // takes effect for libraries added afterwards
  } else if (k == "dict") {
    ctx->dict = value != 0;  // (the dictionaries themselves are built by mrg_ctx_add_library while this is 1)
  } else if (k == "split_mixed") {
    ctx->split_mixed = value != 0;
  } else if (k == "split_min_len") {
    if (value < 0 || value > 32) return fail(MRG_ERR_ARG, "mrg_ctx_set_option: split_min_len must be in [0,32]");
    ctx->split_min_len = value;
  } else if (k == "stratum0_unit") {
    ctx->stratum0_unit = value != 0;
  } else if (k == "seed_buckets") {
    ctx->seed_buckets = value != 0;
  } else if (k == "pos_lists") {
    ctx->pos_lists = value != 0;
  } else if (k == "pos_scan") {
    ctx->pos_scan = value != 0;
  } else if (k == "count_variants") {
    ctx->count_variants = value != 0;
  } else if (k == "walk_cap") {
    if (value < 0 || value > (int64_t)kWalkCap) return fail(MRG_ERR_ARG, "walk_cap must be in [0,256]");
    ctx->walk_cap = value;
  } else if (k == "walk_diag") {
    ctx->walk_diag = value;
  } else if (k == "long_lane") {
    ctx->long_lane = value != 0;
  } else if (k == "seed_wgs") {
    ctx->seed_wgs = value;
  } else if (k == "pair_impl") {
    ctx->pair_impl = value != 0;
  } else if (k == "grid_pct") {
    if (value < 1 || value > 100) return fail(MRG_ERR_ARG, "grid_pct must be in [1,100]");
    ctx->grid_pct = value;
  } else if (k == "seed_impl") {
    if (value < -1 || value > 2) return fail(MRG_ERR_ARG, "seed_impl must be in [-1,2]");
    ctx->seed_impl = value;
  } else if (k == "seed_units") {
    ctx->seed_units = value != 0;
  } else if (k == "fused_step") {
    ctx->fused_step = value != 0;
  } else if (k == "collapse_fast") {
    ctx->collapse_fast = value != 0;
  } else if (k == "device_tables") {
    ctx->device_tables = value != 0;  // takes effect for libraries added afterwards
  } else if (k == "dict_max_bases") {
    if (value < 0) return fail(MRG_ERR_ARG, "dict_max_bases must be >= 0");
    ctx->dict_max_bases = value;  // takes effect for libraries added afterwards
  } else if (k == "dict_key") {
    if (value < 8 || value > 16) return fail(MRG_ERR_ARG, "dict_key must be in [8,16]");
    ctx->dict_key = value;
  } else if (k == "fuse") {
    if (value < 0 || value > 3) return fail(MRG_ERR_ARG, "fuse must be in [0,3]");
    ctx->fuse = value;
  } else {
    return fail(MRG_ERR_ARG, "mrg_ctx_set_option: unknown key '%s'", key);
  }
  return MRG_OK;
}

int mrg_ctx_device_info(const mrg_ctx* ctx, int32_t* n_cu, uint64_t* hbm_bytes, char* arch,
                        uint32_t arch_cap) {
  if (!ctx) return fail(MRG_ERR_ARG, "mrg_ctx_device_info: null argument");
  if (n_cu) *n_cu = ctx->n_cu;
  if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
  if (arch && arch_cap) {
    std::snprintf(arch, arch_cap, "%s", ctx->arch.c_str());
  }
  return MRG_OK;
}

int mrg_ctx_library_stats(const mrg_ctx* ctx, int32_t lib, uint64_t* out4) {
  if (!ctx || !out4) return fail(MRG_ERR_ARG, "mrg_ctx_library_stats: null argument");
  if (lib < 0 || (size_t)lib >= ctx->libs.size()) return fail(MRG_ERR_ARG, "mrg_ctx_library_stats: unknown library %d", lib);
  const DevLib& l = ctx->libs[lib];
  out4[0] = l.dict_n_keys;
  out4[1] = l.dict_n_overflow;
  out4[2] = l.dict_slots ? l.dict_log2 : 0;
  out4[3] = l.buckets ? l.bucket_k : 0;
  return MRG_OK;
}

int mrg_ctx_library_check_tables(mrg_ctx* ctx, int32_t lib, const mrg_index* index, uint64_t* mismatches4) {
  if (!ctx || !index || !mismatches4) return fail(MRG_ERR_ARG, "mrg_ctx_library_check_tables: null argument");
  if (lib < 0 || (size_t)lib >= ctx->libs.size()) return fail(MRG_ERR_ARG, "mrg_ctx_library_check_tables: unknown library %d", lib);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  const DevLib& l = ctx->libs[lib];
  try {
    const mrg::FmIndex& ix = index->derived();
    if (ix.n != l.n) return fail(MRG_ERR_ARG, "mrg_ctx_library_check_tables: not the index this library was made from");
    for (int t = 0; t < 4; ++t) mismatches4[t] = ~0ull;
    std::vector<uint32_t> got, want;
    auto compare = [&](const uint32_t* dev, const uint32_t* host, size_t words) -> uint64_t {
      uint64_t bad = 0;
      const size_t chunk = (size_t)1 << 24;
      for (size_t lo = 0; lo < words; lo += chunk) {
        const size_t m = std::min(chunk, words - lo);
        got.resize(m);
        if (hipMemcpy(got.data(), dev + lo, m * 4, hipMemcpyDeviceToHost) != hipSuccess) return ~0ull - 1;
        for (size_t i = 0; i < m; ++i) bad += got[i] != host[lo + i];
      }
      return bad;
    };
    if (l.ftab) mismatches4[0] = compare(l.ftab, ix.ftab.data(), ix.ftab.size());
    if (l.ctx && !ix.ctx.empty()) mismatches4[1] = compare(l.ctx, ix.ctx.data(), ix.ctx.size());
    if (l.sa16) {
      uint64_t bad = 0;
      const size_t n_rows = ix.sa.size(), chunk = (size_t)1 << 22;
      for (size_t lo = 0; lo < n_rows; lo += chunk) {
        const size_t hi = std::min(n_rows, lo + chunk);
        want.resize((hi - lo) * 4);
        mrg::fill_wide_rows(ix, lo, hi, want.data());
        bad += compare(l.sa16 + lo * 4, want.data(), want.size());
      }
      mismatches4[2] = bad;
    }
    if (l.buckets && l.bucket_k) {
      uint64_t bad = 0;
      const uint64_t n_codes = 1ull << (2 * l.bucket_k), chunk = 1ull << 18;
      std::vector<uint32_t> over_start, over_pos;
      mrg::seed_pos_lists(ix, l.bucket_k, over_start, over_pos);
      if (!l.pos_rows && ctx->pos_lists && !over_pos.empty()) bad += 1;
      for (uint64_t lo = 0; lo < n_codes; lo += chunk) {
        const uint64_t hi = std::min(n_codes, lo + chunk);
        want.resize((hi - lo) * 32);
        mrg::fill_seed_buckets(ix, l.bucket_k, lo, hi, want.data(), l.pos_rows ? over_start.data() : nullptr);
        bad += compare(l.buckets + lo * 32, want.data(), want.size());
      }
      if (l.pos_rows) {  // the position lists themselves: every row = the wide row of the position the host lists there
        if (l.pos_rows_n != over_pos.size()) {
          bad += 1;
        } else {
          std::vector<uint64_t> row_of(ix.n + 1u);
          for (size_t i = 0; i < ix.sa.size(); ++i) row_of[(uint32_t)ix.sa[i]] = ix.sa[i];
          const size_t chunk_rows = (size_t)1 << 20;
          for (size_t r0 = 0; r0 < over_pos.size(); r0 += chunk_rows) {
            const size_t r1 = std::min(over_pos.size(), r0 + chunk_rows);
            want.resize((r1 - r0) * 4);
            for (size_t r = r0; r < r1; ++r) mrg::wide_row_of_row(ix, row_of[over_pos[r]], want.data() + 4 * (r - r0));
            bad += compare(l.pos_rows + r0 * 4, want.data(), want.size());
          }
        }
      }
      mismatches4[3] = bad;
    }
  } catch (const std::exception& e) {
    return fail(MRG_ERR_NOMEM, "mrg_ctx_library_check_tables: %s", e.what());
  }
  return MRG_OK;
}

int mrg_ctx_release_scratch(mrg_ctx* ctx) {
  if (!ctx) return fail(MRG_ERR_ARG, "mrg_ctx_release_scratch: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  if (ctx->scratch) HIP_TRY(hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  return MRG_OK;
}

// --------------------------------------------------------------- cascade
// workspace: [idx A][idx B][idx C]  (each n + kListSlack u32: segmented survivor lists; a producer
//            workgroup's segment holds every read of its chunks, 256, 1024 or 4096 reads each; C parks
//            the one-word reads of a split batch while the other reads run their cascade)
//            [segment counts A, B, C: kMaxSegments u32 each][stats: MRG_MAX_PASSES * 5 u64]
// (16 bytes per entry: the seed launches of a one-word batch write lists that carry their reads -- index, length, read --;
// exact_dict_kernel and the FM kernels keep 4-byte index lists in the same buffers)
static uint64_t ws_idx_bytes(uint64_t n) {
  return (((n + mrg::kListSlack) * 16 + 255) / 256) * 256;
}
static const uint64_t kWsCountsBytes = 3 * mrg::kMaxSegments * 4;
static const uint64_t kWsStatsBytes = MRG_MAX_PASSES * kStatsPerPass * 8;
// round 6: the records of the reads a wave of wave_seed_kernel answers behind its stream (dict.hip; 32 bytes each, kWalkCap per
// wave of the largest grid) -- in the caller's workspace, not in the context: cascades of one context on several streams
// (the e2e leg's chunks) must not share them
static const uint64_t kWsWalkBytes = (uint64_t)mrg::kMaxSegments * (mrg::kSeedThreads / 64u) * kWalkCap * 32ull;

int mrg_cascade_workspace_bytes(uint64_t n, uint64_t* bytes) {
  if (!bytes) return fail(MRG_ERR_ARG, "mrg_cascade_workspace_bytes: null argument");
  *bytes = 3 * ws_idx_bytes(n) + kWsCountsBytes + kWsStatsBytes + kWsWalkBytes;
  return MRG_OK;
}

namespace {
// The index a seed unit searches (SeedLib): cached by member list.
int get_seed_lib(mrg_ctx* ctx, const std::vector<int32_t>& lib_ids, SeedLib** out) {
  std::string key;
  for (int32_t id : lib_ids) key += std::to_string(id) + ",";
  for (auto& sl : ctx->seed_libs)
    if (sl->key == key) {
      *out = sl.get();
      return MRG_OK;
    }
  auto sl = std::make_unique<SeedLib>();
  sl->key = key;
  const DevLib& first = ctx->libs[lib_ids[0]];
  if (lib_ids.size() == 1 && first.host_seqs.empty()) {
    // a large library: its own arrays (the caller made sure it has 16-byte rows)
    sl->lib = first;
    sl->lib.host_seqs.clear();
    sl->lib.kbits_host.clear();
    sl->owned = false;
    sl->entry_lo.assign(1, 0u);
  } else {
    std::vector<std::string> names, seqs;
    for (int32_t id : lib_ids) {
      const DevLib& l = ctx->libs[id];
      sl->entry_lo.push_back((uint32_t)seqs.size());
      for (size_t i = 0; i < l.host_seqs.size(); ++i) {
        names.push_back("u" + std::to_string(id) + "_" + std::to_string(i));
        seqs.push_back(l.host_seqs[i]);
      }
    }
    mrg::FmIndex ix;
    try {
      mrg::build_index(names, seqs, ix);
    } catch (const std::bad_alloc&) {
      return fail(MRG_ERR_NOMEM, "mrg_cascade_run: out of memory indexing libraries %s", key.c_str());
    } catch (const std::exception& e) {
      return fail(MRG_ERR_FORMAT, "mrg_cascade_run: indexing libraries %s: %s", key.c_str(), e.what());
    }
    // build_index leaves jump tables and row context of an index of >= kLazyDeriveBases bases to the
    // device; a context with device_tables = 0 needs them from the host before upload_index reads them
    if (!ix.derived && !(ctx->device_tables && ix.n >= mrg::kLazyDeriveBases)) mrg::derive_tables(ix);
    sl->owned = true;
    struct Guard {
      SeedLib* s;
      ~Guard() {
        if (!s) return;
        free_dev_lib(s->lib);
        (void)hipFree(s->kbits);
      }
    } guard{sl.get()};
    int rc = upload_index(ctx, ix, sl->lib, true, false);
    if (rc) return rc;
    // presence bitmaps of the k-mers, k = 8..11: bit c = some text position starts the k-mer with
    // code c (first base in the low two bits)
    std::vector<uint32_t> bits(mrg::kSeedKbitsWords, 0u);
    for (size_t sg = 0; sg + 1 < ix.seg_start.size(); ++sg) {
      const uint32_t s0 = ix.seg_start[sg], s1 = ix.seg_start[sg + 1];
      for (uint32_t p = s0; p < s1; ++p) {
        const uint32_t w = p >> 4, sh = (p & 15u) * 2u;
        const uint64_t lo64 = (uint64_t)ix.text[w] | ((uint64_t)ix.text[w + 1] << 32);
        const uint32_t win = (uint32_t)(lo64 >> sh);  // 16 bases from p
        for (uint32_t k = 8; k <= 11 && p + k <= s1; ++k) {
          const uint32_t c = win & ((1u << (2 * k)) - 1u);
          bits[mrg::seed_kbits_word_off(k) + (c >> 5)] |= 1u << (c & 31u);
        }
      }
    }
    if ((rc = upload(&sl->kbits, bits))) return rc;
    guard.s = nullptr;
  }
  *out = sl.get();
  ctx->seed_libs.push_back(std::move(sl));
  return MRG_OK;
}
}  // namespace

}  // extern "C"

namespace {
// mrg_cascade_run / mrg_cascade_run_packed: d_packed != null = the one-array output form
int cascade_run_impl(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read,
                    const uint8_t* d_lens, const uint64_t* d_nmask, uint64_t n,
                    const mrg_pass_cfg* passes, uint32_t n_pass, int8_t* d_pass_id,
                    int32_t* d_ref_id, int32_t* d_pos, uint8_t* d_mm, uint32_t* d_packed, uint64_t* d_pass_counts,
                    void* d_workspace, uint64_t workspace_bytes, void* stream_) {
  if (!ctx || !passes || !d_workspace) return fail(MRG_ERR_ARG, "mrg_cascade_run: null argument");
  if (n && (!d_reads || !d_lens || (!d_packed && (!d_pass_id || !d_ref_id || !d_pos || !d_mm))))
    return fail(MRG_ERR_ARG, "mrg_cascade_run: null read/output buffers");
  if (d_packed && n_pass > 15) return fail(MRG_ERR_ARG, "mrg_cascade_run_packed: at most 15 passes fit the packed word");
  if (n_pass == 0 || n_pass > MRG_MAX_PASSES)
    return fail(MRG_ERR_ARG, "mrg_cascade_run: n_pass %u not in [1,%d]", n_pass, MRG_MAX_PASSES);
  if (words_per_read != 1 && words_per_read != 2 && words_per_read != 4 && words_per_read != 8)
    return fail(MRG_ERR_ARG, "mrg_cascade_run: words_per_read must be 1, 2, 4 or 8 (got %u)", words_per_read);
  if (n >= 0xfffffff0ull) return fail(MRG_ERR_ARG, "mrg_cascade_run: at most 2^32-16 reads per call");
  uint64_t need = 0;
  mrg_cascade_workspace_bytes(n, &need);
  if (workspace_bytes < need)
    return fail(MRG_ERR_ARG, "mrg_cascade_run: workspace %llu < %llu bytes",
                (unsigned long long)workspace_bytes, (unsigned long long)need);
  for (uint32_t i = 0; i < n_pass; ++i) {
    const mrg_pass_cfg& c = passes[i];
    if (c.lib < 0 || (size_t)c.lib >= ctx->libs.size())
      return fail(MRG_ERR_ARG, "mrg_cascade_run: pass %u names unknown library %d", i, c.lib);
    if (c.max_mm_seed < 0 || c.max_mm_seed > 3 || c.max_mm_total < c.max_mm_seed || c.trim5 < 0 ||
        c.trim5 > 31 || c.trim3 < 0 || c.seed_len < 1)
      return fail(MRG_ERR_ARG, "mrg_cascade_run: pass %u has an invalid policy", i);
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  if (!ctx->ev_ready) {
    for (auto& e : ctx->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& e : ctx->ev0) HIP_TRY(hipEventCreate(&e));
    ctx->ev_ready = true;
  }

  char* ws = (char*)d_workspace;
  uint32_t* idx[3] = {(uint32_t*)ws, (uint32_t*)(ws + ws_idx_bytes(n)), (uint32_t*)(ws + 2 * ws_idx_bytes(n))};
  uint32_t* counts = (uint32_t*)(ws + 3 * ws_idx_bytes(n));
  uint64_t* stats = (uint64_t*)(ws + 3 * ws_idx_bytes(n) + kWsCountsBytes);
  uint4* const ws_walk = (uint4*)(ws + 3 * ws_idx_bytes(n) + kWsCountsBytes + kWsStatsBytes);
  // per-pass counters only; the outputs need no memset (the last pass writes the
  // "unannotated" values for whatever it does not claim)
  HIP_TRY(hipMemsetAsync(stats, 0, kWsStatsBytes, stream));
  uint32_t prev_grid = 0, prev_seg_cap = 0;
  // "grid_pct": every launch with this share of its workgroups (the kernels loop over their chunks, so fewer
  // workgroups only means more trips each): leaves room on the CUs for the launches of ANOTHER cascade
  // running on another stream at the same time
  auto scale_grid = [&](uint32_t grid) -> uint32_t {
    return std::max<uint32_t>(1u, (uint32_t)((uint64_t)grid * (uint64_t)ctx->grid_pct / 100u));
  };
  // Capacity of a producer workgroup's list segment: it must hold every read the workgroup may be
  // offered -- its share of an identity list in chunks of `chunk` reads, or, reading a list, the
  // chunks it walks (chunk c belongs to segment c % in_nseg and goes to workgroup c % grid: of
  // in_nseg x depth chunks every grid-th one).
  // (0 = the segments would not fit the workspace: every launch adds at most grid x chunk entries of
  // rounding, kListSlack covers sixteen launches)
  auto segment_capacity = [&](uint32_t grid, uint64_t chunk, bool reads_list) -> uint32_t {
    uint64_t cap;
    if (!reads_list) {
      const uint64_t per_trip = chunk * grid;
      cap = ((n + per_trip - 1) / per_trip) * chunk;
    } else {
      const uint64_t depth = ((uint64_t)prev_seg_cap + chunk - 1) / chunk;  // chunks of the longest input segment
      cap = (((uint64_t)prev_grid * depth + grid - 1) / grid) * chunk;
    }
    if (cap * grid > n + mrg::kListSlack) return 0u;
    return (uint32_t)cap;
  };
  int cur_list = 0;        // which list buffer holds the newest survivor list
  bool have_list = false;  // false: the next pass that runs reads the identity list of all reads
  bool list_fat = false;   // the newest list carries its reads (16-byte entries, written by a seed launch or pair_wave_kernel)
  bool out_init = false;   // the first launch wrote every output (exact_dict_kernel, streaming): later ones write claims only
  int pair0 = 0, pair1 = 1;  // the two buffers the running cascade alternates between
  auto other_list = [&](int cur) { return cur == pair0 ? pair1 : pair0; };

  // exact_dict_kernel and the FM kernels read index lists: a list that carries its reads is thinned into the other buffer
  // first (the spike-in pass and non-default plans only)
  auto want_thin_list = [&]() -> int {
    if (!have_list || !list_fat) return MRG_OK;
    const int other = other_list(cur_list);
    HIP_TRY(mrg::launch_list_thin(reinterpret_cast<const uint4*>(idx[cur_list]), idx[other], counts + cur_list * mrg::kMaxSegments,
                                  counts + other * mrg::kMaxSegments, prev_grid, prev_seg_cap, stream));
    cur_list = other;
    list_fat = false;
    return MRG_OK;
  };

  hipEvent_t* evs = ctx->ev;  // the per-pass events of the cascade that is being issued
  uint8_t* ev_ix = ctx->ev_ix;
  uint32_t ev_last = 0;
  const bool timed = !ctx->fused_step;  // ("fused_step": no event lives inside the step: nothing between its launches)
  ctx->last_timed = timed;
  if (timed) HIP_TRY(hipEventRecord(evs[0], stream));
  ev_ix[0] = 0;
  // pass i is issued: `fresh` = something was launched since the boundary before
  auto mark = [&](uint32_t i, bool fresh) -> hipError_t {
    if (!fresh || !timed) {
      ev_ix[i + 1] = (uint8_t)ev_last;
      return hipSuccess;
    }
    ev_ix[i + 1] = (uint8_t)(i + 1);
    ev_last = i + 1;
    return hipEventRecord(evs[i + 1], stream);
  };

  // ---- launch plan: which passes run, and which consecutive ones share a (fused) launch ----
  // A pass whose length window excludes every read of the batch (caller's hint: e.g. the hairpin
  // pass, len > 25, on 22-nt reads) would only copy its input list: it is not launched; its
  // counters stay zero and the next pass reads the same list.  (Never the last pass: that one
  // writes the "unannotated" values.)
  // A 2-mismatch `--best` pass on a library with an exact-match dictionary, right behind a seed
  // launch: its exact stratum -- what most isomiRs are after the -5 / -3 trims -- rides in that launch
  // as a dictionary unit (a 0-mismatch hit is final: nothing beats it, the lowest (entry, offset) wins
  // ties as always); the pass's own launch then searches the reads that are left and does not count
  // "processed" again.
  bool stratum0_done[MRG_MAX_PASSES] = {false};
  bool runs[MRG_MAX_PASSES];
  for (uint32_t i = 0; i < n_pass; ++i) {
    const mrg_pass_cfg& c = passes[i];
    runs[i] = !(i + 1 < n_pass && (c.min_len > ctx->hint_max_len || c.max_len < ctx->hint_min_len));
    ctx->last_lds[i] = 0;
    ctx->last_mode[i] = 0;
    ctx->last_group[i] = i;
    ctx->last_launches[i] = 0;
    ctx->last_kbits_log2[i] = 0;
    ctx->last_pair_anchor[i] = 0;
    ctx->last_variant[i] = 0;
  }
  auto fusable = [&](uint32_t i) {
    // the first launched pass streams the whole read set and keeps the classic kernel (library
    // text + full bitmap in LDS); 2-mismatch policies have their own stratum-first instantiation.
    // (A fused launch counts offered / aligned reads in 16-bit per-lane fields: one workgroup per
    // CU must see fewer than 65536 chunks.)
    return ctx->fuse != 0 && passes[i].max_mm_seed <= 1 && n / (1024ull * (uint64_t)std::max(ctx->n_cu, 1)) < 60000ull;
  };
  auto small_lib = [&](uint32_t i) { return ctx->libs[passes[i].lib].kbits != nullptr && ctx->kmer_filter; };
  // one-word reads without N: the batches the dictionary kernels (dict.hip) take.  A batch with
  // longer reads, reads with N or very short reads is SPLIT: the reads of split_min_len .. 32 nt
  // without N -- most of any small-RNA read set -- go through the cascade as the one-word batch they
  // are (the kernels read word 0 of the plane-major array), the rest through the FM kernels (whose
  // backtracking search does not mind a 16-nt read's 8-base seeds); two cascades over disjoint lists,
  // one after the other, adding to the same counters.  (The length hint only says whether a one-word
  // batch may hold such short reads: a wrong hint costs time, never a result.)
  bool dict_batch = ctx->dict && words_per_read == 1 && !d_nmask;
  uint32_t words_eff = words_per_read;
  const uint64_t* nmask_eff = d_nmask;
  const bool split = ctx->dict && ctx->split_mixed && n > 0 && ctx->force_lds_mode < 0 &&
                     (!dict_batch || ctx->hint_min_len < ctx->split_min_len) &&
                     (ctx->hint_min_len <= 32 || (ctx->long_lane && words_per_read >= 2 && ctx->hint_min_len <= 63)) &&
                     ctx->hint_max_len >= ctx->split_min_len;  // (some read may be on either side)

  // Round 6: which read lengths beyond 32 nt the one-word lane of a split batch takes (bit b = length 33 + b, up to 63): the
  // seed kernels have instantiations that carry a read's second word (seeds from the first 32 bases, the rest compared
  // where an alignment is verified; dictionary units answer such reads by FM search); exact_dict_kernel and the FM kernels
  // of the lane do not, pair_wave_kernel takes a read whose TRIMMED length fits one word.  A length is let in when every
  // pass of the cascade either runs in a seed launch, or cannot hold a read of that length in its window, or
  // (pair_wave_kernel) sees at most 32 bases of it or could not align it at all (longer than the library's longest entry).
  // Decided below, once the lambdas that route a pass exist; 0 = the lane is for reads of at most 32 nt, as before.
  uint64_t long_mask = 0;
  auto window_bits = [](const mrg_pass_cfg& c) -> uint64_t {
    uint64_t m = 0;
    for (int L = 33; L <= 63; ++L)
      if (L >= c.min_len && L <= c.max_len) m |= 1ull << (L - 33);
    return m;
  };
  // the classic path: one match_kernel launch for pass i
  auto run_single = [&](uint32_t i, int32_t k_first, int32_t k_last, bool first_part, bool last_part, bool by_pairs = false) -> int {
    const mrg_pass_cfg& c = passes[i];
    const DevLib& l = ctx->libs[c.lib];
    // (pair_wave_kernel reads either list form; everything else launched here reads index lists)
    const bool pair_wave_route = by_pairs && dict_batch && ctx->pair_impl != 0 && c.max_mm_seed == 2 && ctx->force_lds_mode < 0;
    if (!pair_wave_route) {
      const int rc_form = want_thin_list();
      if (rc_form != MRG_OK) return rc_form;
    }
    // (the dictionary's key is matched letter for letter: only a pass whose seed covers it may take it)
    if (dict_batch && c.max_mm_seed == 0 && l.dict_slots && c.seed_len >= (int32_t)l.dict_key && ctx->force_lds_mode < 0) {
      // no seed mismatch, one-word reads, a library with an exact-match dictionary: dict.hip
      mrg::ExactParams e;
      e.slots = reinterpret_cast<const uint4*>(l.dict_slots);
      e.log2_slots = l.dict_log2;
      e.key_bases = l.dict_key;
      e.kbits = (ctx->kmer_filter && c.seed_len >= (int32_t)mrg::kKmerBitsK) ? l.kbits : nullptr;
      e.blocks = l.blocks;
      e.super = l.super;
      e.primary = l.primary;
      e.ftab = l.ftab;
      e.tabs = l.tabs;
      e.sa = l.sa;
      e.text = l.text;
      e.n = l.n;
      e.seg_start = l.seg_start;
      e.seg_ref = l.seg_ref;
      e.seg_off = l.seg_off;
      e.chunk_seg = l.chunk_seg;
      e.simple_segs = l.simple ? 1u : 0u;
      e.reads = d_reads;
      e.lens = d_lens;
      e.n_total = (uint32_t)n;
      const int next_list = have_list ? other_list(cur_list) : pair0;
      e.idx_in = have_list ? idx[cur_list] : nullptr;
      e.in_count = counts + cur_list * mrg::kMaxSegments;
      e.in_nseg = prev_grid;
      e.in_seg_cap = prev_seg_cap;
      e.idx_out = (i + 1 < n_pass) ? idx[next_list] : nullptr;
      e.out_count = counts + next_list * mrg::kMaxSegments;
      e.pass_id = d_pass_id;
      e.ref_id = d_ref_id;
      e.pos = d_pos;
      e.mm = d_mm;
      e.packed = d_packed;
      e.counters = stats + (size_t)i * kStatsPerPass;
      e.seed_len = c.seed_len;
      e.max_mm_total = c.max_mm_total;
      e.trim5 = c.trim5;
      e.trim3 = c.trim3;
      e.min_len = c.min_len;
      e.max_len = c.max_len;
      e.poly_t = c.poly_t;
      e.pass_index = (int32_t)i;
      uint32_t grid = (uint32_t)ctx->n_cu * 2u;
      if (grid > 512u) grid = 512u;
      grid = scale_grid(grid);
      // (a workgroup takes chunks of 4096 reads: four per lane)
      const uint32_t seg_cap = segment_capacity(grid, mrg::kExactChunk, have_list);
      if (!seg_cap && n) return fail(MRG_ERR_ARG, "mrg_cascade_run: survivor lists outgrew the workspace");
      e.out_seg_cap = seg_cap;
      ctx->last_lds[i] = 0u;
      ctx->last_mode[i] = 7u;
      ctx->last_group[i] = i;
      if (long_mask & window_bits(c))
        return fail(MRG_ERR_ARG, "mrg_cascade_run: internal: reads of more than 32 nt reached exact_dict_kernel (pass %u)", i);
      if (n) HIP_TRY(mrg::launch_exact_dict(e, grid, stream));
      ctx->last_variant[i] = (n && mrg::exact_dict_stretches(e, grid)) ? 16u : 0u;
      ctx->last_launches[i] += 1;
      HIP_TRY(mark(i, true));
      if (n && mrg::exact_dict_streams(e)) out_init = true;  // (every output of the batch is written: later launches write claims only)
      if (e.idx_out) {
        cur_list = next_list;
        have_list = true;
        list_fat = false;
        prev_grid = grid;
        prev_seg_cap = seg_cap;
      }
      return MRG_OK;
    }
    mrg::MatchParams p;
    p.blocks = l.blocks;
    p.super = l.super;
    p.text = l.text;
    p.sa = l.sa;
    p.ctx = l.ctx;
    p.ftab = l.ftab;
    p.tabs = l.tabs;
    if (!ctx->use_ftab) p.tabs.k[0] = 0u;
    p.seg_start = l.seg_start;
    p.seg_ref = l.seg_ref;
    p.seg_off = l.seg_off;
    p.chunk_seg = l.chunk_seg;
    p.n = l.n;
    p.nblk = l.nblk;
    p.nsup = l.nsup;
    p.primary = l.primary;
    p.text_words = l.text_words;
    p.simple_segs = l.simple ? 1u : 0u;
    p.reads = d_reads;
    p.reads_hi = nullptr;
    p.lens = d_lens;
    p.nmask = nmask_eff;
    p.n_total = (uint32_t)n;
    const int next_list = have_list ? other_list(cur_list) : pair0;
    p.idx_in = have_list ? idx[cur_list] : nullptr;
    p.in_count = counts + cur_list * mrg::kMaxSegments;
    p.in_nseg = prev_grid;
    p.in_seg_cap = prev_seg_cap;
    p.idx_out = (i + 1 < n_pass || !last_part) ? idx[next_list] : nullptr;
    p.in_stride = list_fat ? 4u : 1u;
    p.out_init = out_init ? 1u : 0u;
    p.k_first = k_first;
    p.k_last = k_last;
    p.count_processed = first_part ? 1u : 0u;
    // (the length hints only decide which passes are launched: every kernel reads the length array --
    // a caller whose equal hints do not describe the batch gets a skipped pass at worst, never a read
    // aligned with somebody else's length)
    p.uniform_len = 0u;
    p.out_count = counts + next_list * mrg::kMaxSegments;
    p.pass_id = d_pass_id;
    p.ref_id = d_ref_id;
    p.pos = d_pos;
    p.mm = d_mm;
    p.packed = d_packed;
    p.counters = stats + (size_t)i * kStatsPerPass;
    p.seed_len = c.seed_len;
    p.max_mm_seed = c.max_mm_seed;
    p.max_mm_total = c.max_mm_total;
    p.trim5 = c.trim5;
    p.trim3 = c.trim3;
    p.min_len = c.min_len;
    p.max_len = c.max_len;
    p.poly_t = c.poly_t;
    p.pass_index = (int32_t)i;
    p.wstop = (uint32_t)ctx->wstop;
    // a large library's wide intervals are pre-filtered by row context in the cooperative path:
    // send them there early (lane-serial verification of 100+ rows is what hurts short reads)
    p.wide_rows = l.ctx ? (uint32_t)std::min<int64_t>(ctx->wide_rows, ctx->ctx_wide_rows) : (uint32_t)ctx->wide_rows;

    // residency decision.  The superblock table (16 B per 65536 bp) and the segment
    // prefix always sit in LDS.  With the jump table most seed searches need few LF
    // steps, so the packed text (read by every verification) is the first thing worth
    // staging and two resident workgroups per CU (<= 80 KB each) beat a fuller LDS:
    //   2 = blocks + text, 3 = text only, 1 = blocks only, 0 = nothing.
    const uint64_t blk_bytes = (uint64_t)l.nblk * 16, txt_bytes = (uint64_t)l.text_words * 4;
    // a small library's 9-mer bitmap (32 KB; only built for libraries whose text leaves room for
    // it next to a second workgroup) rides in LDS ("kmer_filter" = 0 switches it off)
    const uint64_t kb_bytes = (l.kbits && ctx->kmer_filter) ? (uint64_t)mrg::kKmerBitsWords * 4 : 0;
    const bool use_kbits = kb_bytes != 0;
    p.kbits = use_kbits ? l.kbits : nullptr;
    const uint64_t overhead = (uint64_t)l.nsup * 16 + mrg::kMatchCtlBytes + (use_kbits ? kb_bytes : 0);
    const uint64_t budget = (uint64_t)ctx->lds_budget, hard = 160 * 1024, half = 80 * 1024;
    if (overhead > hard)
      return fail(MRG_ERR_ARG, "mrg_cascade_run: the superblock table of library %d does not fit LDS", c.lib);
    int lds_mode = 0;
    uint64_t lib_bytes = 0;
    if (blk_bytes + txt_bytes <= budget && overhead + blk_bytes + txt_bytes <= half) {
      lds_mode = 2;
      lib_bytes = blk_bytes + txt_bytes;
    } else if (ctx->prefer_two_blocks && txt_bytes <= budget && overhead + txt_bytes <= half) {
      lds_mode = 3;
      lib_bytes = txt_bytes;
    } else if (blk_bytes + txt_bytes <= budget && overhead + blk_bytes + txt_bytes <= hard) {
      lds_mode = 2;
      lib_bytes = blk_bytes + txt_bytes;
    } else if (txt_bytes <= budget && overhead + txt_bytes <= hard) {
      lds_mode = 3;
      lib_bytes = txt_bytes;
    } else if (blk_bytes <= budget && overhead + blk_bytes <= hard) {
      lds_mode = 1;
      lib_bytes = blk_bytes;
    }
    if (ctx->force_lds_mode >= 0) {  // test hook: exercise a specific kernel variant
      const int m = (int)ctx->force_lds_mode;
      const uint64_t need = (m == 2 ? blk_bytes + txt_bytes : m == 3 ? txt_bytes : m == 1 ? blk_bytes : 0);
      if (overhead + need <= hard) {
        lds_mode = m;
        lib_bytes = need;
      }
    }
    // a strata launch of a 2-mismatch pass whose rows are compacted over the wave (stratum_kernel):
    // text in LDS when it fits, occ blocks always from L2
    const bool rows_kernel = c.max_mm_seed == 2 && ctx->force_lds_mode < 0 &&
                             (by_pairs || ctx->stratum_rows == 2 || (ctx->stratum_rows == 1 && k_first == k_last));
    p.pair_anchor = by_pairs ? l.pair_anchor : 0u;
    p.pair_jump = l.pair_jump;
    p.pair_rows = l.pair_rows;
    p.pair_jump_s = l.pair_jump_s;
    for (int t = 0; t < 3; ++t) {
      p.pair_row_off[t] = l.pair_row_off[t];
      p.pair_row_off_s[t] = l.pair_row_off_s[t];
    }
    bool rows_lds_text = false;
    if (rows_kernel) {
      const uint64_t ov = (uint64_t)l.nsup * 16 + mrg::kStratumCtlBytes + (use_kbits ? kb_bytes : 0);
      if (ov > hard)
        return fail(MRG_ERR_ARG, "mrg_cascade_run: the superblock table of library %d does not fit LDS", c.lib);
      rows_lds_text = txt_bytes <= budget && ov + txt_bytes <= hard;
      lds_mode = rows_lds_text ? 3 : 0;
      lib_bytes = rows_lds_text ? txt_bytes : 0;
    }
    // one-word reads without N through the anchor pairs: pair_wave_kernel (dict.hip; every wave on its own, items
    // and rows compacted over the wave; nothing staged in LDS but the wave's 256 reads)
    const bool pair_wave = pair_wave_route && rows_kernel;
    if (pair_wave) {
      lds_mode = 0;
      lib_bytes = 0;
    }
    const uint32_t lds_total = pair_wave ? mrg::pair_wave_lds_total()
                               : rows_kernel ? (uint32_t)((uint64_t)l.nsup * 16 + mrg::kStratumCtlBytes + (use_kbits ? kb_bytes : 0) + lib_bytes)
                                             : (uint32_t)(overhead + lib_bytes);
    const uint32_t lds_bytes = (uint32_t)lib_bytes;
    const uint32_t per_cu = pair_wave ? std::min<uint32_t>(5u, (160u * 1024u) / lds_total) : ((lds_total * 2u <= 160u * 1024u) ? 2u : 1u);
    uint32_t grid = (uint32_t)ctx->n_cu * per_cu;
    if (grid > mrg::kMaxSegments) grid = mrg::kMaxSegments;
    grid = scale_grid(grid);
    // a workgroup's segment must hold every read it may be offered
    const uint32_t seg_cap = segment_capacity(grid, 1024, have_list);
    if (!seg_cap && n) return fail(MRG_ERR_ARG, "mrg_cascade_run: survivor lists outgrew the workspace");
    p.out_seg_cap = seg_cap;
    ctx->last_lds[i] = lds_bytes;
    ctx->last_mode[i] = (uint32_t)lds_mode;
    ctx->last_group[i] = i;
    ctx->last_kbits_log2[i] = use_kbits ? 18u : 0u;
    ctx->last_pair_anchor[i] = p.pair_anchor;
    if (rows_kernel) ctx->last_mode[i] = pair_wave ? 11u : (rows_lds_text ? 5u : 6u);
    if (long_mask && !pair_wave && (long_mask & window_bits(c)))
      return fail(MRG_ERR_ARG, "mrg_cascade_run: internal: reads of more than 32 nt reached a one-word kernel (pass %u)", i);
    if (n && pair_wave) {
      p.reads_hi = long_mask ? d_reads + n : nullptr;
      ctx->last_variant[i] = long_mask ? 8u : 0u;
      HIP_TRY(mrg::launch_pair_wave(p, grid, stream));
    } else if (n && rows_kernel) {
      HIP_TRY(mrg::launch_stratum(p, words_eff, rows_lds_text, grid, lds_total, stream));
    } else if (n) {
      HIP_TRY(mrg::launch_match(p, words_eff, lds_mode, grid, lds_total, stream));
    }
    ctx->last_launches[i] += 1;
    if (last_part) HIP_TRY(mark(i, true));
    if (p.idx_out) {
      cur_list = next_list;
      have_list = true;
      list_fat = pair_wave;
      prev_grid = grid;
      prev_seg_cap = seg_cap;
    }
    return MRG_OK;
  };

  // pair tables of a large library (three anchors of A bases, gaps A and 2 A: pairs.hip), built on the device the first
  // time a cascade meets reads whose seed region is 3 A .. 4 A - 1 bases under a one-mismatch policy
  auto ensure_bpair = [&](DevLib& lm, int64_t A) -> int {
    if (lm.bpair_anchor == (uint32_t)A || lm.bpair_failed) return MRG_OK;
    (void)hipFree(lm.bpair_jump);
    (void)hipFree(lm.bpair_rows);
    lm.bpair_jump = nullptr;
    lm.bpair_rows = nullptr;
    lm.bpair_anchor = 0;
    const uint64_t n_rows = (uint64_t)lm.n + 1, n_codes1 = (1ull << (4u * (uint32_t)A)) + 1ull;
    const uint64_t need = 2 * n_codes1 * 4 + 2 * n_rows * 8 + 4 * n_rows * 4 + (256ull << 20);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (free_b < need || hipMalloc((void**)&lm.bpair_jump, 2 * n_codes1 * 4) != hipSuccess ||
        hipMalloc((void**)&lm.bpair_rows, 2 * n_rows * 8) != hipSuccess ||
        mrg::build_pair_tables_device(lm.sa, lm.text, (uint32_t)n_rows, (uint32_t)A, 2, lm.bpair_jump, lm.bpair_rows, lm.bpair_row_off,
                                      stream) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(lm.bpair_jump);
      (void)hipFree(lm.bpair_rows);
      lm.bpair_jump = nullptr;
      lm.bpair_rows = nullptr;
      lm.bpair_failed = true;
    } else {
      lm.bpair_anchor = (uint32_t)A;
    }
    return MRG_OK;
  };

  // one fused_kernel launch for the running passes among [first, last]
  auto run_fused = [&](const uint32_t* members, uint32_t n_sub, bool ends_cascade) -> int {
    for (uint32_t q = 0; q < n_sub; ++q)
      if (long_mask & window_bits(passes[members[q]]))
        return fail(MRG_ERR_ARG, "mrg_cascade_run: internal: reads of more than 32 nt reached fused_kernel<1> (pass %u)", members[q]);
    mrg::FusedParams fp;
    std::memset(&fp, 0, sizeof fp);
    fp.n_sub = n_sub;
    // which sub-passes filter seed pieces with their library's 9-mer bitmap
    uint32_t lg[mrg::kMaxFused];
    for (uint32_t q = 0; q < n_sub; ++q) lg[q] = small_lib(members[q]) ? 18u : 0u;
    // one 1024-thread workgroup per CU (128 VGPRs per lane), so the whole LDS is its own
    const uint64_t fixed = mrg::fused_fixed_lds_bytes();
    const uint64_t lds_cap = 160 * 1024;
    const uint64_t budget = std::min<uint64_t>((uint64_t)std::max<int64_t>(ctx->lds_budget, (int64_t)fixed), lds_cap) - fixed;
    // rounds: the sub-passes of a run of bitmap-filtered libraries are looked up together (few
    // items per read), with "round_large" = 1 so are those of a run of large ones (measured equal,
    // default 0: every unclaimed read has items in each of them)
    fp.n_rounds = 0;
    for (uint32_t q = 0; q < n_sub;) {
      uint32_t e = q + 1;
      if (lg[q] || ctx->round_large)
        while (e < n_sub && (lg[e] != 0) == (lg[q] != 0) && e - q < mrg::kMaxRoundSubs) ++e;
      fp.round_first[fp.n_rounds] = (uint8_t)q;
      fp.round_count[fp.n_rounds] = (uint8_t)(e - q);
      ++fp.n_rounds;
      q = e;
    }
    // the 9-mer table of each bitmap round: one entry per code with a bit per sub-pass, as many
    // codes (2^18 = every 9-mer, else folded: entry h = OR of the codes = h mod 2^log2) as the LDS
    // left by the other rounds' tables allows
    uint32_t n_tab_rounds = 0;
    for (uint32_t r = 0; r < fp.n_rounds; ++r) n_tab_rounds += lg[fp.round_first[r]] ? 1u : 0u;
    uint32_t kb_words = 0;
    for (uint32_t q = 0; q < n_sub; ++q) {
      const uint32_t i = members[q];
      const mrg_pass_cfg& c = passes[i];
      const DevLib& l = ctx->libs[c.lib];
      mrg::SubPass& sp = fp.sub[q];
      sp.blocks = l.blocks;
      sp.super = l.super;
      sp.text = l.text;
      sp.sa = l.sa;
      sp.ctx = l.ctx;
      sp.sa16 = reinterpret_cast<const uint4*>(l.sa16);
      sp.ftab = l.ftab;
      sp.tabs = l.tabs;
      if (!ctx->use_ftab) sp.tabs.k[0] = 0u;
      sp.seg_start = l.seg_start;
      sp.seg_ref = l.seg_ref;
      sp.seg_off = l.seg_off;
      sp.chunk_seg = l.chunk_seg;
      sp.nmask = nmask_eff;
      sp.counters = stats + (size_t)i * kStatsPerPass;
      sp.n = l.n;
      sp.primary = l.primary;
      sp.simple_segs = l.simple ? 1u : 0u;
      sp.kb_bit = 0xFFu;
      sp.seed_len = c.seed_len;
      sp.max_mm_seed = c.max_mm_seed;
      sp.max_mm_total = c.max_mm_total;
      sp.trim5 = c.trim5;
      sp.trim3 = c.trim3;
      sp.min_len = c.min_len;
      sp.max_len = c.max_len;
      sp.poly_t = c.poly_t;
      sp.pass_index = (int32_t)i;
      sp.pair_anchor = 0u;
      {
        // reads whose seed region is 3A .. 4A - 1 bases: three anchor pairs instead of two pieces of
        // less than 2A bases (tables built on the device at first use, pairs.hip)
        const int64_t A = ctx->pair_big;
        const int64_t r_min = std::min<int64_t>(std::max<int64_t>(ctx->hint_min_len, c.min_len) - c.trim5 - c.trim3, c.seed_len);
        const int64_t r_max = std::min<int64_t>(std::min<int64_t>(ctx->hint_max_len, c.max_len) - c.trim5 - c.trim3, c.seed_len);
        if (A && c.max_mm_seed == 1 && !c.poly_t && l.n >= mrg::kWideRowMinBases && !ctx->round_large && r_min < 4 * A &&
            r_max >= 3 * A) {
          DevLib& lm = ctx->libs[c.lib];
          {
            const int rc_bp = ensure_bpair(lm, A);
            if (rc_bp != MRG_OK) return rc_bp;
          }
          if (lm.bpair_anchor == (uint32_t)A) {
            sp.pair_jump = lm.bpair_jump;
            sp.pair_rows = lm.bpair_rows;
            sp.pair_row_off[0] = lm.bpair_row_off[0];
            sp.pair_row_off[1] = lm.bpair_row_off[1];
            sp.pair_anchor = (uint32_t)A;
          }
        }
      }
      ctx->last_pair_anchor[i] = sp.pair_anchor;
      ctx->last_lds[i] = 0u;
      ctx->last_mode[i] = 4u;
      ctx->last_group[i] = members[0];
      ctx->last_kbits_log2[i] = 0u;
    }
    for (uint32_t r = 0; r < fp.n_rounds; ++r) {
      const uint32_t q0 = fp.round_first[r], nq = fp.round_count[r];
      fp.round_kb_log2[r] = 0;
      fp.round_kb_bits[r] = 0;
      fp.round_kb_off[r] = 0;
      fp.round_kb_src[r] = nullptr;
      if (!lg[q0]) continue;
      const uint32_t bits = nq <= 4 ? 4u : 8u;
      uint32_t log2c = 18;
      while (log2c > 13 && ((uint64_t)(1u << log2c) * bits / 8) * n_tab_rounds > budget) --log2c;
      if (((uint64_t)(1u << log2c) * bits / 8) * n_tab_rounds > budget) {  // no room: unfiltered
        for (uint32_t q = q0; q < q0 + nq; ++q) ctx->last_kbits_log2[members[q]] = 255u;
        continue;
      }
      std::string key;
      for (uint32_t q = q0; q < q0 + nq; ++q) key += std::to_string(passes[members[q]].lib) + ",";
      key += "|" + std::to_string(log2c) + "|" + std::to_string(bits);
      uint32_t* d_tab = nullptr;
      for (auto& kv : ctx->round_tables)
        if (kv.first == key) d_tab = kv.second;
      if (!d_tab) {
        std::vector<uint32_t> tab(((size_t)1 << log2c) * bits / 32, 0u);
        for (uint32_t q = q0; q < q0 + nq; ++q) {
          const std::vector<uint32_t>& kbh = ctx->libs[passes[members[q]].lib].kbits_host;
          for (uint32_t c = 0; c < (1u << 18); ++c) {
            if (!((kbh[c >> 5] >> (c & 31u)) & 1u)) continue;
            const uint64_t bit = (uint64_t)(c & ((1u << log2c) - 1u)) * bits + (q - q0);
            tab[bit >> 5] |= 1u << (bit & 31u);
          }
        }
        HIP_TRY(hipMalloc((void**)&d_tab, tab.size() * 4));
        HIP_TRY(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        ctx->round_tables.emplace_back(key, d_tab);
      }
      fp.round_kb_src[r] = d_tab;
      fp.round_kb_off[r] = kb_words;
      fp.round_kb_log2[r] = (uint8_t)log2c;
      fp.round_kb_bits[r] = (uint8_t)bits;
      const uint32_t words = (uint32_t)(((size_t)1 << log2c) * bits / 32);
      kb_words += words;
      for (uint32_t q = q0; q < q0 + nq; ++q) {
        fp.sub[q].kb_bit = q - q0;
        ctx->last_lds[members[q]] = words * 4u / nq;
        ctx->last_kbits_log2[members[q]] = log2c;
      }
    }
    fp.kb_words = kb_words;
    fp.reads = d_reads;
    fp.lens = d_lens;
    fp.nmask = nmask_eff;
    fp.n_total = (uint32_t)n;
    fp.uniform_len = 0u;  // (see run_single)
    {
      const int rc_form = want_thin_list();
      if (rc_form != MRG_OK) return rc_form;
    }
    const int next_list = have_list ? other_list(cur_list) : pair0;
    fp.idx_in = have_list ? idx[cur_list] : nullptr;
    fp.in_count = counts + cur_list * mrg::kMaxSegments;
    fp.in_nseg = prev_grid;
    fp.in_seg_cap = prev_seg_cap;
    fp.idx_out = ends_cascade ? nullptr : idx[next_list];
    fp.out_count = counts + next_list * mrg::kMaxSegments;
    fp.pass_id = d_pass_id;
    fp.ref_id = d_ref_id;
    fp.pos = d_pos;
    fp.mm = d_mm;
    fp.packed = d_packed;
    fp.wstop = (uint32_t)ctx->wstop;
    const uint32_t lds_total = kb_words * 4u + (uint32_t)fixed;
    // 128 VGPRs per lane (no spills in the pipelined walk): 16 waves = one workgroup per CU
    const uint32_t per_cu = 1u;
    uint32_t grid = (uint32_t)ctx->n_cu * per_cu;
    if (grid > mrg::kMaxSegments) grid = mrg::kMaxSegments;
    grid = scale_grid(grid);
    const uint32_t seg_cap = segment_capacity(grid, 1024, have_list);
    if (!seg_cap && n) return fail(MRG_ERR_ARG, "mrg_cascade_run: survivor lists outgrew the workspace");
    fp.out_seg_cap = seg_cap;
    if (n) HIP_TRY(mrg::launch_fused(fp, words_eff, grid, lds_total, stream));
    ctx->last_launches[members[0]] = 1;
    for (uint32_t q = 0; q < n_sub; ++q) HIP_TRY(mark(members[q], q == 0));
    if (fp.idx_out) {
      cur_list = next_list;
      have_list = true;
      list_fat = false;
      prev_grid = grid;
      prev_seg_cap = seg_cap;
    }
    return MRG_OK;
  };

  // one seed_kernel launch (dict.hip) for the passes of `plan`: each entry = one unit = (kind, member passes)
  struct UnitPlan {
    uint32_t kind;
    std::vector<uint32_t> members;
    bool stratum0 = false;  // the exact stratum of a LATER 2-mismatch pass (see stratum0_done)
  };
  auto run_seed = [&](const std::vector<UnitPlan>& plan, uint32_t first, uint32_t end, bool ends_cascade, bool small) -> int {
    mrg::SeedParams sp;
    std::memset(&sp, 0, sizeof sp);
    sp.n_units = (uint32_t)plan.size();
    for (uint32_t u = 0; u < sp.n_units; ++u) {
      mrg::SeedUnit& un = sp.unit[u];
      const mrg_pass_cfg& c0 = passes[plan[u].members[0]];
      un.kind = plan[u].kind;
      const DevLib* fm = &ctx->libs[c0.lib];
      SeedLib* sl = nullptr;
      if (un.kind == 0u) {
        std::vector<int32_t> ids;
        for (uint32_t i : plan[u].members) ids.push_back(passes[i].lib);
        int rc = get_seed_lib(ctx, ids, &sl);
        if (rc != MRG_OK) return rc;
        fm = &sl->lib;
        un.kbits = sl->kbits;
      } else {
        un.slots = reinterpret_cast<const uint4*>(fm->dict_slots);
        un.log2_slots = fm->dict_log2;
        un.key_bases = fm->dict_key;
      }
      // one large library under a one-mismatch policy, and (by the length hint) reads whose seed region is 3 A .. 4 A - 1
      // bases: its pair tables (wave_seed_kernel parks those reads for the three anchor pairs)
      if (un.kind == 0u && plan[u].members.size() == 1 && !c0.poly_t && c0.max_mm_seed == 1 && ctx->pair_big && !plan[u].stratum0 &&
          ctx->libs[c0.lib].n >= mrg::kWideRowMinBases) {
        const int64_t A = ctx->pair_big;
        const int64_t r_min = std::min<int64_t>(std::max<int64_t>(ctx->hint_min_len, c0.min_len) - c0.trim5 - c0.trim3, c0.seed_len);
        const int64_t r_max = std::min<int64_t>(std::min<int64_t>(ctx->hint_max_len, c0.max_len) - c0.trim5 - c0.trim3, c0.seed_len);
        if (r_min < 4 * A && r_max >= 3 * A) {
          DevLib& lm = ctx->libs[c0.lib];
          const int rc_bp = ensure_bpair(lm, A);
          if (rc_bp != MRG_OK) return rc_bp;
          if (lm.bpair_anchor == (uint32_t)A) {
            un.bpair_jump = lm.bpair_jump;
            un.bpair_rows = lm.bpair_rows;
            un.bpair_row_off[0] = lm.bpair_row_off[0];
            un.bpair_row_off[1] = lm.bpair_row_off[1];
            un.bpair_anchor = (uint32_t)A;
          }
        }
      }
      un.ftab = fm->ftab;
      un.tabs = fm->tabs;
      un.sa16 = reinterpret_cast<const uint4*>(fm->sa16);
      un.buckets = ctx->seed_buckets ? reinterpret_cast<const uint4*>(fm->buckets) : nullptr;
      un.bucket_k = fm->bucket_k;
      un.pos_rows = (ctx->seed_buckets && ctx->pos_scan) ? reinterpret_cast<const uint4*>(fm->pos_rows) : nullptr;
      un.sa = fm->sa;
      un.text = fm->text;
      un.blocks = fm->blocks;
      un.super = fm->super;
      un.primary = fm->primary;
      un.n = fm->n;
      un.seg_start = fm->seg_start;
      un.seg_ref = fm->seg_ref;
      un.seg_off = fm->seg_off;
      un.chunk_seg = fm->chunk_seg;
      un.simple_segs = fm->simple ? 1u : 0u;
      un.max_mm_seed = plan[u].stratum0 ? 0 : c0.max_mm_seed;
      un.trim5 = c0.trim5;
      un.trim3 = c0.trim3;
      un.min_len = c0.min_len;
      un.max_len = c0.max_len;
      un.poly_t = c0.poly_t;
      un.n_members = (uint32_t)plan[u].members.size();
      un.min_seed_len = 0x7FFFFFFF;
      un.max_total = 0;
      for (uint32_t mi = 0; mi < un.n_members; ++mi) {
        const uint32_t i = plan[u].members[mi];
        const mrg_pass_cfg& c = passes[i];
        un.m[mi].pass_index = (int32_t)i;
        un.m[mi].seed_len = c.seed_len;
        un.m[mi].max_mm_total = c.max_mm_total;
        un.m[mi].entry_lo = sl ? sl->entry_lo[mi] : 0u;
        un.min_seed_len = std::min(un.min_seed_len, c.seed_len);
        un.max_total = std::max(un.max_total, c.max_mm_total);
        if (plan[u].stratum0) continue;  // (that pass has its own launch afterwards and reports that one)
        ctx->last_lds[i] = 0u;
        ctx->last_mode[i] = 8u;
        ctx->last_group[i] = first;
        ctx->last_kbits_log2[i] = un.kbits ? 22u : 0u;
      }
    }
    // which kernel: by default (-1) the tile kernel for the small libraries' launch (its waves are bound by what
    // they do per read; with more registers and fewer waves the wave kernel only draws level, and its parked
    // reads leave the list's order) and the wave kernel, 96 registers, for a launch on large libraries (bound by
    // the L2 misses of its bucket / slot lines: no barrier, nothing parked but overflowing buckets: 2.08 -> 1.8 ms)
    {
      const int64_t impl = ctx->seed_impl >= 0 ? ctx->seed_impl : (small ? 0 : 2);
      sp.impl = impl ? 1u : 0u;
      sp.wave_regs = impl >= 2 ? 1u : 0u;
      for (uint32_t q = first; q < end; ++q)
        ctx->last_variant[q] = (sp.impl ? ((sp.wave_regs & 1u) ? 2u : 1u) : 0u) | ((have_list && list_fat) ? 4u : 0u) | (long_mask ? 8u : 0u);
    }
    sp.reads_per_lane = 1u;
    sp.item_cap = mrg::kSeedThreads * sp.reads_per_lane * 2u;
    sp.row_cap = sp.impl ? 192u : (small ? 1024u : 2048u);
    sp.stats = stats;
    sp.reads = d_reads;
    sp.reads_hi = long_mask ? d_reads + n : nullptr;
    sp.lens = d_lens;
    sp.n_total = (uint32_t)n;
    const int next_list = have_list ? other_list(cur_list) : pair0;
    sp.idx_in = have_list ? idx[cur_list] : nullptr;
    sp.in_stride = list_fat ? 4u : 1u;
    sp.out_init = out_init ? 1u : 0u;
    sp.in_count = counts + cur_list * mrg::kMaxSegments;
    sp.in_nseg = prev_grid;
    sp.in_seg_cap = prev_seg_cap;
    sp.idx_out = ends_cascade ? nullptr : idx[next_list];
    sp.out_count = counts + next_list * mrg::kMaxSegments;
    sp.pass_id = d_pass_id;
    sp.ref_id = d_ref_id;
    sp.pos = d_pos;
    sp.mm = d_mm;
    sp.packed = d_packed;
    {
      // which instantiation the launch gets: mrg_pass_stats.lds_mode 8 = seed_kernel<false, 8>, 9 = <true, 6>
      bool with_buckets = false;
      for (uint32_t u = 0; u < sp.n_units; ++u) with_buckets |= sp.unit[u].kind == 0u && sp.unit[u].buckets != nullptr;
      if (with_buckets)
        for (uint32_t q = first; q < end; ++q)
          if (ctx->last_mode[q] == 8u) ctx->last_mode[q] = 9u;
    }
    const uint32_t lds = mrg::seed_lds_bytes(sp);
    (void)lds;
    uint32_t grid = (uint32_t)ctx->n_cu * (ctx->seed_wgs > 0 ? (uint32_t)ctx->seed_wgs : mrg::seed_wgs_per_cu(sp));
    if (grid > mrg::kMaxSegments) grid = mrg::kMaxSegments;
    grid = scale_grid(grid);
    const uint32_t seg_cap = segment_capacity(grid, (uint64_t)mrg::kSeedThreads * sp.reads_per_lane, have_list);
    if (!seg_cap && n) return fail(MRG_ERR_ARG, "mrg_cascade_run: survivor lists outgrew the workspace");
    sp.out_seg_cap = seg_cap;
    sp.walk_buf = nullptr;
    sp.walk_cap = 0;
    sp.walk_diag = (uint32_t)ctx->walk_diag;
    if (sp.impl) {
      // units with position lists (round 6): every wave keeps the records of the reads it leaves to them in its own stretch
      // of a context buffer and walks the lists behind its stream
      bool lists = false;
      for (uint32_t u = 0; u < sp.n_units; ++u) lists |= sp.unit[u].kind == 0u && sp.unit[u].pos_rows != nullptr;
      if (lists) {
        sp.walk_buf = ws_walk;
        sp.walk_cap = (uint32_t)std::min<int64_t>(std::max<int64_t>(ctx->walk_cap, 0), (int64_t)kWalkCap);
      }
    }
    if (n) HIP_TRY(mrg::launch_seed(sp, grid, stream));
    ctx->last_launches[first] = 1;
    for (uint32_t q = first; q < end; ++q) HIP_TRY(mark(q, q == first));
    if (sp.idx_out) {
      cur_list = next_list;
      have_list = true;
      list_fat = true;
      prev_grid = grid;
      prev_seg_cap = seg_cap;
    }
    return MRG_OK;
  };
  // a pass seed_kernel can take as (part of) a unit, and the size class of its library
  auto seedable = [&](uint32_t i) {
    if (!dict_batch || !ctx->seed_units || ctx->force_lds_mode >= 0 || !fusable(i)) return false;
    const mrg_pass_cfg& c = passes[i];
    const DevLib& l = ctx->libs[c.lib];
    if (c.max_mm_seed == 0 && l.dict_slots && c.seed_len >= (int32_t)l.dict_key) return true;
    return !l.host_seqs.empty() || l.sa16 != nullptr;
  };
  auto small_class = [&](uint32_t i) { return !ctx->libs[passes[i].lib].host_seqs.empty(); };

  uint64_t long_plan = 0;  // the lengths the lane will take (long_mask is set while the lane's cascade runs)
  bool long_plan_tried = false;
  if (split && ctx->long_lane && words_per_read >= 2 && ctx->hint_max_len > 32) {
    long_plan_tried = true;
    // a dry run of the lane's launch plan (the loop below with dict_batch = true)
    long_plan = (1ull << 31) - 1ull;
    const bool saved = dict_batch;
    dict_batch = true;
    bool la = false;
    for (uint32_t i = 0; i < n_pass && long_plan; ++i) {
      if (!runs[i]) continue;
      const mrg_pass_cfg& c = passes[i];
      const DevLib& l = ctx->libs[c.lib];
      // a seed launch: every length (the lane's first launch may be one when its pass can hold such reads: below)
      if ((la || window_bits(c)) && seedable(i)) {
        la = true;
        continue;
      }
      la = true;
      const bool pair_route = c.max_mm_seed == 2 && ctx->pair_seeds && l.pair_anchor && ctx->force_lds_mode < 0 && !c.poly_t &&
                              c.seed_len >= (int32_t)(4u * l.pair_anchor) && ctx->pair_impl != 0;
      if (pair_route) {
        for (int L = 33; L <= 63; ++L) {
          const int Lt = L - c.trim5 - c.trim3;
          if (L >= c.min_len && L <= c.max_len && Lt > 32 && Lt <= (int)l.max_ref_len) long_plan &= ~(1ull << (L - 33));
        }
      } else {
        long_plan &= ~window_bits(c);  // exact_dict_kernel, an FM kernel: only lengths its window keeps out
      }
    }
    dict_batch = saved;
  }
  uint32_t split_grid = 0, split_cap = 0;
  if (split) {
    mrg::SplitParams sp;
    sp.lens = d_lens;
    sp.nmask = d_nmask;
    sp.n_total = (uint32_t)n;
    sp.long_ok = long_plan;
    sp.nmask_hi = (d_nmask && words_per_read >= 2) ? d_nmask + n : nullptr;
    sp.min_len = (uint32_t)ctx->split_min_len;
    split_grid = std::min<uint32_t>((uint32_t)ctx->n_cu * 2u, mrg::kMaxSegments);
    split_cap = segment_capacity(split_grid, 1024, false);
    if (!split_cap) return fail(MRG_ERR_ARG, "mrg_cascade_run: survivor lists outgrew the workspace");
    sp.seg_cap = split_cap;
    sp.idx_rest = idx[0];
    sp.cnt_rest = counts;
    sp.idx_short = idx[2];
    sp.cnt_short = counts + 2 * mrg::kMaxSegments;
    HIP_TRY(mrg::launch_split(sp, split_grid, stream));
  }
  ctx->last_split = split ? 1u : 0u;
  // (a split batch: each cascade records its own per-pass events: mrg_pass_stats.ms / .ms_rest)
  for (int chain = 0; chain < (split ? 2 : 1); ++chain) {
  if (split) {
    have_list = true;
    prev_grid = split_grid;
    prev_seg_cap = split_cap;
    for (auto& b : stratum0_done) b = false;
    list_fat = false;  // (split_kernel writes index lists)
    out_init = false;
    if (chain == 0) {  // long reads and reads with N: lists in buffers 0 / 1
      pair0 = 0, pair1 = 1, cur_list = 0;
      dict_batch = false;
      evs = ctx->ev0;
      ev_ix = ctx->ev0_ix;
      ev_last = 0;
      ev_ix[0] = 0;
      if (timed) HIP_TRY(hipEventRecord(evs[0], stream));
    } else {
      evs = ctx->ev;
      ev_ix = ctx->ev_ix;
      ev_last = 0;
      ev_ix[0] = 0;
      if (timed) HIP_TRY(hipEventRecord(evs[0], stream));
    }
    if (chain == 1) {  // the one-word reads: parked in buffer 2, alternating with buffer 1
      pair0 = 2, pair1 = 1, cur_list = 2;
      dict_batch = true;
      words_eff = 1u;
      nmask_eff = nullptr;
      long_mask = long_plan;  // (... and the reads of 33..63 nt the plan lets in)
    }
  }
  bool launched_any = false;
  for (uint32_t i = 0; i < n_pass;) {
    if (!runs[i]) {
      HIP_TRY(mark(i, false));
      ++i;
      continue;
    }
    // (a batch that may hold reads of 33..63 nt: the one-word lane reads a list from its first launch on, and only a seed
    // launch can take such reads -- e.g. a batch of long reads only, whose first pass to run is the hairpin pass, len > 25)
    const bool lane_first = split && chain == 1 && ctx->long_lane && words_per_read >= 2 && long_plan_tried && window_bits(passes[i]) != 0;
    if ((launched_any || lane_first) && seedable(i)) {
      std::vector<UnitPlan> plan;
      const bool cls = small_class(i);
      uint32_t j = i;
      while (j < n_pass) {
        if (!runs[j]) {
          ++j;
          continue;
        }
        if (!seedable(j) || small_class(j) != cls) break;
        const mrg_pass_cfg& c = passes[j];
        const bool k1 = c.max_mm_seed == 0 && ctx->libs[c.lib].dict_slots != nullptr && c.seed_len >= (int32_t)ctx->libs[c.lib].dict_key;
        int join = -1;
        if (!k1 && cls)  // small libraries searched with one policy: one unit over their concatenation
          for (size_t u = 0; u < plan.size(); ++u) {
            if (plan[u].kind != 0u || plan[u].members.size() >= mrg::kSeedMaxMembers) continue;
            const mrg_pass_cfg& d = passes[plan[u].members[0]];
            if (d.max_mm_seed == c.max_mm_seed && d.trim5 == c.trim5 && d.trim3 == c.trim3 && d.min_len == c.min_len &&
                d.max_len == c.max_len && d.poly_t == c.poly_t) {
              join = (int)u;
              break;
            }
          }
        if (join >= 0) {
          plan[join].members.push_back(j);
        } else {
          if (plan.size() == mrg::kSeedMaxUnits) break;
          plan.push_back(UnitPlan{k1 ? 1u : 0u, {j}});
        }
        ++j;
      }
      // (passes skipped by the length hint at the end of the run stay with it: never the last pass)
      if (ctx->stratum0_unit && j < n_pass && runs[j] && passes[j].max_mm_seed == 2 && !passes[j].poly_t &&
          ctx->libs[passes[j].lib].dict_slots && passes[j].seed_len >= (int32_t)ctx->libs[passes[j].lib].dict_key &&
          plan.size() < mrg::kSeedMaxUnits && ctx->pair_seeds && ctx->force_lds_mode < 0) {
        UnitPlan s0{1u, {j}};
        s0.stratum0 = true;
        plan.push_back(s0);
        stratum0_done[j] = true;
      }
      int rc = run_seed(plan, i, j, j == n_pass, cls);
      if (rc != MRG_OK) return rc;
      i = j;
      continue;
    }
    uint32_t members[mrg::kMaxFused];
    uint32_t n_sub = 0, j = i;
    if (launched_any && fusable(i)) {
      const bool cls = small_lib(i);
      while (j < n_pass && n_sub < mrg::kMaxFused) {
        if (!runs[j]) {
          ++j;
          continue;
        }
        if (!fusable(j)) break;
        if (ctx->fuse != 2 && small_lib(j) != cls) break;
        if (ctx->fuse == 3 && !cls) break;
        members[n_sub++] = j++;
      }
      // passes skipped by the hint at the end of the run stay with the group (their events are
      // recorded below); a trailing skipped pass is never the last pass of the cascade
    }
    if (n_sub >= 2) {
      const bool ends = members[n_sub - 1] + 1 == n_pass;
      // events of hint-skipped passes inside the group
      int rc = run_fused(members, n_sub, ends);
      if (rc != MRG_OK) return rc;
      for (uint32_t q = i; q < j; ++q)
        if (!runs[q]) HIP_TRY(mark(q, false));
      i = j;
    } else {
      const int32_t kfull = passes[i].max_mm_seed + 1;
      int rc;
      const DevLib& l8 = ctx->libs[passes[i].lib];
      if (passes[i].max_mm_seed == 2 && ctx->pair_seeds && l8.pair_anchor && ctx->force_lds_mode < 0 && !passes[i].poly_t &&
          passes[i].seed_len >= (int32_t)(4u * l8.pair_anchor)) {
        // anchor pairs for every read long enough to hold the four anchors, all strata of the
        // pigeonhole search for the shorter ones: one launch (kernels.hip: stratum_kernel)
        rc = run_single(i, 1, kfull, !stratum0_done[i], true, true);
      } else if (passes[i].max_mm_seed == 2 && ctx->split_strata) {
        // strata 1..2 on the incoming reads, stratum 3 on the compacted survivors (kernels.hpp)
        rc = run_single(i, 1, 2, !stratum0_done[i], false);
        if (rc == MRG_OK) rc = run_single(i, 3, 3, false, true);
      } else {
        rc = run_single(i, 1, kfull, !stratum0_done[i], true);
      }
      if (rc != MRG_OK) return rc;
      ++i;
    }
    launched_any = true;
  }
  }
  ctx->pending_export_out = nullptr;
  if (d_pass_counts) {
    if (ctx->fused_step) {
      ctx->pending_export_out = d_pass_counts;  // (the tally launch behind this cascade copies them: mrg_tally_run*)
      ctx->pending_export_stream = stream;
    } else {
      HIP_TRY(mrg::launch_export_pass_counts(stats, n_pass, d_pass_counts, stream));
    }
  }
  ctx->last_stream = stream;
  ctx->last_stats_dev = stats;
  ctx->last_n_pass = n_pass;
  ++ctx->run_id;
  // a length hint describes ONE batch: it never outlives the run it was set for
  ctx->hint_min_len = 0;
  ctx->hint_max_len = 255;
  return MRG_OK;
}
}  // namespace

extern "C" {

int mrg_cascade_run(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens, const uint64_t* d_nmask,
                    uint64_t n, const mrg_pass_cfg* passes, uint32_t n_pass, int8_t* d_pass_id, int32_t* d_ref_id, int32_t* d_pos,
                    uint8_t* d_mm, uint64_t* d_pass_counts, void* d_workspace, uint64_t workspace_bytes, void* stream) {
  return cascade_run_impl(ctx, d_reads, words_per_read, d_lens, d_nmask, n, passes, n_pass, d_pass_id, d_ref_id, d_pos, d_mm, nullptr,
                          d_pass_counts, d_workspace, workspace_bytes, stream);
}

int mrg_cascade_run_packed(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens,
                           const uint64_t* d_nmask, uint64_t n, const mrg_pass_cfg* passes, uint32_t n_pass, uint32_t* d_packed,
                           uint64_t* d_pass_counts, void* d_workspace, uint64_t workspace_bytes, void* stream) {
  if (n && !d_packed) return fail(MRG_ERR_ARG, "mrg_cascade_run_packed: null output buffer");
  return cascade_run_impl(ctx, d_reads, words_per_read, d_lens, d_nmask, n, passes, n_pass, nullptr, nullptr, nullptr, nullptr,
                          d_packed, d_pass_counts, d_workspace, workspace_bytes, stream);
}

int mrg_pack_assignments(mrg_ctx* ctx, const int8_t* d_pass_id, const int32_t* d_ref_id, const int32_t* d_pos, const uint8_t* d_mm,
                         uint64_t n, uint32_t* d_packed, void* stream) {
  if (!ctx) return fail(MRG_ERR_ARG, "mrg_pack_assignments: null argument");
  if (n && (!d_pass_id || !d_ref_id || !d_pos || !d_mm || !d_packed)) return fail(MRG_ERR_ARG, "mrg_pack_assignments: null buffers");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(mrg::launch_pack_assignments(d_pass_id, d_ref_id, d_pos, d_mm, n, d_packed, (hipStream_t)stream));
  return MRG_OK;
}

int mrg_cascade_run_id(const mrg_ctx* ctx, uint64_t* run_id) {
  if (!ctx || !run_id) return fail(MRG_ERR_ARG, "mrg_cascade_run_id: null argument");
  *run_id = ctx->run_id;
  return MRG_OK;
}

int mrg_cascade_stats(mrg_ctx* ctx, mrg_pass_stats* out, uint32_t n_pass) {
  if (!ctx || !out) return fail(MRG_ERR_ARG, "mrg_cascade_stats: null argument");
  if (!ctx->last_stats_dev || n_pass != ctx->last_n_pass)
    return fail(MRG_ERR_ARG, "mrg_cascade_stats: no matching cascade run (%u vs %u passes)", n_pass,
                ctx->last_n_pass);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->last_stream));
  uint64_t host[MRG_MAX_PASSES * kStatsPerPass];
  HIP_TRY(hipMemcpy(host, ctx->last_stats_dev, (size_t)n_pass * kStatsPerPass * 8, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n_pass; ++i) {
    out[i].processed = host[i * kStatsPerPass + 0];
    out[i].aligned = host[i * kStatsPerPass + 1];
    out[i].steps = host[i * kStatsPerPass + 2];
    out[i].candidates = host[i * kStatsPerPass + 3];
    out[i].lookups = host[i * kStatsPerPass + 4];
    float ms = 0.f;
    if (ctx->last_timed) HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[ctx->ev_ix[i]], ctx->ev[ctx->ev_ix[i + 1]]));
    out[i].ms = ms;
    out[i].ms_rest = 0.f;
    if (ctx->last_split && ctx->last_timed) HIP_TRY(hipEventElapsedTime(&out[i].ms_rest, ctx->ev0[ctx->ev0_ix[i]], ctx->ev0[ctx->ev0_ix[i + 1]]));
    out[i].lds_bytes = ctx->last_lds[i];
    out[i].lds_mode = ctx->last_mode[i];
    out[i].group = ctx->last_group[i];
    out[i].n_launches = ctx->last_launches[i];
    out[i].kbits_log2 = ctx->last_kbits_log2[i];
    out[i].pair_anchor = ctx->last_pair_anchor[i];
    out[i].variant = ctx->last_variant[i];
    out[i].reserved = 0;
  }
  return MRG_OK;
}

// ------------------------------------------------------- reads of any length
int mrg_cascade_run_long(mrg_ctx* ctx, const uint64_t* d_words, const uint64_t* d_nmask, const uint64_t* d_word_off,
                         const uint32_t* d_lens, uint64_t n, const mrg_pass_cfg* passes, uint32_t n_pass, int8_t* d_pass_id,
                         int32_t* d_ref_id, int32_t* d_pos, uint8_t* d_mm, uint64_t* d_pass_counts, mrg_pass_stats* stats,
                         void* stream_) {
  if (!ctx || !passes) return fail(MRG_ERR_ARG, "mrg_cascade_run_long: null argument");
  if (n && (!d_words || !d_word_off || !d_lens || !d_pass_id || !d_ref_id || !d_pos || !d_mm))
    return fail(MRG_ERR_ARG, "mrg_cascade_run_long: null read/output buffers");
  if (n_pass == 0 || n_pass > MRG_MAX_PASSES)
    return fail(MRG_ERR_ARG, "mrg_cascade_run_long: n_pass %u not in [1,%d]", n_pass, MRG_MAX_PASSES);
  if (n >= 0xfffffff0ull) return fail(MRG_ERR_ARG, "mrg_cascade_run_long: at most 2^32-16 reads per call");
  std::vector<mrg::LongPass> host(n_pass);
  for (uint32_t i = 0; i < n_pass; ++i) {
    const mrg_pass_cfg& c = passes[i];
    if (c.lib < 0 || (size_t)c.lib >= ctx->libs.size())
      return fail(MRG_ERR_ARG, "mrg_cascade_run_long: pass %u names unknown library %d", i, c.lib);
    if (c.max_mm_seed < 0 || c.max_mm_seed > 3 || c.max_mm_total < c.max_mm_seed || c.max_mm_total > 255 || c.trim5 < 0 ||
        c.trim3 < 0 || c.seed_len < 1)
      return fail(MRG_ERR_ARG, "mrg_cascade_run_long: pass %u has an invalid policy", i);
    const DevLib& l = ctx->libs[c.lib];
    mrg::LongPass& q = host[i];
    std::memset(&q, 0, sizeof q);
    q.blocks = l.blocks;
    q.super = l.super;
    q.text = l.text;
    q.sa = l.sa;
    q.ftab = l.ftab;
    q.tabs = l.tabs;
    if (!ctx->use_ftab || !l.ftab) q.tabs.k[0] = 0u;
    q.seg_start = l.seg_start;
    q.seg_ref = l.seg_ref;
    q.seg_off = l.seg_off;
    q.chunk_seg = l.chunk_seg;
    q.n = l.n;
    q.primary = l.primary;
    q.simple_segs = l.simple ? 1u : 0u;
    q.seed_len = c.seed_len;
    q.max_mm_seed = c.max_mm_seed;
    q.max_mm_total = c.max_mm_total;
    q.trim5 = c.trim5;
    q.trim3 = c.trim3;
    q.min_len = c.min_len;
    q.max_len = c.max_len;
    q.poly_t = c.poly_t;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  struct Scratch {
    void* p = nullptr;
    ~Scratch() { (void)hipFree(p); }
  } scratch;
  const size_t table_bytes = ((size_t)n_pass * sizeof(mrg::LongPass) + 255) / 256 * 256;
  const size_t stats_bytes = (size_t)n_pass * kStatsPerPass * 8;
  HIP_TRY(hipMalloc(&scratch.p, table_bytes + stats_bytes));
  uint64_t* d_stats = reinterpret_cast<uint64_t*>((char*)scratch.p + table_bytes);
  HIP_TRY(hipMemcpyAsync(scratch.p, host.data(), (size_t)n_pass * sizeof(mrg::LongPass), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemsetAsync(d_stats, 0, stats_bytes, stream));
  struct Events {  // (destroyed on every way out: advisor, round 5)
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evs_long;
  HIP_TRY(hipEventCreate(&evs_long.a));
  HIP_TRY(hipEventCreate(&evs_long.b));
  const hipEvent_t e0 = evs_long.a, e1 = evs_long.b;
  hipError_t err = hipEventRecord(e0, stream);
  if (err == hipSuccess && n) {
    mrg::LongParams p;
    std::memset(&p, 0, sizeof p);
    p.pass = reinterpret_cast<const mrg::LongPass*>(scratch.p);
    p.n_pass = n_pass;
    p.words = d_words;
    p.nmask = d_nmask;
    p.word_off = d_word_off;
    p.lens = d_lens;
    p.n = (uint32_t)n;
    p.wstop = (uint32_t)ctx->wstop;
    p.pass_id = d_pass_id;
    p.ref_id = d_ref_id;
    p.pos = d_pos;
    p.mm = d_mm;
    p.counters = d_stats;
    p.pass_counts = d_pass_counts;
    // one wave per read, four per workgroup; more reads than resident waves: the kernel strides
    const uint64_t want = (n + 3) / 4;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->n_cu * 8u));
    err = mrg::launch_long_reads(p, grid, stream);
  }
  if (err == hipSuccess) err = hipEventRecord(e1, stream);
  uint64_t got[MRG_MAX_PASSES * kStatsPerPass];
  if (err == hipSuccess) err = hipMemcpyAsync(got, d_stats, stats_bytes, hipMemcpyDeviceToHost, stream);
  if (err == hipSuccess) err = hipStreamSynchronize(stream);
  float ms = 0.f;
  if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
  HIP_TRY(err);
  if (stats) {
    uint64_t offered = 0;
    for (uint32_t i = 0; i < n_pass; ++i) offered += got[i * kStatsPerPass];
    for (uint32_t i = 0; i < n_pass; ++i) {
      stats[i].processed += got[i * kStatsPerPass + 0];
      stats[i].aligned += got[i * kStatsPerPass + 1];
      stats[i].steps += got[i * kStatsPerPass + 2];
      stats[i].candidates += got[i * kStatsPerPass + 3];
      stats[i].lookups += got[i * kStatsPerPass + 4];
      // one launch carries every pass: its time is shared out by the reads each pass was offered
      if (offered) stats[i].ms += ms * (float)((double)got[i * kStatsPerPass] / (double)offered);
    }
  }
  return MRG_OK;
}

// ----------------------------------------------------------------- tally
int mrg_tally_counts_len(uint32_t n_mirna, uint32_t n_samples, uint32_t n_pass, uint64_t* len) {
  if (!len) return fail(MRG_ERR_ARG, "mrg_tally_counts_len: null argument");
  *len = 2ull * n_mirna * n_samples + (uint64_t)(n_pass + 1) * n_samples + n_samples;
  return MRG_OK;
}

}  // extern "C"

namespace {
int tally_run_impl(mrg_ctx* ctx, const int8_t* d_pass_id, const int32_t* d_ref_id, const uint32_t* d_packed,
                  const uint32_t* d_quant, uint64_t n, uint32_t n_samples, uint32_t n_mirna,
                  uint32_t n_pass, int32_t canon_pass, int32_t isomir_pass, uint64_t* d_counts,
                  void* stream_) {
  if (!ctx || !d_counts) return fail(MRG_ERR_ARG, "mrg_tally_run: null argument");
  if (n && ((!d_packed && (!d_pass_id || !d_ref_id)) || !d_quant)) return fail(MRG_ERR_ARG, "mrg_tally_run: null buffers");
  if (n_samples == 0 || n_pass == 0 || n_pass > MRG_MAX_PASSES)
    return fail(MRG_ERR_ARG, "mrg_tally_run: bad n_samples/n_pass");
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return MRG_OK;
  mrg::TallyParams p;
  p.packed = d_packed;
  p.pass_id = d_pass_id;
  p.ref_id = d_ref_id;
  p.quant = d_quant;
  p.n = n;
  p.n_samples = n_samples;
  p.n_mirna = n_mirna;
  p.n_pass = n_pass;
  p.canon_pass = canon_pass;
  p.isomir_pass = isomir_pass;
  p.counts = d_counts;
  p.export_stats = nullptr;
  p.export_out = nullptr;
  p.export_n_pass = 0;
  if (ctx->pending_export_out && ctx->pending_export_stream == (hipStream_t)stream_) {
    // "fused_step": the cascade in front left its per-pass counters to this launch
    p.export_stats = ctx->last_stats_dev;
    p.export_out = ctx->pending_export_out;
    p.export_n_pass = ctx->last_n_pass;
  }
  ctx->pending_export_out = nullptr;
  const bool in_ok = d_packed ? ((uintptr_t)d_packed % 16 == 0) : (((uintptr_t)d_pass_id % 4 == 0) && ((uintptr_t)d_ref_id % 16 == 0));
  p.vec4 = (n_samples == 1 && in_ok && ((uintptr_t)d_quant % 16 == 0)) ? 1u : 0u;
  uint64_t bins = 0;
  mrg_tally_counts_len(n_mirna, n_samples, n_pass, &bins);
  // LDS histogram: the category bins are replicated (kernels.hip: tally_kernel)
  const uint64_t lds = (bins + (uint64_t)(n_pass + 1) * n_samples * (mrg::kTallyCatReplicas - 1)) * 8;
  const bool lds_hist = lds <= (uint64_t)ctx->lds_budget;
  uint64_t want = ((p.vec4 ? (n + 3) / 4 : n) + mrg::kTallyThreads - 1) / mrg::kTallyThreads;
  uint32_t per_cu = lds_hist ? (lds * 2 <= 160 * 1024 ? 2u : 1u) : 2u;
  uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)ctx->n_cu * per_cu);
  HIP_TRY(mrg::launch_tally(p, lds_hist, grid, lds_hist ? (uint32_t)lds : 0u, (hipStream_t)stream_));
  return MRG_OK;
}
}  // namespace

extern "C" {

int mrg_tally_run(mrg_ctx* ctx, const int8_t* d_pass_id, const int32_t* d_ref_id, const uint32_t* d_quant, uint64_t n,
                  uint32_t n_samples, uint32_t n_mirna, uint32_t n_pass, int32_t canon_pass, int32_t isomir_pass, uint64_t* d_counts,
                  void* stream) {
  return tally_run_impl(ctx, d_pass_id, d_ref_id, nullptr, d_quant, n, n_samples, n_mirna, n_pass, canon_pass, isomir_pass, d_counts, stream);
}

int mrg_tally_run_packed(mrg_ctx* ctx, const uint32_t* d_packed, const uint32_t* d_quant, uint64_t n, uint32_t n_samples,
                         uint32_t n_mirna, uint32_t n_pass, int32_t canon_pass, int32_t isomir_pass, uint64_t* d_counts, void* stream) {
  if (n && !d_packed) return fail(MRG_ERR_ARG, "mrg_tally_run_packed: null buffers");
  // (the packed word's entry field saturates at 0x3FFFF: a larger library would be tallied into the wrong bin)
  if (n_mirna > 0x3FFFFu)
    return fail(MRG_ERR_ARG, "mrg_tally_run_packed: %u miRNA entries do not fit the packed word's 18-bit entry field (use mrg_tally_run)", n_mirna);
  return tally_run_impl(ctx, nullptr, nullptr, d_packed, d_quant, n, n_samples, n_mirna, n_pass, canon_pass, isomir_pass, d_counts, stream);
}

// ------------------------------------------------------------ multi-GPU (RCCL over xGMI)
#define RCCL_TRY(api, expr)                                                                         \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess)                                                                          \
      return fail(MRG_ERR_HIP, "%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(r_) : "?"); \
  } while (0)

int mrg_comm_unique_id(void* id128) {
  if (!id128) return fail(MRG_ERR_ARG, "mrg_comm_unique_id: null argument");
  RcclApi* api = rccl_api();
  if (!api->handle) return fail(MRG_ERR_IO, "mrg_comm_unique_id: %s", api->error.c_str());
  ncclUniqueId id;
  RCCL_TRY(api, api->GetUniqueId(&id));
  static_assert(sizeof(id) == MRG_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof id);
  return MRG_OK;
}

int mrg_comm_init(mrg_ctx* ctx, const void* id128, int32_t rank, int32_t world) {
  if (!ctx || !id128) return fail(MRG_ERR_ARG, "mrg_comm_init: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(MRG_ERR_ARG, "mrg_comm_init: rank %d of %d", rank, world);
  if (ctx->comm) return fail(MRG_ERR_ARG, "mrg_comm_init: the context already has a communicator");
  RcclApi* api = rccl_api();
  if (!api->handle) return fail(MRG_ERR_IO, "mrg_comm_init: %s", api->error.c_str());
  HIP_TRY(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  RCCL_TRY(api, api->CommInitRank(&ctx->comm, world, id, rank));
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return MRG_OK;
}

int mrg_allreduce(mrg_ctx* ctx, uint64_t* d_buf, uint64_t n, void* stream) {
  if (!ctx || (n && !d_buf)) return fail(MRG_ERR_ARG, "mrg_allreduce: null argument");
  if (!ctx->comm) {
    if (ctx->comm_world == 1) return MRG_OK;  // one process, nothing to add
    return fail(MRG_ERR_ARG, "mrg_allreduce: call mrg_comm_init first");
  }
  if (n == 0) return MRG_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  RcclApi* api = rccl_api();
  RCCL_TRY(api, api->AllReduce(d_buf, d_buf, (size_t)n, ncclUint64, ncclSum, ctx->comm, (hipStream_t)stream));
  return MRG_OK;
}

int mrg_comm_destroy(mrg_ctx* ctx) {
  if (!ctx) return fail(MRG_ERR_ARG, "mrg_comm_destroy: null argument");
  if (ctx->comm) {
    RcclApi* api = rccl_api();
    ncclComm_t c = ctx->comm;
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    RCCL_TRY(api, api->CommDestroy(c));
  }
  return MRG_OK;
}

// ------------------------------------------------------------ A-to-I position tally
int mrg_edit_counts_len(uint32_t n_bins, uint32_t n_samples, uint64_t* len) {
  if (!len) return fail(MRG_ERR_ARG, "mrg_edit_counts_len: null argument");
  *len = (uint64_t)n_bins * n_samples * (3ull + mrg::kEditPositions);
  return MRG_OK;
}

}  // extern "C"

namespace {
int edit_tally_impl(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens,
                       const uint64_t* d_nmask, const int8_t* d_pass_id, const int32_t* d_ref_id,
                       const int32_t* d_pos, const uint32_t* d_packed, const uint32_t* d_quant, const uint8_t* d_keep,
                       const uint32_t* d_remap, uint64_t n, uint32_t n_samples, uint32_t n_bins, int32_t lib,
                       int32_t canon_pass, int32_t isomir_pass, int32_t isomir_trim5, uint32_t flank5,
                       uint32_t flank3, uint32_t from_base, uint32_t to_base, uint64_t* d_counts, void* stream) {
  if (!ctx || !d_counts) return fail(MRG_ERR_ARG, "mrg_edit_tally_run: null argument");
  if (n && (!d_reads || !d_lens || (!d_packed && (!d_pass_id || !d_ref_id || !d_pos)) || !d_quant))
    return fail(MRG_ERR_ARG, "mrg_edit_tally_run: null buffers");
  if (lib < 0 || (size_t)lib >= ctx->libs.size()) return fail(MRG_ERR_ARG, "mrg_edit_tally_run: unknown library %d", lib);
  if (words_per_read != 1 && words_per_read != 2 && words_per_read != 4 && words_per_read != 8)
    return fail(MRG_ERR_ARG, "mrg_edit_tally_run: words_per_read must be 1, 2, 4 or 8");
  if (n_samples == 0 || n_bins == 0 || from_base > 3 || to_base > 3 || from_base == to_base)
    return fail(MRG_ERR_ARG, "mrg_edit_tally_run: bad n_samples / n_bins / bases");
  const DevLib& l = ctx->libs[lib];
  if (l.n_seg != l.n_ref || !l.simple)
    return fail(MRG_ERR_ARG, "mrg_edit_tally_run: the miRNA library must hold one N-free segment per entry");
  if (!d_remap && n_bins != l.n_ref)
    return fail(MRG_ERR_ARG, "mrg_edit_tally_run: n_bins %u != %u library entries (no remap given)", n_bins, l.n_ref);
  if (l.max_ref_len > mrg::kEditPositions + flank5 + flank3)
    return fail(MRG_ERR_ARG, "mrg_edit_tally_run: a library entry is longer than %u + flanks", mrg::kEditPositions);
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return MRG_OK;
  mrg::EditParams p;
  p.reads = d_reads;
  p.lens = d_lens;
  p.nmask = d_nmask;
  p.words_per_read = words_per_read;
  p.packed = d_packed;
  p.pass_id = d_pass_id;
  p.ref_id = d_ref_id;
  p.pos = d_pos;
  p.quant = d_quant;
  p.keep = d_keep;
  p.remap = d_remap;
  p.n = n;
  p.n_samples = n_samples;
  p.n_bins = n_bins;
  p.canon_pass = canon_pass;
  p.isomir_pass = isomir_pass;
  p.isomir_trim5 = isomir_trim5;
  p.flank5 = flank5;
  p.flank3 = flank3;
  p.from_base = from_base;
  p.to_base = to_base;
  p.text = l.text;
  p.seg_start = l.seg_start;
  p.counts = d_counts;
  const bool in_ok = d_packed ? ((uintptr_t)d_packed % 16 == 0)
                              : (((uintptr_t)d_pass_id % 4 == 0) && ((uintptr_t)d_ref_id % 16 == 0) && ((uintptr_t)d_pos % 16 == 0));
  p.vec4 = (n_samples == 1 && words_per_read == 1 && ((uintptr_t)d_reads % 16 == 0) && ((uintptr_t)d_lens % 4 == 0) && in_ok &&
            ((uintptr_t)d_quant % 16 == 0))
               ? 1u
               : 0u;
  p.text_words = l.text_words;
  p.n_entries = l.n_ref;
  // LDS: the per-entry totals first (every kept read hits them), then -- if two workgroups per CU
  // still fit -- the library's text and entry starts, so that a read costs no gather
  const uint64_t hist_b = mrg::edit_hist_lds_bytes(n_bins, n_samples);
  const uint64_t lib_b = (uint64_t)l.text_words * 4 + (((uint64_t)l.n_ref + 4) & ~3ull) * 4;
  const uint64_t budget = (uint64_t)std::max<int64_t>(ctx->lds_budget, (int64_t)mrg::kEditHashLdsBytes) - mrg::kEditHashLdsBytes;
  const bool lds_hist = hist_b <= budget;
  const bool lds_lib = (lds_hist ? hist_b : 0) + lib_b + mrg::kEditHashLdsBytes <= std::min<uint64_t>(budget, 80 * 1024);
  const uint64_t lds = (lds_hist ? hist_b : 0) + (lds_lib ? lib_b : 0) + mrg::kEditHashLdsBytes;
  const uint64_t want = ((p.vec4 ? (n + 3) / 4 : n) + mrg::kEditThreads - 1) / mrg::kEditThreads;
  const uint32_t per_cu = lds * 2 <= 160 * 1024 ? 2u : 1u;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)ctx->n_cu * per_cu);
  HIP_TRY(mrg::launch_edit_tally(p, lds_hist, lds_lib, grid, (uint32_t)lds, (hipStream_t)stream));
  return MRG_OK;
}
}  // namespace

extern "C" {

int mrg_edit_tally_run(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens, const uint64_t* d_nmask,
                       const int8_t* d_pass_id, const int32_t* d_ref_id, const int32_t* d_pos, const uint32_t* d_quant,
                       const uint8_t* d_keep, const uint32_t* d_remap, uint64_t n, uint32_t n_samples, uint32_t n_bins, int32_t lib,
                       int32_t canon_pass, int32_t isomir_pass, int32_t isomir_trim5, uint32_t flank5, uint32_t flank3,
                       uint32_t from_base, uint32_t to_base, uint64_t* d_counts, void* stream) {
  return edit_tally_impl(ctx, d_reads, words_per_read, d_lens, d_nmask, d_pass_id, d_ref_id, d_pos, nullptr, d_quant, d_keep, d_remap, n,
                         n_samples, n_bins, lib, canon_pass, isomir_pass, isomir_trim5, flank5, flank3, from_base, to_base, d_counts,
                         stream);
}

int mrg_edit_tally_run_packed(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens,
                              const uint64_t* d_nmask, const uint32_t* d_packed, const uint32_t* d_quant, const uint8_t* d_keep,
                              const uint32_t* d_remap, uint64_t n, uint32_t n_samples, uint32_t n_bins, int32_t lib, int32_t canon_pass,
                              int32_t isomir_pass, int32_t isomir_trim5, uint32_t flank5, uint32_t flank3, uint32_t from_base,
                              uint32_t to_base, uint64_t* d_counts, void* stream) {
  if (n && !d_packed) return fail(MRG_ERR_ARG, "mrg_edit_tally_run_packed: null buffers");
  // (entry and offset fields of the packed word saturate at 0x3FFFF / 0xFF)
  if (ctx && lib >= 0 && (size_t)lib < ctx->libs.size() && (ctx->libs[lib].n_ref > 0x3FFFFu || ctx->libs[lib].max_ref_len >= 0xFFu))
    return fail(MRG_ERR_ARG, "mrg_edit_tally_run_packed: the library (%u entries, longest %u) does not fit the packed word (use mrg_edit_tally_run)",
                ctx->libs[lib].n_ref, ctx->libs[lib].max_ref_len);
  return edit_tally_impl(ctx, d_reads, words_per_read, d_lens, d_nmask, nullptr, nullptr, nullptr, d_packed, d_quant, d_keep, d_remap, n,
                         n_samples, n_bins, lib, canon_pass, isomir_pass, isomir_trim5, flank5, flank3, from_base, to_base, d_counts,
                         stream);
}

// ------------------------------------------------------------ count best
namespace {

int fill_count_params(mrg_ctx* ctx, const char* who, const uint64_t* d_reads, uint32_t words_per_read,
                      const uint8_t* d_lens, const uint64_t* d_nmask, uint64_t n, int32_t lib, int32_t seed_len,
                      int32_t max_mm_seed, int32_t max_mm_total, mrg::CountParams* p, uint32_t* grid,
                      uint32_t* lds_bytes) {
  if (!ctx) return fail(MRG_ERR_ARG, "%s: null argument", who);
  if (n && (!d_reads || !d_lens)) return fail(MRG_ERR_ARG, "%s: null buffers", who);
  if (lib < 0 || (size_t)lib >= ctx->libs.size()) return fail(MRG_ERR_ARG, "%s: unknown library %d", who, lib);
  if (words_per_read != 1 && words_per_read != 2 && words_per_read != 4 && words_per_read != 8)
    return fail(MRG_ERR_ARG, "%s: words_per_read must be 1, 2, 4 or 8", who);
  if (max_mm_seed < 0 || max_mm_seed > 2 || max_mm_total < max_mm_seed || seed_len < 1)
    return fail(MRG_ERR_ARG, "%s: invalid policy", who);
  if (n >= 0x7fffffffull) return fail(MRG_ERR_ARG, "%s: too many reads for one call", who);
  const DevLib& l = ctx->libs[lib];
  const uint64_t lds = (uint64_t)l.nsup * 16;
  if (lds > 160 * 1024) return fail(MRG_ERR_ARG, "%s: library too large for one call (split it)", who);
  std::memset(p, 0, sizeof(*p));
  p->blocks = l.blocks;
  p->super = l.super;
  p->text = l.text;
  p->sa = l.sa;
  p->ctx = l.ctx;
  p->ftab = l.ftab;
  p->tabs = l.tabs;
  if (!ctx->use_ftab) p->tabs.k[0] = 0u;
  p->n = l.n;
  p->nsup = l.nsup;
  p->primary = l.primary;
  p->seg_start = l.seg_start;
  p->seg_ref = l.seg_ref;
  p->seg_off = l.seg_off;
  p->chunk_seg = l.chunk_seg;
  p->reads = d_reads;
  p->lens = d_lens;
  p->nmask = d_nmask;
  p->n_reads = n;
  p->seed_len = seed_len;
  p->max_mm_seed = max_mm_seed;
  p->max_mm_total = max_mm_total;
  p->wstop = (uint32_t)ctx->wstop;
  p->max_rows = 4096u;
  const uint64_t want = (n + mrg::kCountThreads - 1) / mrg::kCountThreads;
  const uint32_t per_cu = lds ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(8, (160 * 1024) / lds)) : 8u;
  *grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->n_cu * per_cu));
  *lds_bytes = (uint32_t)lds;
  return MRG_OK;
}

}  // namespace

int mrg_count_best(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens,
                   const uint64_t* d_nmask, uint64_t n, int32_t lib, int32_t seed_len, int32_t max_mm_seed,
                   int32_t max_mm_total, uint8_t* d_best_mm, uint8_t* d_count, void* stream) {
  mrg::CountParams p;
  uint32_t grid = 0, lds = 0;
  int rc = fill_count_params(ctx, "mrg_count_best", d_reads, words_per_read, d_lens, d_nmask, n, lib, seed_len,
                             max_mm_seed, max_mm_total, &p, &grid, &lds);
  if (rc != MRG_OK) return rc;
  if (n && (!d_best_mm || !d_count)) return fail(MRG_ERR_ARG, "mrg_count_best: null buffers");
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) return MRG_OK;
  p.best_mm = d_best_mm;
  p.count = d_count;
  // one-mismatch run on one-word reads without N: the largest jump table's K-mers and their variants first (kernels.hip:
  // count_variants_kernel); the pigeonhole kernel then takes what that left (reads shorter than a usable table, longer than
  // the seed) -- nothing, for the 18..21-nt reads the -ai path submits
  if (ctx->count_variants && words_per_read == 1 && !d_nmask && max_mm_seed == 1 && ctx->use_ftab) {
    const uint64_t want = (n + 7) / 8;
    const uint32_t vgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->n_cu * 32u));
    HIP_TRY(mrg::launch_count_variants(p, vgrid, (hipStream_t)stream));
    p.only_todo = 1u;
  }
  HIP_TRY(mrg::launch_count(p, words_per_read, grid, lds, (hipStream_t)stream));
  return MRG_OK;
}

int mrg_list_best_count(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens,
                        const uint64_t* d_nmask, uint64_t n, int32_t lib, int32_t seed_len, int32_t max_mm_seed,
                        int32_t max_mm_total, uint8_t* d_best_mm, uint64_t* d_offsets, uint64_t* total,
                        void* stream) {
  mrg::CountParams p;
  uint32_t grid = 0, lds = 0;
  int rc = fill_count_params(ctx, "mrg_list_best_count", d_reads, words_per_read, d_lens, d_nmask, n, lib,
                             seed_len, max_mm_seed, max_mm_total, &p, &grid, &lds);
  if (rc != MRG_OK) return rc;
  if (!d_offsets || !total || (n && !d_best_mm)) return fail(MRG_ERR_ARG, "mrg_list_best_count: null buffers");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  // per-read stratum sizes live in the context's scratch buffer (grown, never shrunk: repeated
  // calls do not allocate)
  const uint64_t need_bytes = (n + 1) * sizeof(uint32_t);
  if (ctx->scratch_bytes < need_bytes) {
    HIP_TRY(hipStreamSynchronize(st));
    (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->scratch, need_bytes));
    ctx->scratch_bytes = need_bytes;
  }
  uint32_t* cnt = (uint32_t*)ctx->scratch;
  hipError_t e = hipMemsetAsync(cnt, 0, (n + 1) * sizeof(uint32_t), st);
  if (e == hipSuccess && n) {
    p.best_mm = d_best_mm;
    p.count32 = cnt;
    p.max_rows = 1u << 20;
    e = mrg::launch_count(p, words_per_read, grid, lds, st);
  }
  if (e == hipSuccess) e = mrg::exclusive_sum_u32_u64(cnt, d_offsets, n + 1, st);
  if (e == hipSuccess) e = hipMemcpyAsync(total, d_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  HIP_TRY(e);
  return MRG_OK;
}

int mrg_list_best_fill(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens,
                       const uint64_t* d_nmask, uint64_t n, int32_t lib, int32_t seed_len, int32_t max_mm_seed,
                       int32_t max_mm_total, const uint8_t* d_best_mm, const uint64_t* d_offsets, uint64_t cap,
                       int32_t* d_ref, int32_t* d_pos, void* stream) {
  mrg::CountParams p;
  uint32_t grid = 0, lds = 0;
  int rc = fill_count_params(ctx, "mrg_list_best_fill", d_reads, words_per_read, d_lens, d_nmask, n, lib,
                             seed_len, max_mm_seed, max_mm_total, &p, &grid, &lds);
  if (rc != MRG_OK) return rc;
  if (n && (!d_best_mm || !d_offsets)) return fail(MRG_ERR_ARG, "mrg_list_best_fill: null buffers");
  if (cap && (!d_ref || !d_pos)) return fail(MRG_ERR_ARG, "mrg_list_best_fill: null output buffers");
  HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0 || cap == 0) return MRG_OK;
  p.best_mm = const_cast<uint8_t*>(d_best_mm);
  p.offsets = d_offsets;
  p.out_ref = d_ref;
  p.out_pos = d_pos;
  p.out_cap = cap;
  p.max_rows = 1u << 20;
  HIP_TRY(mrg::launch_count(p, words_per_read, grid, lds, (hipStream_t)stream));
  return MRG_OK;
}

// ----------------------------------------------------- host convenience
int mrg_annotate_host(mrg_ctx* ctx, const uint64_t* reads, uint32_t words_per_read,
                      const uint8_t* lens, const uint64_t* nmask, uint64_t n,
                      const mrg_pass_cfg* passes, uint32_t n_pass, int8_t* pass_id, int32_t* ref_id,
                      int32_t* pos, uint8_t* mm, mrg_pass_stats* stats, const uint32_t* quant,
                      uint32_t n_samples, uint32_t n_mirna, int32_t canon_pass, int32_t isomir_pass,
                      uint64_t* counts) {
  if (!ctx || !passes) return fail(MRG_ERR_ARG, "mrg_annotate_host: null argument");
  if (n && (!reads || !lens || !pass_id || !ref_id || !pos || !mm))
    return fail(MRG_ERR_ARG, "mrg_annotate_host: null read/output buffers");
  HIP_TRY(hipSetDevice(ctx->device));
  struct Bufs {
    std::vector<void*> v;
    ~Bufs() {
      for (void* p : v) (void)hipFree(p);
    }
    int get(void** p, size_t bytes) {
      hipError_t e = hipMalloc(p, bytes ? bytes : 16);
      if (e != hipSuccess) return fail(MRG_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      v.push_back(*p);
      return MRG_OK;
    }
  } bufs;
  int rc;
  uint64_t *d_reads = nullptr, *d_nmask = nullptr, *d_counts = nullptr, *d_pc = nullptr;
  uint8_t *d_lens = nullptr, *d_mm = nullptr;
  int8_t* d_pass = nullptr;
  int32_t *d_ref = nullptr, *d_pos = nullptr;
  uint32_t* d_quant = nullptr;
  void* d_ws = nullptr;
  uint64_t ws_bytes = 0;
  mrg_cascade_workspace_bytes(n, &ws_bytes);
  const size_t rbytes = (size_t)n * words_per_read * 8;
  if ((rc = bufs.get((void**)&d_reads, rbytes))) return rc;
  if ((rc = bufs.get((void**)&d_lens, n))) return rc;
  if (nmask && (rc = bufs.get((void**)&d_nmask, rbytes))) return rc;
  if ((rc = bufs.get((void**)&d_pass, n))) return rc;
  if ((rc = bufs.get((void**)&d_ref, n * 4))) return rc;
  if ((rc = bufs.get((void**)&d_pos, n * 4))) return rc;
  if ((rc = bufs.get((void**)&d_mm, n))) return rc;
  if ((rc = bufs.get((void**)&d_pc, 2 * MRG_MAX_PASSES * 8))) return rc;
  if ((rc = bufs.get(&d_ws, ws_bytes))) return rc;
  if (n) {
    HIP_TRY(hipMemcpy(d_reads, reads, rbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_lens, lens, n, hipMemcpyHostToDevice));
    if (nmask) HIP_TRY(hipMemcpy(d_nmask, nmask, rbytes, hipMemcpyHostToDevice));
  }
  {
    // the lengths are on the host here: give the cascade the exact range of this batch (and put
    // the caller's own hints back afterwards)
    uint8_t lo = 255, hi = 0;
    for (uint64_t r = 0; r < n; ++r) {
      lo = std::min(lo, lens[r]);
      hi = std::max(hi, lens[r]);
    }
    ctx->hint_min_len = n ? lo : 0;
    ctx->hint_max_len = n ? hi : 255;
    rc = mrg_cascade_run(ctx, d_reads, words_per_read, d_lens, d_nmask, n, passes, n_pass, d_pass, d_ref,
                         d_pos, d_mm, d_pc, d_ws, ws_bytes, nullptr);
  }
  if (rc) return rc;
  std::vector<mrg_pass_stats> st(n_pass);
  if ((rc = mrg_cascade_stats(ctx, st.data(), n_pass))) return rc;
  if (stats) std::memcpy(stats, st.data(), n_pass * sizeof(mrg_pass_stats));
  if (counts) {
    if (!quant || n_samples == 0) return fail(MRG_ERR_ARG, "mrg_annotate_host: counts requested without quant");
    uint64_t clen = 0;
    mrg_tally_counts_len(n_mirna, n_samples, n_pass, &clen);
    if ((rc = bufs.get((void**)&d_counts, clen * 8))) return rc;
    if ((rc = bufs.get((void**)&d_quant, n * n_samples * 4))) return rc;
    HIP_TRY(hipMemset(d_counts, 0, clen * 8));
    if (n) HIP_TRY(hipMemcpy(d_quant, quant, n * n_samples * 4, hipMemcpyHostToDevice));
    if ((rc = mrg_tally_run(ctx, d_pass, d_ref, d_quant, n, n_samples, n_mirna, n_pass, canon_pass,
                            isomir_pass, d_counts, nullptr)))
      return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(counts, d_counts, clen * 8, hipMemcpyDeviceToHost));
  }
  if (n) {
    HIP_TRY(hipMemcpy(pass_id, d_pass, n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ref_id, d_ref, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pos, d_pos, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mm, d_mm, n, hipMemcpyDeviceToHost));
  }
  return MRG_OK;
}

// the long-read lane for a caller without a device allocator (the ctypes stub of INTEGRATION.md)
int mrg_annotate_long_host(mrg_ctx* ctx, const uint64_t* words, const uint64_t* nmask, const uint64_t* word_off, const uint32_t* lens,
                           uint64_t n, const mrg_pass_cfg* passes, uint32_t n_pass, int8_t* pass_id, int32_t* ref_id, int32_t* pos,
                           uint8_t* mm, mrg_pass_stats* stats) {
  if (!ctx || !passes) return fail(MRG_ERR_ARG, "mrg_annotate_long_host: null argument");
  if (n && (!words || !word_off || !lens || !pass_id || !ref_id || !pos || !mm))
    return fail(MRG_ERR_ARG, "mrg_annotate_long_host: null read/output buffers");
  if (n == 0) return MRG_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  struct Bufs {
    std::vector<void*> v;
    ~Bufs() {
      for (void* p : v) (void)hipFree(p);
    }
    int get(void** p, size_t bytes) {
      hipError_t e = hipMalloc(p, bytes ? bytes : 16);
      if (e != hipSuccess) return fail(MRG_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      v.push_back(*p);
      return MRG_OK;
    }
  } bufs;
  const size_t wbytes = (size_t)word_off[n] * 8;
  uint64_t *d_words = nullptr, *d_nmask = nullptr, *d_off = nullptr;
  uint32_t* d_lens = nullptr;
  int8_t* d_pass = nullptr;
  int32_t *d_ref = nullptr, *d_pos = nullptr;
  uint8_t* d_mm = nullptr;
  int rc;
  if ((rc = bufs.get((void**)&d_words, wbytes))) return rc;
  if (nmask && (rc = bufs.get((void**)&d_nmask, wbytes))) return rc;
  if ((rc = bufs.get((void**)&d_off, (n + 1) * 8))) return rc;
  if ((rc = bufs.get((void**)&d_lens, n * 4))) return rc;
  if ((rc = bufs.get((void**)&d_pass, n))) return rc;
  if ((rc = bufs.get((void**)&d_ref, n * 4))) return rc;
  if ((rc = bufs.get((void**)&d_pos, n * 4))) return rc;
  if ((rc = bufs.get((void**)&d_mm, n))) return rc;
  if (wbytes) HIP_TRY(hipMemcpy(d_words, words, wbytes, hipMemcpyHostToDevice));
  if (nmask && wbytes) HIP_TRY(hipMemcpy(d_nmask, nmask, wbytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_off, word_off, (n + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_lens, lens, n * 4, hipMemcpyHostToDevice));
  if ((rc = mrg_cascade_run_long(ctx, d_words, d_nmask, d_off, d_lens, n, passes, n_pass, d_pass, d_ref, d_pos, d_mm, nullptr, stats,
                                 nullptr)))
    return rc;
  HIP_TRY(hipMemcpy(pass_id, d_pass, n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ref_id, d_ref, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(pos, d_pos, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(mm, d_mm, n, hipMemcpyDeviceToHost));
  return MRG_OK;
}

// -------------------------------------------------------------- ingest
int mrg_adapter_locate(const char* adapter, const char* read, double max_error_rate, int32_t min_overlap,
                       int32_t* out6) {
  if (!adapter || !read || !out6) return fail(MRG_ERR_ARG, "mrg_adapter_locate: null argument");
  try {
    const mrg::AdapterMatch m =
        mrg::locate_adapter_3p(adapter, read, std::strlen(read), max_error_rate, min_overlap);
    out6[0] = m.found ? 1 : 0;
    out6[1] = (int32_t)m.read_start;
    out6[2] = (int32_t)m.read_stop;
    out6[3] = m.adapter_stop;
    out6[4] = m.matches;
    out6[5] = m.errors;
    return MRG_OK;
  } catch (const std::exception& e) {
    return fail(MRG_ERR_NOMEM, "mrg_adapter_locate: %s", e.what());
  }
}

int mrg_fastq_load(const char* path, int32_t qual_cutoff, int32_t min_len, const char* adapter, int32_t threads,
                   mrg_fastq** out) {
  if (!path || !out) return fail(MRG_ERR_ARG, "mrg_fastq_load: null argument");
  try {
    auto h = std::make_unique<mrg_fastq>();
    mrg::load_fastq(path, qual_cutoff, min_len, adapter, threads, h->d);
    *out = h.release();
    return MRG_OK;
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_fastq_load: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_fastq_load: %s", e.what());
  }
}

int mrg_fastq_load_part(const char* path, int32_t qual_cutoff, int32_t min_len, const char* adapter, int32_t threads, int32_t part,
                        int32_t n_parts, mrg_fastq** out) {
  if (!path || !out) return fail(MRG_ERR_ARG, "mrg_fastq_load_part: null argument");
  if (n_parts < 1 || part < 0 || part >= n_parts) return fail(MRG_ERR_ARG, "mrg_fastq_load_part: part %d of %d", part, n_parts);
  try {
    auto h = std::make_unique<mrg_fastq>();
    mrg::load_fastq(path, qual_cutoff, min_len, adapter, threads, h->d, part, n_parts);
    *out = h.release();
    return MRG_OK;
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_fastq_load_part: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_fastq_load_part: %s", e.what());
  }
}

int mrg_fastq_get_info(const mrg_fastq* fq, mrg_fastq_info* info) {
  if (!fq || !info) return fail(MRG_ERR_ARG, "mrg_fastq_get_info: null argument");
  info->n_total = fq->d.n_total;
  info->n_kept = fq->d.n_kept;
  info->phred = fq->d.phred;
  info->words_per_read = fq->d.words_per_read;
  info->max_len = fq->d.max_len;
  info->has_n = fq->d.has_n ? 1 : 0;
  info->n_long = fq->d.long_reads.size();
  return MRG_OK;
}

int mrg_fastq_long_read(const mrg_fastq* fq, uint64_t i, const char** seq) {
  if (!fq || !seq) return fail(MRG_ERR_ARG, "mrg_fastq_long_read: null argument");
  if (i >= fq->d.long_reads.size()) return fail(MRG_ERR_ARG, "mrg_fastq_long_read: index out of range");
  *seq = fq->d.long_reads[i].c_str();
  return MRG_OK;
}

int mrg_fastq_copy(const mrg_fastq* fq, uint32_t words_per_read, uint64_t* words, uint8_t* lens,
                   uint64_t* nmask) {
  if (!fq || (fq->d.n_kept && (!words || !lens))) return fail(MRG_ERR_ARG, "mrg_fastq_copy: null argument");
  const mrg::FastqData& d = fq->d;
  if (words_per_read < d.words_per_read || words_per_read > MRG_MAX_WORDS)
    return fail(MRG_ERR_ARG, "mrg_fastq_copy: words_per_read %u, file needs %u", words_per_read, d.words_per_read);
  if (d.has_n && !nmask) return fail(MRG_ERR_ARG, "mrg_fastq_copy: the file has N bases, nmask is required");
  const size_t n = d.n_kept;
  for (uint32_t w = 0; w < words_per_read; ++w) {
    if (w < d.words_per_read) {
      std::memcpy(words + (size_t)w * n, d.words.data() + (size_t)w * n, n * 8);
      if (nmask) std::memcpy(nmask + (size_t)w * n, d.nmask.data() + (size_t)w * n, n * 8);
    } else {
      std::memset(words + (size_t)w * n, 0, n * 8);
      if (nmask) std::memset(nmask + (size_t)w * n, 0, n * 8);
    }
  }
  if (n) std::memcpy(lens, d.lens.data(), n);
  return MRG_OK;
}

void mrg_fastq_free(mrg_fastq* fq) { delete fq; }

int mrg_gz_open(const char* path, int32_t threads, mrg_gz** out) {
  if (!path || !out) return fail(MRG_ERR_ARG, "mrg_gz_open: null argument");
  try {
    std::unique_ptr<mrg_gz> h(new mrg_gz());
    h->rd.reset(new mrg::GzipReader(path, threads));
    *out = h.release();
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_gz_open: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_gz_open: %s", e.what());
  }
  return MRG_OK;
}

int mrg_gz_read(mrg_gz* gz, void* buf, uint64_t len, uint64_t* got) {
  if (!gz || !got || (len && !buf)) return fail(MRG_ERR_ARG, "mrg_gz_read: null argument");
  try {
    *got = gz->rd->read(static_cast<char*>(buf), (size_t)len);
  } catch (const std::bad_alloc&) {
    return fail(MRG_ERR_NOMEM, "mrg_gz_read: out of memory");
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_gz_read: %s", e.what());
  }
  return MRG_OK;
}

int mrg_gz_info(const mrg_gz* gz, int32_t* parallel, uint64_t* merged) {
  if (!gz) return fail(MRG_ERR_ARG, "mrg_gz_info: null argument");
  if (parallel) *parallel = gz->rd->parallel() ? 1 : 0;
  if (merged) *merged = gz->rd->chunks_merged();
  return MRG_OK;
}

void mrg_gz_close(mrg_gz* gz) { delete gz; }

int mrg_fastq_block_cut(const char* buf, uint64_t len, int32_t at_eof, uint64_t* cut) {
  if (!buf || !cut) return fail(MRG_ERR_ARG, "mrg_fastq_block_cut: null argument");
  *cut = mrg::fastq_block_cut(buf, (size_t)len, at_eof != 0);
  return MRG_OK;
}

int mrg_fastq_parse_device(mrg_ctx* ctx, const char* d_text, uint64_t n_bytes, int32_t phred, int32_t qual_cutoff, int32_t min_len,
                           int32_t cut, uint32_t words_per_read, uint64_t cap, uint64_t* d_words, uint8_t* d_lens, uint64_t* d_nmask,
                           mrg_fastq_device_info* info, void* stream) {
  return mrg_fastq_parse_device_ad(ctx, d_text, n_bytes, phred, qual_cutoff, min_len, cut, nullptr, words_per_read, cap, d_words, d_lens,
                                   d_nmask, info, stream);
}

int mrg_fastq_parse_device_ad(mrg_ctx* ctx, const char* d_text, uint64_t n_bytes, int32_t phred, int32_t qual_cutoff, int32_t min_len,
                              int32_t cut, const char* adapters, uint32_t words_per_read, uint64_t cap, uint64_t* d_words,
                              uint8_t* d_lens, uint64_t* d_nmask, mrg_fastq_device_info* info, void* stream) {
  if (!ctx || !info) return fail(MRG_ERR_ARG, "mrg_fastq_parse_device: null argument");
  mrg::AdapterSet ads;
  std::memset(&ads, 0, sizeof ads);
  if (adapters && *adapters) {
    mrg::TrimSpec spec;
    try {
      spec = mrg::parse_trim_spec(adapters);
    } catch (const std::exception& e) {
      return fail(MRG_ERR_ARG, "mrg_fastq_parse_device_ad: %s", e.what());
    }
    if (spec.cut) return fail(MRG_ERR_ARG, "mrg_fastq_parse_device_ad: `+N` goes in `cut`, adapters takes sequences");
    if (spec.adapters.size() > mrg::kAdapterMax)
      return fail(MRG_ERR_ARG, "mrg_fastq_parse_device_ad: at most %u adapter sequences", mrg::kAdapterMax);
    for (const std::string& a : spec.adapters) {
      if (a.size() > mrg::kAdapterMaxLen)
        return fail(MRG_ERR_ARG, "mrg_fastq_parse_device_ad: adapters of at most %u bases", mrg::kAdapterMaxLen);
      const uint32_t q = ads.n++;
      ads.len[q] = (uint8_t)a.size();
      ads.k[q] = (uint8_t)(int)(0.12 * (double)a.size());
      std::memcpy(ads.seq[q], a.data(), a.size());
    }
  }
  if (n_bytes && (!d_text || !d_words || !d_lens)) return fail(MRG_ERR_ARG, "mrg_fastq_parse_device: null buffers");
  if (words_per_read != 1 && words_per_read != 2 && words_per_read != 4 && words_per_read != 8)
    return fail(MRG_ERR_ARG, "mrg_fastq_parse_device: words_per_read must be 1, 2, 4 or 8");
  if (phred != 33 && phred != 64) return fail(MRG_ERR_ARG, "mrg_fastq_parse_device: phred must be 33 or 64");
  if (n_bytes >= 0x7fffffffull) return fail(MRG_ERR_ARG, "mrg_fastq_parse_device: at most 2^31 - 2 bytes of text per call");
  HIP_TRY(hipSetDevice(ctx->device));
  uint64_t h[7];
  HIP_TRY(mrg::fastq_parse_device(d_text, n_bytes, phred, qual_cutoff, min_len, cut, ads, words_per_read, cap, d_words, d_lens, d_nmask, h,
                                  (hipStream_t)stream));
  info->n_records = h[0];
  info->n_kept = h[1];
  info->n_long = h[2];
  info->max_len = (uint32_t)h[3];
  info->has_n = (int32_t)h[4];
  info->status = (int32_t)h[5];
  info->bad_record = h[6];
  return MRG_OK;
}

int mrg_expand_compact(mrg_ctx* ctx, const uint64_t* d_bits, uint64_t n_words, const uint32_t* runs, uint32_t n_runs,
                       const uint8_t* d_quant8, const uint32_t* d_esc, uint64_t n_esc, uint64_t n, uint32_t n_samples, uint64_t* d_reads,
                       uint8_t* d_lens, uint32_t* d_quant, void* stream) {
  if (!ctx) return fail(MRG_ERR_ARG, "mrg_expand_compact: null argument");
  if (n && (!d_bits || !runs || !d_reads || !d_lens)) return fail(MRG_ERR_ARG, "mrg_expand_compact: null buffers");
  if (d_quant8 && (!d_quant || !n_samples || (n_esc && !d_esc))) return fail(MRG_ERR_ARG, "mrg_expand_compact: null count buffers");
  if (n >= 0xfffffff0ull || n * (uint64_t)(n_samples ? n_samples : 1u) >= 0xfffffff0ull)
    return fail(MRG_ERR_ARG, "mrg_expand_compact: at most 2^32-16 reads (and counts) per call");
  if (n_runs > mrg::kCompactMaxRuns || (n && !n_runs))
    return fail(MRG_ERR_ARG, "mrg_expand_compact: 1..%u length runs (got %u)", mrg::kCompactMaxRuns, n_runs);
  if (((uintptr_t)d_bits % 8) || ((uintptr_t)d_quant8 % 4) || ((uintptr_t)d_quant % 16))
    return fail(MRG_ERR_ARG, "mrg_expand_compact: buffers must be aligned (bits 8, quant8 4, quant 16 bytes)");
  mrg::CompactRuns cr;
  std::memset(&cr, 0, sizeof cr);
  cr.n = n_runs;
  uint64_t end = 0, base = 0;
  for (uint32_t r = 0; r < n_runs; ++r) {
    const uint32_t len = runs[2 * r], count = runs[2 * r + 1];
    if (len > 32u) return fail(MRG_ERR_ARG, "mrg_expand_compact: run %u has reads of %u nt (one-word reads only)", r, len);
    end += count;
    if (end > n) break;
    cr.end[r] = (uint32_t)end;
    cr.base[r] = (uint32_t)base;
    cr.len[r] = (uint8_t)len;
    base += ((uint64_t)count * 2u * len + 63u) / 64u;  // every run starts a word
  }
  if (end != n) return fail(MRG_ERR_ARG, "mrg_expand_compact: the runs count %llu reads, n is %llu", (unsigned long long)end, (unsigned long long)n);
  // (the kernel may read one word past a run's last read)
  if (base + 1 > n_words || base >= 0xffffffffull)
    return fail(MRG_ERR_ARG, "mrg_expand_compact: the runs need %llu + 1 words of bit stream, %llu were given", (unsigned long long)base,
                (unsigned long long)n_words);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(mrg::expand_compact(d_bits, cr, d_quant8, d_esc, n_esc, n, n_samples, d_reads, d_lens, d_quant, (hipStream_t)stream));
  return MRG_OK;
}

int mrg_collapse_run(mrg_ctx* ctx, const uint64_t* d_reads, uint32_t words_per_read, const uint8_t* d_lens,
                     const uint64_t* d_nmask, const uint16_t* d_sample, uint64_t n, uint32_t n_samples,
                     uint32_t max_len, uint64_t cap, uint64_t* d_u_reads, uint8_t* d_u_lens,
                     uint64_t* d_u_nmask, uint32_t* d_quant, uint64_t* d_len_hist, uint64_t* n_unique,
                     void* stream) {
  if (!ctx || !n_unique || !d_len_hist) return fail(MRG_ERR_ARG, "mrg_collapse_run: null argument");
  if (n && (!d_reads || !d_lens || !d_u_reads || !d_u_lens || !d_quant))
    return fail(MRG_ERR_ARG, "mrg_collapse_run: null buffers");
  if (n_samples == 0 || n_samples > 65535) return fail(MRG_ERR_ARG, "mrg_collapse_run: n_samples out of range");
  if (n_samples > 1 && n && !d_sample) return fail(MRG_ERR_ARG, "mrg_collapse_run: d_sample required for >1 samples");
  if (words_per_read == 0 || words_per_read > MRG_MAX_WORDS)
    return fail(MRG_ERR_ARG, "mrg_collapse_run: bad words_per_read");
  if (n >= 0x7fffffffull) return fail(MRG_ERR_ARG, "mrg_collapse_run: at most 2^31-2 reads per call");
  if (d_nmask && !d_u_nmask) return fail(MRG_ERR_ARG, "mrg_collapse_run: d_u_nmask required with d_nmask");
  HIP_TRY(hipSetDevice(ctx->device));
  uint32_t nu = 0;
  // temporaries of the keys-only path (3 x 8 B per read + the sort's own) out of the context's
  // scratch, grown on demand and kept: per-call hipMalloc / hipFree of gigabytes costs milliseconds
  const uint64_t want = n * 40ull + (64ull << 20);
  if (ctx->scratch_bytes < want) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // (nothing queued may still use the old arena)
    (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    if (hipMalloc(&ctx->scratch, want) == hipSuccess) ctx->scratch_bytes = want;
    else (void)hipGetLastError();  // (the collapse then allocates per call)
  }
  hipError_t e = mrg::collapse_reads(d_reads, words_per_read, d_lens, d_nmask, d_sample, (uint32_t)n, n_samples,
                                     max_len, cap, d_u_reads, d_u_lens, d_u_nmask, d_quant, d_len_hist, &nu,
                                     (hipStream_t)stream, ctx->scratch, ctx->scratch_bytes, ctx->n_cu, ctx->collapse_fast != 0);
  if (e == hipErrorInvalidValue)
    return fail(MRG_ERR_ARG, "mrg_collapse_run: cap %llu is smaller than the number of unique reads",
                (unsigned long long)cap);
  if (e == hipErrorInvalidDevicePointer)
    return fail(MRG_ERR_ARG, "mrg_collapse_run: a sample id is not below n_samples (%u)", n_samples);
  if (e != hipSuccess) return fail(MRG_ERR_HIP, "mrg_collapse_run: %s", hipGetErrorString(e));
  *n_unique = nu;
  return MRG_OK;
}

// ------------------------------------------------------------- table writers
int mrg_write_read_table(const char* path, int32_t mapped, const char* header, int32_t append, const uint64_t* reads,
                         uint32_t words_per_read, uint64_t stride, const uint8_t* lens, const uint64_t* nmask,
                         uint64_t n, const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant,
                         uint32_t n_samples, uint32_t n_slots, const char* const* names, const uint64_t* names_off,
                         uint64_t* rows) {
  if (!path || (n && (!reads || !lens || !pass_id || !ref_id || (n_samples && !quant))))
    return fail(MRG_ERR_ARG, "mrg_write_read_table: null argument");
  if (mapped && (!names || !names_off)) return fail(MRG_ERR_ARG, "mrg_write_read_table: mapped rows need the entry names");
  if (words_per_read == 0 || words_per_read > MRG_MAX_WORDS || stride < n || n_slots > MRG_MAX_PASSES)
    return fail(MRG_ERR_ARG, "mrg_write_read_table: bad words_per_read / stride / n_slots");
  try {
    const uint64_t k = mrg::write_read_table(path, mapped != 0, header, append != 0, reads, words_per_read, stride, lens,
                                             nmask, n, pass_id, ref_id, quant, n_samples, n_slots, names, names_off);
    if (rows) *rows = k;
    return MRG_OK;
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_write_read_table: %s", e.what());
  }
}

int mrg_write_isomir_tables(const char* isomirs_path, const char* samples_path, const char* header1, const char* header2,
                            const uint64_t* reads, uint32_t words_per_read, uint64_t stride, const uint8_t* lens,
                            const uint64_t* nmask, uint64_t n, const int8_t* pass_id, const int32_t* ref_id, const uint32_t* quant,
                            uint32_t n_samples, int32_t canon_pass, int32_t isomir_pass, const int32_t* group_of_entry,
                            uint64_t n_entries, const char* const* group_names, uint32_t n_groups, const double* filtered,
                            uint64_t* rows) {
  if (!isomirs_path || !samples_path || !header1 || !header2 || !filtered || !n_samples ||
      (n && (!reads || !lens || !pass_id || !ref_id || !quant || !group_of_entry || !group_names)))
    return fail(MRG_ERR_ARG, "mrg_write_isomir_tables: null argument");
  if (words_per_read == 0 || words_per_read > MRG_MAX_WORDS || stride < n)
    return fail(MRG_ERR_ARG, "mrg_write_isomir_tables: bad words_per_read / stride");
  try {
    const uint64_t k = mrg::write_isomir_tables(isomirs_path, samples_path, header1, header2, reads, words_per_read, stride, lens, nmask, n,
                                                pass_id, ref_id, quant, n_samples, canon_pass, isomir_pass, group_of_entry, n_entries,
                                                group_names, n_groups, filtered);
    if (rows) *rows = k;
    return MRG_OK;
  } catch (const std::exception& e) {
    return fail(MRG_ERR_IO, "mrg_write_isomir_tables: %s", e.what());
  }
}

// ------------------------------------------------------------- packing
int mrg_pack_reads(const char* const* seqs, uint64_t n, uint32_t words_per_read, uint64_t* reads,
                   uint8_t* lens, uint64_t* nmask, int* has_n) {
  if ((n && (!seqs || !reads || !lens)) || words_per_read == 0 || words_per_read > MRG_MAX_WORDS)
    return fail(MRG_ERR_ARG, "mrg_pack_reads: bad argument");
  int any_n = 0;
  for (uint64_t r = 0; r < n; ++r) {
    const char* s = seqs[r];
    size_t L = std::strlen(s);
    if (L > (size_t)words_per_read * 32)
      return fail(MRG_ERR_ARG, "mrg_pack_reads: read %llu has %zu nt, more than %u words hold",
                  (unsigned long long)r, L, words_per_read);
    lens[r] = (uint8_t)L;
    for (uint32_t w = 0; w < words_per_read; ++w) {
      reads[(size_t)w * n + r] = 0;
      if (nmask) nmask[(size_t)w * n + r] = 0;
    }
    for (size_t i = 0; i < L; ++i) {
      uint64_t code = 0;
      bool isn = false;
      switch (s[i]) {
        case 'A': case 'a': code = 0; break;
        case 'C': case 'c': code = 1; break;
        case 'G': case 'g': code = 2; break;
        case 'T': case 't': code = 3; break;
        default: isn = true; break;
      }
      reads[(size_t)(i >> 5) * n + r] |= code << ((i & 31) * 2);
      if (isn) {
        any_n = 1;
        if (nmask) nmask[(size_t)(i >> 5) * n + r] |= 1ull << ((i & 31) * 2);
      }
    }
  }
  if (has_n) *has_n = any_n;
  return MRG_OK;
}

// the ragged form of mrg_cascade_run_long: read r = words[word_off[r] .. word_off[r + 1]), ceil(len / 32) words
static int pack_ragged(const char* const* seqs, uint64_t n, uint64_t* words, uint64_t* nmask, uint64_t* word_off, uint32_t* lens,
                       int* has_n) {
  int any_n = 0;
  uint64_t at = 0;
  for (uint64_t r = 0; r < n; ++r) {
    const char* s = seqs[r];
    const size_t L = std::strlen(s);
    if (L > 0x7fffffffull) return fail(MRG_ERR_ARG, "mrg_pack_reads_ragged: read %llu is too long", (unsigned long long)r);
    const uint64_t nw = (L + 31) / 32;
    if (word_off) word_off[r] = at;
    if (lens) lens[r] = (uint32_t)L;
    if (words) {
      for (uint64_t w = 0; w < nw; ++w) {
        words[at + w] = 0;
        if (nmask) nmask[at + w] = 0;
      }
      for (size_t i = 0; i < L; ++i) {
        uint64_t code = 0;
        bool isn = false;
        switch (s[i]) {
          case 'A': case 'a': code = 0; break;
          case 'C': case 'c': code = 1; break;
          case 'G': case 'g': code = 2; break;
          case 'T': case 't': code = 3; break;
          default: isn = true; break;
        }
        words[at + (i >> 5)] |= code << ((i & 31) * 2);
        if (isn) {
          any_n = 1;
          if (nmask) nmask[at + (i >> 5)] |= 1ull << ((i & 31) * 2);
        }
      }
    }
    at += nw;
  }
  if (word_off) word_off[n] = at;
  if (has_n) *has_n = any_n;
  return MRG_OK;
}

int mrg_pack_reads_ragged(const char* const* seqs, uint64_t n, uint64_t* words, uint64_t* nmask, uint64_t* word_off,
                          uint32_t* lens, int* has_n) {
  if (n && !seqs) return fail(MRG_ERR_ARG, "mrg_pack_reads_ragged: null argument");
  if (!word_off) return fail(MRG_ERR_ARG, "mrg_pack_reads_ragged: word_off is required");
  return pack_ragged(seqs, n, words, nmask, word_off, lens, has_n);
}

int mrg_fastq_copy_long(const mrg_fastq* fq, uint64_t* words, uint64_t* nmask, uint64_t* word_off, uint32_t* lens, int* has_n) {
  if (!fq || !word_off) return fail(MRG_ERR_ARG, "mrg_fastq_copy_long: null argument");
  const auto& lr = fq->d.long_reads;
  std::vector<const char*> ptrs(lr.size());
  for (size_t i = 0; i < lr.size(); ++i) ptrs[i] = lr[i].c_str();
  return pack_ragged(ptrs.data(), lr.size(), words, nmask, word_off, lens, has_n);
}

}  // extern "C"
