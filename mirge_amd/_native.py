"""ctypes binding of the C-ABI in include/mirge_amd.h.

The shared library is built in-tree (mirge_amd/lib/libmirge_amd.so) by
`make -C mirge_amd/csrc` / `__graft_entry__.build()`.  There is no Python or
CPU fallback: if the library is missing, loading raises; if no gfx950 device is
usable, mrg_ctx_create fails with MRG_ERR_NO_DEVICE and `check` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmirge_amd.so")

MRG_MAX_PASSES = 16
MRG_MAX_WORDS = 8
MRG_ERR_NO_DEVICE = -3


class MirgeAmdError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mirge_amd C-ABI error %d: %s" % (code, message))
        self.code = code


class PassCfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("lib", "seed_len", "max_mm_seed", "max_mm_total", "trim5", "trim3", "min_len",
                 "max_len", "poly_t", "reserved")]


class PassStats(C.Structure):
    _fields_ = [("processed", C.c_uint64), ("aligned", C.c_uint64), ("steps", C.c_uint64),
                ("candidates", C.c_uint64), ("lookups", C.c_uint64), ("ms", C.c_float),
                ("lds_bytes", C.c_uint32), ("lds_mode", C.c_uint32), ("group", C.c_uint32),
                ("n_launches", C.c_uint32), ("kbits_log2", C.c_uint32),
                ("pair_anchor", C.c_uint32), ("ms_rest", C.c_float), ("variant", C.c_uint32), ("reserved", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("n_ref", C.c_uint32), ("n_seg", C.c_uint32), ("n_bases", C.c_uint32),
                ("n_blocks", C.c_uint32), ("n_super", C.c_uint32), ("primary", C.c_uint32),
                ("text_words", C.c_uint32), ("ftab_ks", C.c_uint8 * 4),
                ("C", C.c_uint32 * 4), ("bytes_fm", C.c_uint64), ("bytes_sa", C.c_uint64)]


class FastqInfo(C.Structure):
    _fields_ = [("n_total", C.c_uint64), ("n_kept", C.c_uint64), ("phred", C.c_int32),
                ("words_per_read", C.c_uint32), ("max_len", C.c_uint32), ("has_n", C.c_int32),
                ("n_long", C.c_uint64)]


class FastqDeviceInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_kept", C.c_uint64), ("n_long", C.c_uint64), ("bad_record", C.c_uint64),
                ("max_len", C.c_uint32), ("has_n", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32)]


class IndexView(C.Structure):
    _fields_ = [("blocks", C.POINTER(C.c_uint32)), ("super", C.POINTER(C.c_uint32)),
                ("text", C.POINTER(C.c_uint32)), ("sa", C.POINTER(C.c_uint64)),
                ("ftab", C.POINTER(C.c_uint32))] + \
               [(k, C.POINTER(C.c_uint32)) for k in ("seg_start", "seg_ref", "seg_off", "chunk_seg", "ctx", "kbits")]


class DictView(C.Structure):
    _fields_ = [("slots", C.POINTER(C.c_uint64)), ("log2_slots", C.c_uint32), ("key_bases", C.c_uint32),
                ("n_keys", C.c_uint64), ("n_overflow", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/mirge_amd.h declares
SIGNATURES = {
    "mrg_version": (C.c_int, []),
    "mrg_last_error": (C.c_char_p, []),
    "mrg_index_build": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32,
                                  C.POINTER(C.c_void_p)]),
    "mrg_index_build_fasta": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mrg_index_build_ebwt": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mrg_index_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mrg_index_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mrg_index_free": (None, [C.c_void_p]),
    "mrg_index_get_info": (C.c_int, [C.c_void_p, C.POINTER(IndexInfo)]),
    "mrg_index_name": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p)]),
    "mrg_index_seq": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32,
                                C.POINTER(C.c_uint32)]),
    "mrg_index_get_view": (C.c_int, [C.c_void_p, C.POINTER(IndexView)]),
    "mrg_index_get_dict": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(DictView)]),
    "mrg_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mrg_ctx_destroy": (None, [C.c_void_p]),
    "mrg_ctx_add_library": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "mrg_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mrg_ctx_device_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                      C.c_char_p, C.c_uint32]),
    "mrg_ctx_library_stats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]),
    "mrg_ctx_library_check_tables": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64)]),
    "mrg_ctx_release_scratch": (C.c_int, [C.c_void_p]),
    "mrg_cascade_workspace_bytes": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint64)]),
    "mrg_cascade_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.POINTER(PassCfg), C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.c_void_p]),
    "mrg_pack_assignments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.c_void_p]),
    "mrg_cascade_run_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(PassCfg),
                                         C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mrg_tally_run_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mrg_edit_tally_run_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                            C.c_void_p]),
    "mrg_cascade_run_long": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(PassCfg),
                                       C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PassStats),
                                       C.c_void_p]),
    "mrg_cascade_run_id": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mrg_cascade_stats": (C.c_int, [C.c_void_p, C.POINTER(PassStats), C.c_uint32]),
    "mrg_tally_counts_len": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.POINTER(C.c_uint64)]),
    "mrg_tally_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p]),
    "mrg_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mrg_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "mrg_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mrg_comm_destroy": (C.c_int, [C.c_void_p]),
    "mrg_edit_counts_len": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "mrg_edit_tally_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mrg_count_best": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "mrg_list_best_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_uint64), C.c_void_p]),
    "mrg_list_best_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mrg_annotate_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_uint64, C.POINTER(PassCfg), C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PassStats),
                                    C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                    C.c_void_p]),
    "mrg_annotate_long_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(PassCfg),
                                         C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PassStats)]),
    "mrg_fastq_load": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32,
                                 C.POINTER(C.c_void_p)]),
    "mrg_fastq_load_part": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_void_p)]),
    "mrg_adapter_locate": (C.c_int, [C.c_char_p, C.c_char_p, C.c_double, C.c_int32, C.POINTER(C.c_int32)]),
    "mrg_fastq_get_info": (C.c_int, [C.c_void_p, C.POINTER(FastqInfo)]),
    "mrg_fastq_long_read": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p)]),
    "mrg_fastq_copy": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mrg_fastq_free": (None, [C.c_void_p]),
    "mrg_fastq_block_cut": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(C.c_uint64)]),
    "mrg_gz_open": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "mrg_gz_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "mrg_gz_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "mrg_gz_close": (None, [C.c_void_p]),
    "mrg_fastq_parse_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                         C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FastqDeviceInfo), C.c_void_p]),
    "mrg_fastq_parse_device_ad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p,
                                            C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FastqDeviceInfo),
                                            C.c_void_p]),
    "mrg_expand_compact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mrg_collapse_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "mrg_write_read_table": (C.c_int, [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_uint32,
                                       C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p), C.c_void_p,
                                       C.POINTER(C.c_uint64)]),
    "mrg_write_isomir_tables": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                          C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                          C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint32,
                                          C.c_void_p, C.POINTER(C.c_uint64)]),
    "mrg_pack_reads": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mrg_pack_reads_ragged": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int)]),
    "mrg_fastq_copy_long": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process: torch wheels bundle their own libamdhip64
    (SONAME libamdhip64.so.7, the name this library needs), and the streams and
    device pointers the host passes in come from torch.  Loading torch's copy
    first makes the dynamic loader bind libmirge_amd.so to it instead of
    opening a second runtime from /opt/rocm, which then finds no GPU."""
    if os.environ.get("MIRGE_AMD_STANDALONE_HIP") == "1":
        return
    try:
        import torch
    except ImportError:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load libmirge_amd.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C mirge_amd/csrc` (hipcc --offload-arch=gfx950). "
            "mirge_amd has no CPU fallback." % LIB_PATH)
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().mrg_last_error()
        raise MirgeAmdError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
