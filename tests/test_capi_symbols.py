"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/mirge_amd.h declares; host-only entry points behave; compute entry
points refuse to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mirge_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrg_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(native_lib):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(native_lib, n), "libmirge_amd.so lacks %s" % n
    from mirge_amd import _native
    assert sorted(_native.SIGNATURES) == names, "ctypes table and header disagree"


def test_error_convention(native_lib):
    rc = native_lib.mrg_index_load(b"/nonexistent/x.mrgfm", C.byref(C.c_void_p()))
    assert rc < 0
    assert b"cannot open" in native_lib.mrg_last_error()
    rc = native_lib.mrg_index_build_fasta(b"/nonexistent.fa", C.byref(C.c_void_p()))
    assert rc < 0


def test_null_and_invalid_arguments_are_errors_not_crashes(native_lib):
    """Every entry point checks its pointers (host-only calls; nothing here needs a GPU)."""
    L = native_lib
    out6 = (C.c_int32 * 6)()
    assert L.mrg_adapter_locate(None, b"ACGT", 0.12, 3, out6) < 0
    assert L.mrg_adapter_locate(b"ACGT", None, 0.12, 3, out6) < 0
    assert L.mrg_adapter_locate(b"ACGT", b"ACGT", 0.12, 3, None) < 0
    assert b"null" in L.mrg_last_error()
    assert L.mrg_fastq_load(None, 10, 16, b"none", 0, C.byref(C.c_void_p())) < 0
    assert L.mrg_fastq_load(b"/nonexistent.fastq", 10, 16, b"none", 0, C.byref(C.c_void_p())) < 0
    assert L.mrg_fastq_load(b"/nonexistent.fastq", 10, 16, b"+x", 0, C.byref(C.c_void_p())) < 0
    assert L.mrg_index_build(None, None, 1, C.byref(C.c_void_p())) < 0
    assert L.mrg_index_get_info(None, None) < 0
    assert L.mrg_ctx_add_library(None, None, None) < 0
    assert L.mrg_count_best(None, None, 1, None, None, 0, 0, 28, 1, 2, None, None, None) < 0
    assert L.mrg_list_best_count(None, None, 1, None, None, 0, 0, 28, 1, 2, None, None, None, None) < 0
    assert L.mrg_list_best_fill(None, None, 1, None, None, 0, 0, 28, 1, 2, None, None, 0, None, None, None) < 0
    assert L.mrg_cascade_run(None, None, 1, None, None, 0, None, 0, None, None, None, None, None, None, 0, None) < 0
    assert L.mrg_tally_run(None, None, None, None, 0, 1, 1, 9, 0, 8, None, None) < 0
    bytes_ = C.c_uint64()
    assert L.mrg_cascade_workspace_bytes(1000, C.byref(bytes_)) == 0 and bytes_.value > 4000
    assert L.mrg_cascade_workspace_bytes(1000, None) < 0


def test_index_file_round_trip_and_rejects_garbage(native_lib, tmp_path):
    """save / load (jump tables are rebuilt, not stored); a foreign or truncated file is refused."""
    import numpy as np
    from mirge_amd._native import MirgeAmdError
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(4)
    names = ["e%d" % i for i in range(20)]
    seqs = ["".join("ACGTN"[c] for c in rng.choice(5, int(L), p=[.24, .24, .24, .24, .04])) for L in rng.integers(30, 300, 20)]
    ix = FmIndex.build(names, seqs)
    p = str(tmp_path / "lib.mrgfm")
    ix.save(p)
    jx = FmIndex.load(p)
    assert jx.names == names and [jx.sequence(i) for i in range(20)] == seqs
    a, b = ix.view(), jx.view()
    for k in ("blocks", "super", "text", "sa", "ftab", "seg_start", "seg_ref", "seg_off", "chunk_seg"):
        assert np.array_equal(a[k], b[k]), k
    assert a["ftab_ks"] == b["ftab_ks"]
    bad = tmp_path / "bad.mrgfm"
    bad.write_bytes(b"MRGFM4\0\0" + b"\0" * 64)          # an older format
    with pytest.raises(MirgeAmdError):
        FmIndex.load(str(bad))
    data = open(p, "rb").read()
    bad.write_bytes(data[:len(data) // 2])
    with pytest.raises(MirgeAmdError):
        FmIndex.load(str(bad))


def test_pack_reads_c_helper_matches_numpy(native_lib):
    import numpy as np
    from mirge_amd import pack
    seqs = ["ACGT", "TTTTNACG", "A" * 40, ""]
    W = 2
    n = len(seqs)
    arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
    reads = np.zeros((W, n), dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint8)
    nm = np.zeros((W, n), dtype=np.uint64)
    has_n = C.c_int(0)
    assert native_lib.mrg_pack_reads(arr, n, W, reads.ctypes.data, lens.ctypes.data, nm.ctypes.data,
                                     C.byref(has_n)) == 0
    w2, l2, nm2 = pack.pack_reads(seqs, W)
    assert np.array_equal(reads, w2) and np.array_equal(lens, l2) and np.array_equal(nm, nm2)
    assert has_n.value == 1
    assert pack.unpack_reads(reads, lens, nm) == seqs


def test_no_cpu_fallback_without_gpu(native_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mirge_amd import _native
    from mirge_amd.engine import Engine
    with pytest.raises(_native.MirgeAmdError) as ei:
        Engine(0)
    assert ei.value.code == _native.MRG_ERR_NO_DEVICE
