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
