import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """libmirge_amd.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from mirge_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "mirge_amd", "csrc")], check=True)
    return _native.load()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import model
    model.build()
    return model.lib()


@pytest.fixture(scope="session")
def ebwt_writer():
    """tests/helpers/ebwt_writer.cpp (TEST INFRASTRUCTURE, stand-alone g++ build into tests/_build/):
    write_ebwt(prefix, names, seqs, ftab_chars=10, line_rate=6, lines_per_side=1) -> `<prefix>.1.ebwt`."""
    import ctypes as C
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "helpers", "ebwt_writer.cpp")
    out = os.path.join(here, "_build", "libebwt_writer.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, src], check=True)
    lib = C.CDLL(out)
    lib.ebwt_write.restype = C.c_int
    lib.ebwt_write.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32, C.c_int32, C.c_int32, C.c_int32]

    def write_ebwt(prefix, names, seqs, ftab_chars=10, line_rate=6, lines_per_side=1):
        n = len(names)
        a = (C.c_char_p * n)(*[s.encode() for s in names])
        b = (C.c_char_p * n)(*[s.encode() for s in seqs])
        rc = lib.ebwt_write(os.fsencode(prefix), a, b, n, int(ftab_chars), int(line_rate), int(lines_per_side))
        assert rc == 0, "ebwt_write failed (%d)" % rc
    return write_ebwt
