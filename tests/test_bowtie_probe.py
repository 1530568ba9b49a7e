"""oracle/bowtie_probe.py: the reference-shaped cascade bench.py runs when a real bowtie 1 is on the
box.  No bowtie exists in the build image, so the plumbing (argv of RAP:577-599/:688, FASTA / SAM /
log files, survivor selection, poly-T fan-out) is driven here with stand-in `bowtie` /
`bowtie-build` executables that answer with the oracle's exhaustive-scan model."""
import os
import stat
import sys
import textwrap

import numpy as np

from oracle import bowtie_probe, cascade, model
from tests.util import LIB_ORDER, World

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STAND_IN = textwrap.dedent('''\
    #!%(python)s
    import os, sys
    sys.path.insert(0, %(root)r)
    from oracle import model
    argv = sys.argv[1:]
    if os.path.basename(sys.argv[0]) == "bowtie-build":
        pos = [a for a in argv if not a.startswith("-")]
        assert os.path.exists(pos[1] + ".fa")
        # a real `.1.ebwt` through the test suite's stand-alone writer (tests/helpers/ebwt_writer.cpp)
        import ctypes as C
        lib = model.Library.from_fasta(pos[1] + ".fa")
        w = C.CDLL(%(writer)r)
        n = len(lib.names)
        a = (C.c_char_p * n)(*[x.encode() for x in lib.names])
        b = (C.c_char_p * n)(*[x.encode() for x in lib.seqs])
        w.ebwt_write.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
        sys.exit(w.ebwt_write(pos[1].encode(), a, b, n, 10, 6, 1))
    mode, mm, t5, t3, pos, i = "n", 2, 0, 0, [], 0
    while i < len(argv):
        a = argv[i]
        if a in ("--threads", "-n", "-v", "-5", "-3"):
            v = int(argv[i + 1]); i += 2
            if a == "-n": mode, mm = "n", v
            elif a == "-v": mode, mm = "v", v
            elif a == "-5": t5 = v
            elif a == "-3": t3 = v
            continue
        if not a.startswith("-"): pos.append(a)
        i += 1
    assert "--norc" in argv and "-f" in argv and "-S" in argv
    lib = model.Library.from_fasta(pos[0] + ".fa")
    reads = [l.strip() for l in open(pos[1]) if l.strip() and l[0] != ">"]
    seed_len, mm_seed, mm_total = (28, mm, 2) if mode == "n" else (1 << 20, mm, mm)
    trimmed = [r[t5:len(r) - t3] if t3 else r[t5:] for r in reads]
    ref, p0, nm = model.align_batch(lib, trimmed, seed_len, mm_seed, mm_total)
    aligned = 0
    for n in lib.names:
        sys.stdout.write("@SQ\\tSN:%%s\\tLN:1\\n" %% n)
    for k, r in enumerate(reads):
        if ref[k] < 0:
            sys.stdout.write("%%s\\t4\\t*\\t0\\t0\\t*\\t*\\t0\\t0\\t%%s\\tI\\tXM:i:0\\n" %% (r, trimmed[k]))
            continue
        aligned += 1
        sys.stdout.write("%%s\\t0\\t%%s\\t%%d\\t255\\t%%dM\\t*\\t0\\t0\\t%%s\\tI\\tNM:i:%%d\\n" %%
                         (r, lib.names[ref[k]], p0[k] + 1, len(trimmed[k]), trimmed[k], nm[k]))
    sys.stderr.write("# reads processed: %%d\\n# reads with at least one reported alignment: %%d (0.00%%%%)\\n" %%
                     (len(reads), aligned))
    ''')


def _install(bindir):
    os.makedirs(bindir, exist_ok=True)
    for name in ("bowtie", "bowtie-build"):
        path = os.path.join(bindir, name)
        with open(path, "w") as fh:
            fh.write(STAND_IN % dict(python=sys.executable, root=ROOT, writer=os.path.join(ROOT, "tests", "_build", "libebwt_writer.so")))
        os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR)


def test_probe_reports_absence(tmp_path, monkeypatch):
    monkeypatch.setenv("PATH", str(tmp_path))
    assert bowtie_probe.find_bowtie() is None


def test_reference_cascade_through_stand_in_bowtie(tmp_path, oracle_lib, native_lib, ebwt_writer):
    bindir = str(tmp_path / "bin")
    _install(bindir)
    found = bowtie_probe.find_bowtie(extra_dirs=[bindir])
    assert found is not None
    w = World(scale=0.02, n_fixed=1500, n_var=400, with_n=False)
    from mirge_amd.index import FmIndex

    def read_ebwt(prefix):
        ix = FmIndex.from_ebwt(prefix)
        return ix.names, [ix.sequence(i) for i in range(ix.n_ref)]
    ref = bowtie_probe.reference_cascade(found[0], found[1], w.libs.libs, w.reads, threads=2,
                                         workdir=str(tmp_path / "work"), ebwt_reader=read_ebwt)
    # every index the (stand-in) bowtie-build wrote went through the product's `.1.ebwt` reader: the hook that
    # pins the reader the moment a real bowtie-build exists on a box
    assert sorted(ref["ebwt"]) == sorted(set(k for k, _, _ in bowtie_probe.PASSES))
    assert all(v["ok"] for v in ref["ebwt"].values()), ref["ebwt"]
    wrong = bowtie_probe.check_ebwt_reader("x", ["a", "b"], ["ACGT", "GGCC"], lambda p: (["a", "b"], ["ACGT", "GGCA"]))
    assert not wrong["ok"] and "entry 1" in wrong["detail"]
    # the oracle's own dict-shaped cascade on the same reads
    libs = {k: model.Library(*w.libs.libs[k]) for k in LIB_ORDER}
    seq_dic = {r: cascade.new_seq_record(r, 1) for r in w.reads}
    log_dic = {"quantStats": [{}], "annotStats": []}
    align = {}
    cascade.run_annotation_pipeline(seq_dic, libs, log_dic, align_dic=align)
    pass_id = np.array([align[r][0] if r in align else -1 for r in w.reads], dtype=np.int8)
    names = [libs[cascade.PASS_TABLE[align[r][0]][0]].names[align[r][1]] if r in align else "" for r in w.reads]
    pos = np.array([align[r][2] if r in align else -1 for r in w.reads], dtype=np.int32)
    assert bowtie_probe.compare(ref, pass_id, names, pos) == dict(D1=0, D2=0, D3=0)
    assert [tuple(s) for s in ref["stats"]] == [(s["readsProcessed"], s["readsAligned"])
                                                for s in log_dic["annotStats"]]
    assert (ref["pass_id"] == 3).sum() > 0 and (ref["pass_id"] == 8).sum() > 0 and (ref["pass_id"] == 1).sum() > 0
