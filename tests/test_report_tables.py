"""mirge_amd.report against the reference's own writers (tables captured by
tests/golden/make_golden.py from writeDataToCSV.py / generateReport.py)."""
import copy
import json
import os

import pytest

from mirge_amd import report
from tests.conftest import ROOT

ANNOT = ["exact miRNA", "hairpin miRNA", "mature tRNA", "primary tRNA", "snoRNA", "rRNA", "ncrna others",
         "mRNA", "isomiR miRNA"]


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "cascade_small.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def written(golden, tmp_path_factory):
    exp = golden["expected"]
    out = tmp_path_factory.mktemp("tables")
    log_dic = {"quantStats": copy.deepcopy(exp["quantStats_after_filter"])}
    for i, q in enumerate(log_dic["quantStats"]):
        q["totalReads"] = exp["totalReads"][i]
        q["trimmedReads"] = exp["trimmedReads"][i]
    report.writeDataToCSV(str(out), ANNOT, golden["sample_list"], True, False, log_dic,
                          exp["seqDic"], exp["mirDic_after_filter"])
    report.write_annotation_report_csv(str(out / "annotation.report.csv"), golden["sample_list"], log_dic)
    return {fn: open(str(out / fn)).read().split("\n") for fn in exp["tables"]}


def cells_equal(a, b):
    """Exact for text/integers; floats may differ only by Python 2's 12-digit str()."""
    if a == b:
        return True
    try:
        fa, fb = float(a), float(b)
    except ValueError:
        return False
    return abs(fa - fb) <= 1e-11 * max(1.0, abs(fa), abs(fb))


def rows_equal(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        gc, wc = g.split(","), w.split(",")
        assert len(gc) == len(wc), (g, w)
        assert all(cells_equal(x, y) for x, y in zip(gc, wc)), (g, w)


@pytest.mark.parametrize("name", ["miR.Counts.csv", "annotation.report.csv"])
def test_integer_tables_are_byte_identical(written, golden, name):
    assert written[name] == golden["expected"]["tables"][name]


@pytest.mark.parametrize("name", ["mapped.csv", "unmapped.csv"])
def test_read_tables_same_rows(written, golden, name):
    got, want = written[name], golden["expected"]["tables"][name]
    assert got[0] == want[0]
    assert sorted(got[1:]) == sorted(want[1:])      # dict order is arbitrary in the reference


def test_rpm_table(written, golden):
    rows_equal(written["miR.RPM.csv"], golden["expected"]["tables"]["miR.RPM.csv"])


def test_isomir_entropy_tables(written, golden):
    for name in ("isomirs.csv", "isomirs.samples.csv"):
        got, want = written[name], golden["expected"]["tables"][name]
        assert got[0] == want[0]
        rows_equal(sorted(got[1:]), sorted(want[1:]))
    assert len(golden["expected"]["tables"]["isomirs.csv"]) > 100


def test_calc_entropy_known_answers(golden):
    for values, want in golden["expected"]["calcEntropy"]:
        assert report.calc_entropy(values) == want


def test_python2_float_formatting():
    # what CPython 2.7 prints for str(x)
    cases = {0.1: "0.1", 1.0 / 3: "0.333333333333", 1000000.0 * 5 / 7: "714285.714286", 1e16: "1e+16",
             123456789012.0: "123456789012.0", 1234567890123.0: "1.23456789012e+12", 0.0: "0.0",
             2.5e-05: "2.5e-05", 100.0: "100.0", 66.66666666666666: "66.6666666667"}
    for x, want in cases.items():
        assert report.py2_float_str(x) == want


def test_read_table_writer_many_blocks_equals_one_thread(native_lib, tmp_path, monkeypatch):
    """mapped.csv / unmapped.csv of 700 000 reads (several blocks of rows per formatting thread, one to four packed
    words, reads with N, three samples): the same bytes whatever the number of formatting threads, and the rows a
    Python loop over the arrays writes."""
    import numpy as np
    from mirge_amd import columnar, pack
    rng = np.random.default_rng(12)
    n = 700_000
    lens = rng.integers(16, 41, n).astype(np.uint8)
    lens[::1000] = 128
    W = 4
    words = rng.integers(0, 1 << 63, (W, n), dtype=np.uint64)
    nmask = np.zeros((W, n), dtype=np.uint64)
    nmask[0, ::777] = 1 << 10
    pass_id = rng.integers(-1, 3, n).astype(np.int8)
    names = [["a%d" % i for i in range(50)], ["b%d" % i for i in range(7)], ["c%d" % i for i in range(300)]]
    ref_id = np.where(pass_id >= 0, rng.integers(0, 7, n), -1).astype(np.int32)
    quant = rng.integers(0, 1000, (n, 3)).astype(np.uint32)
    outs = {}
    for threads in ("1", "5", None):
        if threads is None:
            monkeypatch.delenv("MIRGE_AMD_TABLE_THREADS", raising=False)
        else:
            monkeypatch.setenv("MIRGE_AMD_TABLE_THREADS", threads)
        d = tmp_path / ("t%s" % threads)
        d.mkdir()
        rows = columnar.write_read_tables(str(d), ["A", "B", "C"], ["s0", "s1", "s2"], words, lens, nmask, quant, pass_id, ref_id, names)
        assert rows["mapped.csv"] == int((pass_id >= 0).sum()) and rows["unmapped.csv"] == int((pass_id < 0).sum())
        outs[threads] = {fn: open(str(d / fn), "rb").read() for fn in ("mapped.csv", "unmapped.csv")}
    assert outs["1"] == outs["5"] == outs[None]
    seqs = pack.unpack_reads(words[:, :3000].copy(), lens[:3000], nmask[:, :3000].copy())
    want_m, want_u = [], []
    for i, s in enumerate(seqs):
        p = int(pass_id[i])
        slots = ["", "", ""]
        if p >= 0:
            slots[p] = names[p][int(ref_id[i])]
        row = "%s,%d,%s,%s" % (s, 1 if p >= 0 else 0, ",".join(slots), ",".join(str(int(x)) for x in quant[i]))
        (want_m if p >= 0 else want_u).append(row)
    got_m = outs["1"]["mapped.csv"].decode().split("\n")
    got_u = outs["1"]["unmapped.csv"].decode().split("\n")
    assert got_m[0] == "uniqueSequence,annotFlag,A,B,C,s0,s1,s2" and got_m[1:1 + len(want_m)] == want_m
    assert got_u[1:1 + len(want_u)] == want_u
    # an entry index beyond its pass's names is an error, not a crash or a wrong row
    bad = ref_id.copy()
    bad[np.flatnonzero(pass_id == 1)[-1]] = 7
    from mirge_amd._native import MirgeAmdError
    with pytest.raises(MirgeAmdError, match="out of range"):
        columnar.write_read_tables(str(tmp_path), ["A", "B", "C"], ["s0", "s1", "s2"], words, lens, nmask, quant, pass_id, bad, names)


@pytest.mark.parametrize("n_samples", [1, 3])
def test_native_isomir_tables_equal_the_python_path(native_lib, tmp_path, n_samples):
    """columnar.write_isomir_tables (mrg_write_isomir_tables, round 6) = report.write_isomir_tables over
    isomir_dic(read_subset(...)), byte for byte: groups in order of first appearance, SNP suffixes folded, Python 2's
    str(float), the entropy sums in the reference's order, the isomirs.samples.csv row quirk with several samples."""
    import numpy as np
    from mirge_amd import columnar, pack
    rng = np.random.default_rng(600 + n_samples)
    M = 120
    mirna_names = []
    for i in range(M):
        base = "syn-miR-%d-5p" % (i // 2 if i % 7 == 0 else i)
        mirna_names.append(base + (".SNP%d" % i if i % 5 == 0 else ""))
    n = 6000
    reads = ["".join("ACGTN"[c] for c in rng.choice(5, int(rng.integers(16, 31)), p=[.249, .249, .249, .249, .004])) for _ in range(n)]
    reads = list(dict.fromkeys(reads))
    n = len(reads)
    words, lens, nmask = pack.pack_reads(reads)
    quant = rng.integers(0, 400, (n, n_samples)).astype(np.uint32)
    quant[rng.random((n, n_samples)) < 0.3] = 1          # (entropy skips entries <= 1)
    pass_id = rng.choice(np.array([-1, 0, 3, 8], dtype=np.int8), n, p=[0.2, 0.15, 0.05, 0.6])
    ref_id = rng.integers(0, M, n).astype(np.int32)
    ref_id[rng.random(n) < 0.5] = 7                      # one abundant miRNA
    npp = [mirna_names] + [["x%d" % i for i in range(M)]] * 7 + [mirna_names]
    log_dic = {"quantStats": [{"mirnaReadsFiltered": int(1000 + 977 * i)} for i in range(n_samples)]}
    samples = ["s%d.fastq" % i for i in range(n_samples)]
    sub, _ = columnar.read_subset(words, lens, nmask, quant, pass_id, ref_id, np.zeros(n, np.int32), np.zeros(n, np.uint8), npp, (0, 8))
    py = [str(tmp_path / "py_isomirs.csv"), str(tmp_path / "py_samples.csv")]
    report.write_isomir_tables(py[0], py[1], samples, columnar.isomir_dic(sub, n_samples), log_dic)
    nat = [str(tmp_path / "isomirs.csv"), str(tmp_path / "samples.csv")]
    rows = columnar.write_isomir_tables(nat[0], nat[1], samples, words, lens, nmask, quant, pass_id, ref_id, mirna_names, log_dic)
    for a, b in zip(py, nat):
        assert open(a, "rb").read() == open(b, "rb").read(), b
    assert rows == int((pass_id == 8).sum()) and rows > 1000
