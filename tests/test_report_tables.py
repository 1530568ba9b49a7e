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
