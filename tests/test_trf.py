"""mirge_amd.trf (-trf typing and tables) against the reference's own run (tests/golden/trf.json;
bowtie is a stand-in there and random.choice is pinned to min)."""
import copy
import json
import os
import types

import pytest

from mirge_amd import trf
from tests.conftest import ROOT

FILES = ("tRFs.potential.report.tsv", "tRF.Counts.csv", "tRF.RP100K.csv",
         "discarded.reads.summary.assigningtRFs.csv")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "trf.json")) as fh:
        return json.load(fh)


def write_world(golden, root):
    from mirge_amd import synth
    ns = types.SimpleNamespace(libs={k: tuple(v) for k, v in golden["libraries"].items()}, merges=golden["merges"])
    synth.SynthLibraries.write_layout(ns, root, species="human", db="miRBase")
    for suffix, text in golden["tables"].items():
        with open(os.path.join(root, "human", "annotation.Libs", "human" + suffix), "w") as fh:
            fh.write(text)


def check_files(golden, outdir):
    exp = golden["expected"]["files"]
    for fn in FILES[1:]:
        assert open(os.path.join(str(outdir), fn)).read().split("\n") == exp[fn], fn
    got = open(os.path.join(str(outdir), FILES[0])).read().split("\n")
    want = exp[FILES[0]]
    assert got[0] == want[0] and len(got) == len(want) > 100

    def norm(line):
        f = line.split("\t")
        # hit columns follow dict order (= SAM line order) and, for the de-duplicated ones,
        # list(set(...)) order in the reference (W2C:686-704): compared as multisets
        return f[:4] + [sorted(c.split(",")) for c in f[4:10]] + f[10:]
    assert sorted(map(norm, got[1:-1]), key=repr) == sorted(map(norm, want[1:-1]), key=repr)


def oracle_lister(golden):
    from oracle import model
    libs = {k: model.Library(*golden["libraries"][k]) for k in ("mature_trna", "pre_trna")}

    def lister(reads, key, v):
        out = []
        for r in reads:
            hits, _ = model.align_all_best(libs[key], r, 1 << 20, v, v)
            out.append([(libs[key].names[e], o) for e, o in hits])
        return out
    return lister


def run_tables(golden, content, outdir, root):
    tables = trf.load_trf_tables(root, "human")
    log_dic = {"quantStats": copy.deepcopy(golden["state"]["quantStats"])}
    pre = dict(zip(*golden["libraries"]["pre_trna"]))
    trf.write_trf_tables(str(outdir), golden["sample_list"], log_dic, content, tables, pre)


def test_known_answers(golden, tmp_path):
    write_world(golden, str(tmp_path))
    stru = trf.load_trf_tables(str(tmp_path), "human")["trnaStruDic"]
    name = golden["libraries"]["mature_trna"][0][0]
    for start, ln, want in golden["expected"]["trfTypes"]:
        assert trf.trfTypes("A" * ln, name, start, stru, {}) == want, (start, ln)
    assert {w for _, _, w in golden["expected"]["trfTypes"]} == {"tRF-whole", "5'-half", "5'-tRF", "3'-half",
                                                                  "3'-tRF", "i-tRF"}
    assert trf.trfTypes("ACGT", "pre_x_trailer", 3, stru, {"pre_x_trailer": "ACGT"}) == "tRF-1"
    for a, b, dist, coord in golden["expected"]["distance"]:
        assert trf.get_distance2(a, b) == dist and list(trf.coordinate(b)) == coord
    assert trf.add_dash_new("ACGT", 10, 3, 6) == "--ACGT----"
    assert trf.strip_poly_t("ACGTACGTACGTTT") == "ACGTACGTACG" and trf.strip_poly_t("ACGTACGTACTTT") is None
    assert trf.strip_poly_t("ACGTACGTACGATT") is None


def test_tables_from_reference_content(golden, tmp_path):
    """write_trf_tables on the reference's own trfContentDic."""
    write_world(golden, str(tmp_path / "libs"))
    content = copy.deepcopy(golden["expected"]["trfContentDic_after_cascade"])
    run_tables(golden, content, tmp_path, str(tmp_path / "libs"))
    check_files(golden, tmp_path)
    # like the reference, only the selected tRNA stays on each read
    assert all(len([k for k in rec if k not in ("uid", "RPM", "count")]) == 1 for rec in content.values())


def test_content_from_exhaustive_scan_listings(golden, oracle_lib, tmp_path):
    """collect_trf_content with the oracle's `-a --best --strata` listings reproduces the
    reference's trfContentDic."""
    write_world(golden, str(tmp_path))
    stru = trf.load_trf_tables(str(tmp_path), "human")["trnaStruDic"]
    pre = dict(zip(*golden["libraries"]["pre_trna"]))
    content = {}
    trf.collect_trf_content(content, golden["state"]["seqDic"], golden["sample_list"], stru, pre,
                            oracle_lister(golden))
    assert content == golden["expected"]["trfContentDic_after_cascade"]
    assert sum(len(r) > 3 for r in content.values()) > 20      # reads with several candidate tRNAs


@pytest.mark.gpu
def test_cli_trf_from_fastq_matches_reference_files(golden, native_lib, tmp_path):
    """`annotate -trf` end to end from FASTQ on the GPU -> the reference's tRF tables."""
    from mirge_amd import cli
    root = str(tmp_path / "libs")
    write_world(golden, root)
    fastqs = []
    for name, reads in zip(golden["sample_list"], golden["samples"]):
        p = str(tmp_path / name)
        with open(p, "w") as fh:
            for k, r in enumerate(reads):
                fh.write("@r%d\n%s\n+\n%s\n" % (k, r, "I" * len(r)))
        fastqs.append(p)
    out = cli.annotate_main(cli.build_parser().parse_args(
        ["annotate", "-s"] + fastqs + ["-lib", root, "-sp", "human", "-o", str(tmp_path), "-trf"]), materialize=True)
    st = golden["state"]["seqDic"]
    assert {s: r["annot"] for s, r in out["seqDic"].items()} == {s: r["annot"] for s, r in st.items()}
    check_files(golden, out["outdir"])
    with pytest.raises(SystemExit):
        cli.annotate_main(cli.build_parser().parse_args(
            ["annotate", "-s"] + fastqs + ["-lib", root, "-sp", "mouse", "-o", str(tmp_path), "-trf"]))
