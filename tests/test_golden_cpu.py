"""The oracle's bookkeeping restatement and the product's host-side functions
against vectors captured from the reference's own Python
(tests/golden/make_golden.py -> tests/golden/cascade_small.json)."""
import copy
import json
import os

import pytest

from oracle import cascade, model
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "cascade_small.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def oracle_run(golden, oracle_lib):
    libs = {k: model.Library(*v) for k, v in golden["libraries"].items()}
    seq_dic, len_dic = cascade.collapse(golden["samples"])
    log_dic = {"quantStats": [{} for _ in golden["sample_list"]], "annotStats": []}
    cascade.run_annotation_pipeline(seq_dic, libs, log_dic)
    return libs, seq_dic, len_dic, log_dic


def test_collapse_matches_quantReads(golden, oracle_run):
    _, seq_dic, len_dic, _ = oracle_run
    exp = golden["expected"]
    assert set(seq_dic) == set(exp["seqDic"])
    for s, rec in seq_dic.items():
        assert rec["quant"] == exp["seqDic"][s]["quant"]
        assert rec["length"] == exp["seqDic"][s]["length"]
    assert {str(k): v for k, v in len_dic.items()} == exp["readLengthDic"]


def test_cascade_annot_and_counters(golden, oracle_run):
    _, seq_dic, _, log_dic = oracle_run
    exp = golden["expected"]
    for s, rec in seq_dic.items():
        assert rec["annot"] == exp["seqDic"][s]["annot"], s
    assert log_dic["annotStats"] == exp["annotStats"]
    # the fixture exercises every pass, the >25-nt hairpin branch and the poly-T fan-out
    assert all(a["readsAligned"] > 0 for a in exp["annotStats"])


def test_reads_of_every_length_are_offered_to_the_passes(oracle_lib):
    """tests/golden/long_reads.json: reads of 33..300 nt through the reference's collapse + cascade (RAP:543-554
    caps only the first pass at a length): the oracle's restatement gives the same annot rows and per-pass
    counters -- for all reads, and for the run without the reads beyond 255 nt (what the product aligns)."""
    with open(os.path.join(ROOT, "tests", "golden", "long_reads.json")) as fh:
        g = json.load(fh)
    exp = g["expected"]
    libs = {k: model.Library(*v) for k, v in g["libraries"].items()}
    seq_dic, len_dic = cascade.collapse(g["samples"])
    assert {str(k): v for k, v in len_dic.items()} == exp["readLengthDic"]
    log_dic = {"quantStats": [{}], "annotStats": []}
    cascade.run_annotation_pipeline(seq_dic, libs, log_dic)
    assert set(seq_dic) == set(exp["seqDic"])
    for s, rec in seq_dic.items():
        assert rec["annot"] == exp["seqDic"][s]["annot"] and rec["quant"] == exp["seqDic"][s]["quant"], s
    assert log_dic["annotStats"] == exp["annotStats"]
    by_len = lambda lo, hi: sum(1 for s, r in exp["seqDic"].items() if lo < len(s) <= hi and r["annot"][0])
    assert min(by_len(32, 64), by_len(64, 128), by_len(128, 255), by_len(255, 999)) > 20
    seq_le = {s: cascade.new_seq_record(s, r["quant"][0]) for s, r in exp["seqDic"].items() if len(s) <= 255}
    log_le = {"quantStats": [{}], "annotStats": []}
    cascade.run_annotation_pipeline(seq_le, libs, log_le)
    assert log_le["annotStats"] == exp["annotStats_le255"] != exp["annotStats"]


def test_summarize_merge_filter(golden, oracle_run):
    libs, seq_dic, _, log_dic = oracle_run
    exp = golden["expected"]
    log_dic = copy.deepcopy(log_dic)
    mir_dic = {}
    cascade.summarize(seq_dic, golden["sample_list"], log_dic, mir_dic, libs["mirna"].names)
    assert mir_dic == exp["mirDic_after_summarize"]
    assert log_dic["quantStats"] == exp["quantStats_after_summarize"]
    cascade.mirna_merge(golden["merges"], golden["sample_list"], mir_dic)
    assert mir_dic == exp["mirDic_after_merge"]
    assert any("/" in k for k in mir_dic)  # merged families exist in the fixture
    cascade.filter_mirnas(mir_dic, golden["sample_list"], log_dic, golden["cano_ratio"])
    assert mir_dic == exp["mirDic_after_filter"]
    assert log_dic["quantStats"] == exp["quantStats_after_filter"]


def test_product_host_functions_merge_filter_collapse(golden, tmp_path):
    """mirge_amd.annotate's host-side functions (no GPU involved)."""
    from mirge_amd import annotate
    exp = golden["expected"]
    names, seqs = golden["libraries"]["mirna"]
    fa = tmp_path / "mirna.fa"
    mature = {n: s[2:-6] for n, s in zip(names, seqs)}
    # as SynthLibraries.write_layout: the SNP_pseudo FASTA lists merged names too (W2C:1295)
    fa.write_text("".join(">%s\n%s\n" % (n, mature[n]) for n in names) +
                  "".join(">%s\n%s\n" % (l.split(",")[0], mature[l.split(",")[1]]) for l in golden["merges"]))
    merges = tmp_path / "merges.csv"
    merges.write_text("".join(l + "\n" for l in golden["merges"]))
    mir_dic = copy.deepcopy(exp["mirDic_after_summarize"])
    name_seq = {}
    annotate.miRNAmerge(str(merges), golden["sample_list"], mir_dic, str(fa), name_seq)
    assert mir_dic == exp["mirDic_after_merge"]
    assert name_seq == exp["mirNameSeqDic"]
    log_dic = {"quantStats": copy.deepcopy(exp["quantStats_after_summarize"])}
    annotate.filter(mir_dic, golden["sample_list"], log_dic, golden["cano_ratio"])
    assert mir_dic == exp["mirDic_after_filter"]
    assert log_dic["quantStats"] == exp["quantStats_after_filter"]
    seq_dic, len_dic = {}, {}
    for si, reads in enumerate(golden["samples"]):
        annotate.quantReads(reads, seq_dic, len_dic, 2, si)
    assert {s: r["quant"] for s, r in seq_dic.items()} == {s: r["quant"] for s, r in exp["seqDic"].items()}


def test_filter_aborts_on_empty_sample():
    from mirge_amd import annotate
    mir = {"a": {"quant": [5, 0], "iscan": [5, 0]}}
    log = {"quantStats": [{}, {}]}
    with pytest.raises(SystemExit) as e:
        annotate.filter(mir, ["s0", "s1"], log, "0.1")
    assert e.value.code == 1
    with pytest.raises(cascade.NoMirnaReads):
        cascade.filter_mirnas({"a": {"quant": [5, 0], "iscan": [5, 0]}}, ["s0", "s1"],
                              {"quantStats": [{}, {}]}, "0.1")
