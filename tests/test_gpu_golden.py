"""GPU (-m gpu): the product's reference-shaped host API
(mirge_amd.annotate.runAnnotationPipeline / summarize / miRNAmerge / filter)
reproduces the vectors captured from the reference's own Python."""
import copy
import json
import os

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "cascade_small.json")) as fh:
        return json.load(fh)


def test_reference_shaped_pipeline_matches_golden(golden, native_lib, tmp_path):
    from mirge_amd import annotate
    from mirge_amd.engine import Engine
    exp = golden["expected"]
    ixdir = tmp_path / "index.Libs"
    ixdir.mkdir()
    fname = {"mirna": "mirna_miRBase", "hairpin": "hairpin_miRBase"}
    prefix = {}
    for key, (names, seqs) in golden["libraries"].items():
        prefix[key] = str(ixdir / ("syn_" + fname.get(key, key)))
        with open(prefix[key] + ".fa", "w") as fh:
            for n, s in zip(names, seqs):
                fh.write(">%s\n%s\n" % (n, s))
    names, seqs = golden["libraries"]["mirna"]
    fa = tmp_path / "mirna_SNP_pseudo.fa"
    fa.write_text("".join(">%s\n%s\n" % (n, s[2:-6]) for n, s in zip(names, seqs)))
    merges = tmp_path / "merges.csv"
    merges.write_text("".join(l + "\n" for l in golden["merges"]))

    sample_list = golden["sample_list"]
    seq_dic, len_dic = {}, {}
    for si, reads in enumerate(golden["samples"]):
        annotate.quantReads(reads, seq_dic, len_dic, len(sample_list), si)
    log_dic = {"quantStats": [{} for _ in sample_list], "annotStats": []}
    eng = Engine(0)
    annot_names = ["exact miRNA", "hairpin miRNA", "mature tRNA", "primary tRNA", "snoRNA", "rRNA",
                   "ncrna others", "mRNA", "isomiR miRNA"]
    annotate.runAnnotationPipeline(
        eng, seq_dic, "1", False, annot_names, str(tmp_path), log_dic, prefix["mirna"],
        prefix["hairpin"], prefix["mature_trna"], prefix["pre_trna"], prefix["snorna"], prefix["rrna"],
        prefix["ncrna_others"], prefix["mrna"], False, None, False, None, None, "miRBase", False, None,
        None, sample_list)
    assert set(seq_dic) == set(exp["seqDic"])
    for s, rec in seq_dic.items():
        assert rec["annot"] == exp["seqDic"][s]["annot"], s
        assert rec["quant"] == exp["seqDic"][s]["quant"]
    got_stats = [{k: a[k] for k in ("readsProcessed", "readsAligned")} for a in log_dic["annotStats"]]
    assert got_stats == exp["annotStats"]

    mir_dic = {}
    annotate.summarize(seq_dic, sample_list, log_dic, mir_dic, prefix["mirna"], str(tmp_path), False, eng)
    assert mir_dic == exp["mirDic_after_summarize"]
    assert log_dic["quantStats"] == exp["quantStats_after_summarize"]
    name_seq = {}
    annotate.miRNAmerge(str(merges), sample_list, mir_dic, str(fa), name_seq)
    assert mir_dic == exp["mirDic_after_merge"]
    annotate.filter(mir_dic, sample_list, log_dic, golden["cano_ratio"])
    assert mir_dic == exp["mirDic_after_filter"]
    assert log_dic["quantStats"] == exp["quantStats_after_filter"]


def test_reads_of_every_length_match_the_reference(native_lib, tmp_path):
    """tests/golden/long_reads.json (768 reads of 33..300 nt through the reference's own collapse + cascade, which
    offers a read of any length to every pass: RAP:543-554): the product's annot rows equal the reference's for EVERY
    read -- two, four and eight packed words, and the 35 reads beyond 255 nt through the long-read lane
    (mrg_cascade_run_long), 28 of which the reference annotates -- and the per-pass counters are those of the
    reference's unrestricted run."""
    from mirge_amd import annotate
    from mirge_amd.engine import Engine
    with open(os.path.join(ROOT, "tests", "golden", "long_reads.json")) as fh:
        g = json.load(fh)
    exp = g["expected"]
    fname = {"mirna": "mirna_miRBase", "hairpin": "hairpin_miRBase"}
    prefix = {}
    for key, (names, seqs) in g["libraries"].items():
        prefix[key] = str(tmp_path / ("syn_" + fname.get(key, key)))
        with open(prefix[key] + ".fa", "w") as fh:
            for n, s in zip(names, seqs):
                fh.write(">%s\n%s\n" % (n, s))
    seq_dic, len_dic = {}, {}
    annotate.quantReads(g["samples"][0], seq_dic, len_dic, 1, 0)
    assert {str(k): v for k, v in len_dic.items()} == exp["readLengthDic"]
    log_dic = {"quantStats": [{}], "annotStats": []}
    eng = Engine(0)
    annot_names = ["exact miRNA", "hairpin miRNA", "mature tRNA", "primary tRNA", "snoRNA", "rRNA",
                   "ncrna others", "mRNA", "isomiR miRNA"]
    annotate.runAnnotationPipeline(
        eng, seq_dic, "1", False, annot_names, str(tmp_path), log_dic, prefix["mirna"],
        prefix["hairpin"], prefix["mature_trna"], prefix["pre_trna"], prefix["snorna"], prefix["rrna"],
        prefix["ncrna_others"], prefix["mrna"], False, None, False, None, None, "miRBase", False, None,
        None, g["sample_list"])
    assert set(seq_dic) == set(exp["seqDic"])
    n_by_words = {2: 0, 4: 0, 8: 0, 9: 0}
    for s, rec in seq_dic.items():
        assert rec["quant"] == exp["seqDic"][s]["quant"]
        assert rec["annot"] == exp["seqDic"][s]["annot"], s
        if len(s) > 32 and rec["annot"][0]:
            n_by_words[2 if len(s) <= 64 else (4 if len(s) <= 128 else (8 if len(s) <= 255 else 9))] += 1
    assert min(n_by_words.values()) > 20 and n_by_words[9] == 28
    got_stats = [{k: a[k] for k in ("readsProcessed", "readsAligned")} for a in log_dic["annotStats"]]
    assert got_stats == exp["annotStats"] != exp["annotStats_le255"]
    # the alignments of the long reads are left for the downstream consumers like everyone else's
    assert sum(1 for s in log_dic["_alignments"] if len(s) > 255) == 28


def test_gff_path_matches_reference(native_lib, tmp_path):
    """-gff: the isomiRContentDic the reference fills during its cascade and the per-sample
    GFF files, reproduced from the GPU alignments (tests/golden/isomir_gff.json)."""
    from mirge_amd import annotate, isomir, report
    from mirge_amd.engine import Engine
    with open(os.path.join(ROOT, "tests", "golden", "isomir_gff.json")) as fh:
        g = json.load(fh)
    exp = g["expected"]
    fname = {"mirna": "mirna_miRBase", "hairpin": "hairpin_miRBase"}
    prefix = {}
    for key, (names, seqs) in g["libraries"].items():
        prefix[key] = str(tmp_path / ("syn_" + fname.get(key, key)))
        with open(prefix[key] + ".fa", "w") as fh:
            for n, s in zip(names, seqs):
                fh.write(">%s\n%s\n" % (n, s))
    gff3 = tmp_path / "syn_miRBase.gff3"
    gff3.write_text(g["gff3"])
    pre_name = isomir.extract_premir_name(str(gff3), "miRBase")
    assert pre_name == exp["miRNamePreNameDic"]
    sample_list = g["sample_list"]
    seq_dic, len_dic = {}, {}
    for si, reads in enumerate(g["samples"]):
        annotate.quantReads(reads, seq_dic, len_dic, len(sample_list), si)
    log_dic = {"quantStats": [{} for _ in sample_list], "annotStats": []}
    content = {}
    eng = Engine(0)
    annot_names = ["exact miRNA", "hairpin miRNA", "mature tRNA", "primary tRNA", "snoRNA", "rRNA",
                   "ncrna others", "mRNA", "isomiR miRNA"]
    annotate.runAnnotationPipeline(
        eng, seq_dic, "1", False, annot_names, str(tmp_path), log_dic, prefix["mirna"], prefix["hairpin"],
        prefix["mature_trna"], prefix["pre_trna"], prefix["snorna"], prefix["rrna"], prefix["ncrna_others"],
        prefix["mrna"], False, None, True, pre_name, content, "miRBase", False, None, None, sample_list)
    assert {s: r["annot"] for s, r in seq_dic.items()} == exp["seqDic_annot"]
    assert content == exp["isomiRContentDic_after_cascade"]
    isomir.write_isomir_gff(str(tmp_path), sample_list, content, seq_dic, "miRBase")
    for fn, lines in exp["gff_files"].items():
        got = open(str(tmp_path / fn)).read().split("\n")
        assert got[:4] == lines[:4]
        assert sorted(got[4:]) == sorted(lines[4:])
