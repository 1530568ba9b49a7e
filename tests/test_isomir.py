"""mirge_amd.isomir (the -gff classification) against known answers and end-to-end
vectors captured from the reference's own functions (tests/golden/isomir_gff.json)."""
import json
import os

import pytest

from mirge_amd import isomir
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "isomir_gff.json")) as fh:
        return json.load(fh)


def test_make_id(golden):
    for seq, want in golden["expected"]["make_id"]:
        assert isomir.make_id(seq) == want, seq
    # SURVEY.md 8c known answers
    assert isomir.make_id("TGAGGTAGTAGGTTGTATAGTT") == "7AwhRzwL2"
    assert isomir.make_id("TAGCTTATCAGACTGATGTTGA") == "JHemqbR@2"


def test_make_cigar(golden):
    for a, b, want in golden["expected"]["make_cigar"]:
        assert isomir.make_cigar(a, b) == want, (a, b)
    assert isomir.make_cigar("ACGTACGTACGT", "ACGTACGTACGT") == "12M"


def test_classify_alignment_700_cases(golden):
    kinds = set()
    for P, E, R, start, iv, want in golden["expected"]["classify"]:
        got = isomir.classify_alignment(P, E, R, start, iv)
        assert (None if got is None else list(got)) == want, (P, E, R, start, iv)
        if want:
            kinds.update(v.split(":")[0] for v in want[1].split(","))
    assert {"NA", "iso_snp", "iso_snp_seed", "iso_snp_central_offset", "iso_snp_central",
            "iso_snpcentral_supp", "iso_add", "iso_5p", "iso_3p"} <= kinds


def test_survey_known_answers():
    """precursor GGGATGAGAC + mature + TTAGGGTCACACCCACC."""
    mature = "TGAGGTAGTAGGTTGTATAGTT"
    pre = "GGGATGAGAC" + mature + "TTAGGGTCACACCCACC"
    lib = "AC" + mature + "TTAGGG"
    c = isomir.classify_alignment
    assert c(pre, lib, mature, 3, 0) == ("ref_miRNA", "NA", 11, 32, "22M")
    assert c(pre, lib, mature[:-2], 3, 0) == ("isomiR", "iso_3p:-2", 11, 30, "20M")
    assert c(pre, lib, "C" + mature, 2, 0) == ("isomiR", "iso_5p:+1", 10, 32, "23M")
    assert c(pre, lib, mature + "TT", 3, 0) == ("isomiR", "iso_3p:+2", 11, 34, "24M")
    assert c(pre, lib, mature + "A", 4, 8) == ("isomiR", "iso_add:+1", 11, 33, "22MA")
    snp = mature[:9] + "C" + mature[10:]
    assert c(pre, lib, snp, 4, 8) == ("isomiR", "iso_snp", 11, 32, "9MC12M")
    assert c(pre, lib, mature[1:], 4, 0) == ("isomiR", "iso_5p:-1", 12, 32, "21M")


def test_infer_and_extract_premir_name(golden, tmp_path):
    tab = golden["expected"]["inferPremiRName"]["table"]
    for name, db, want in golden["expected"]["inferPremiRName"]["cases"]:
        assert isomir.infer_premir_name(name, tab, db) == want, name
    p = tmp_path / "x.gff3"
    p.write_text(golden["gff3"])
    assert isomir.extract_premir_name(str(p), "miRBase") == golden["expected"]["miRNamePreNameDic"]


def test_content_and_gff_from_golden_alignments(golden, tmp_path):
    """build_isomir_content + write_isomir_gff fed with the alignments the reference saw
    (start/miRName of its isomiRContentDic): host-only, no GPU."""
    exp = golden["expected"]
    libs = golden["libraries"]
    hairpin = dict(zip(*libs["hairpin"]))
    mirna = dict(zip(*libs["mirna"]))
    want = exp["isomiRContentDic_after_cascade"]
    annot = exp["seqDic_annot"]
    content = {}
    for pass_index, slot in ((0, 1), (8, 9)):
        hits = {}
        for read, rec in want.items():
            if annot[read][slot] != "":
                trim = 0 if pass_index == 0 else 3
                hits[read] = (rec["miRName"], int(rec["start"]), "%dM" % (len(read) - trim))
        # reads the reference dropped (mature not in precursor) are absent from `want`; fine
        isomir.build_isomir_content(content, hits, pass_index, exp["miRNamePreNameDic"], hairpin, mirna,
                                    "miRBase")
    assert content == want
    from mirge_amd.annotate import quantReads
    seq_dic, len_dic = {}, {}
    for si, reads in enumerate(golden["samples"]):
        quantReads(reads, seq_dic, len_dic, 2, si)
    isomir.write_isomir_gff(str(tmp_path), golden["sample_list"], content, seq_dic, "miRBase")
    for fn, lines in exp["gff_files"].items():
        got = open(str(tmp_path / fn)).read().split("\n")
        assert got[:4] == lines[:4]
        assert sorted(got[4:]) == sorted(lines[4:])
