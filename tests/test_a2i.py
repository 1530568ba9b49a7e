"""mirge_amd.a2i (the -ai A-to-I tally) against the reference's own run
(tests/golden/a2i.json; the genome bowtie runs are the stand-in bowtie there, pairwise2 is the
generator's own restatement of Biopython's localms -- make_golden.pairwise2_localms, which imports
nothing from mirge_amd)."""
import copy
import json
import os

import pytest

from mirge_amd import a2i
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "a2i.json")) as fh:
        return json.load(fh)


def run_report(golden, genome, outdir):
    st = golden["state"]
    log_dic = {"quantStats": copy.deepcopy(st["quantStats"])}
    return a2i.a_to_i_report(str(outdir), golden["sample_list"], log_dic, st["seqDic"], st["mirDic"],
                             golden["mirNameSeqDic"], golden["mirMergedNameDic"], golden["removedMiRNAList"],
                             genome)


def check_files(golden, outdir):
    exp = golden["expected"]["files"]
    fn = "a2IEditing.report.csv"
    assert open(os.path.join(str(outdir), fn)).read().split("\n") == exp[fn], fn
    # newform: identical but for the log2RPM column, which the reference prints with Python 2's
    # 12-digit str(float) (the golden was captured under Python 3: 17 digits)
    fn = "a2IEditing.report.newform.csv"
    got = open(os.path.join(str(outdir), fn)).read().split("\n")
    assert len(got) == len(exp[fn]) and got[0] == exp[fn][0]
    for g, w in zip(got[1:], exp[fn][1:]):
        gc, wc = g.split(","), w.split(",")
        assert gc[:-1] == wc[:-1]
        if gc[-1] != wc[-1]:
            assert abs(float(gc[-1]) - float(wc[-1])) < 1e-10 and len(gc[-1]) <= 14
    # the detail file is written miRNA by miRNA in dict order (arbitrary in the reference)
    got = open(os.path.join(str(outdir), "a2IEditing.detail.txt")).read().split("\n")
    assert sorted(got) == sorted(exp["a2IEditing.detail.txt"])
    assert len(exp["a2IEditing.report.csv"]) > 10


def test_report_with_exhaustive_scan_genome(golden, oracle_lib, tmp_path):
    from oracle import model
    genome = model.ScanGenome(model.Library(*golden["libraries"]["genome"]))
    run_report(golden, genome, tmp_path)
    check_files(golden, tmp_path)


def test_judge_align_known_answers():
    t = "TGAGGTAGTAGGTTGTATAGTT"
    j = lambda s: a2i.judge_align(*a2i.local_pair(t, s))
    assert j(t) is True
    assert j(t[2:]) is False                  # read starting 2 nt late
    assert j(t[:8] + "CC" + t[10:]) is False  # 2 internal mismatches
    assert j(t[1:]) is True and j(t[:9] + "G" + t[10:]) is True
    assert a2i.refine_name("a0.fastq") == "a0" and a2i.refine_name("x.fastq.gz") == "x.gz"


def test_local_pair_is_the_best_ungapped_diagonal():
    t = "TGAGGTAGTAGGTTGTATAGTT"
    assert a2i.local_pair(t, "GAGGTAGTAGGTTGTATAGTTA") == (t + "-", "-GAGGTAGTAGGTTGTATAGTTA")
    assert a2i.local_pair(t, "TT" + t) == ("--" + t, "TT" + t)
    frame, states = a2i.align_to_target(t, [t, "A" + t, t[1:] + "CC"])
    assert len({len(x) for x in frame}) == 1 and frame[0].strip("-") == t and states == [True, True, True]


def test_local_pair_ties_follow_pairwise2_order():
    """Equal-score alignments: Biopython lists first the one ending furthest along the miRNA; a
    gapped alignment that runs through another best cell is a zero-score extension and is dropped;
    a gapped alignment that BEATS every ungapped run is taken (merged families with an indel)."""
    # homopolymer: A^17 inside A^18 fits at offsets 0 and 1 with 34 each -> the later one
    assert a2i.local_pair("GC" + "A" * 18 + "TG", "A" * 17) == ("GC" + "A" * 18 + "TG", "---" + "A" * 17 + "--")
    # dinucleotide repeat: offsets 2 and 4 tie
    assert a2i.local_pair("TG" + "AC" * 9 + "GT", "AC" * 8) == ("TG" + "AC" * 9 + "GT", "----" + "AC" * 8 + "--")
    # 22-nt miRNA, read lacks its 12th base: gapped 21 * 2 - 20 = 22 = the ungapped 11 pairs; the gapped
    # walk passes the cell that ends those 11 pairs -> dropped, the ungapped diagonal is listed first
    m = "TGCATCGGATC" + "G" + "TACCTGAAGT"
    assert a2i.local_pair(m, m[:11] + m[12:]) == (m, m[:11] + m[12:] + "-")
    # 24-nt miRNA, read lacks its 13th base: gapped 23 * 2 - 20 = 26 > 24
    m = "CAGTTCGAGCTA" + "T" + "GGACTTCAAGC"
    assert a2i.local_pair(m, m[:12] + m[13:]) == (m, m[:12] + "-" + m[13:])


def test_local_pair_equals_the_generators_pairwise2(golden):
    """Two independent formulations of the published algorithm (the product's depth-first walk over
    recomputed options, the generator's trace-bit matrix + stack) agree on random, low-complexity
    and indel pairs; the fixture really holds tie cases."""
    import random
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    calls = golden["pairwise2_calls"]
    assert calls["several_listed"] > 0 and calls["gapped_first"] > 0 and calls["calls"] > 1000
    rnd = random.Random(606)

    def rs(n, alpha):
        return "".join(rnd.choice(alpha) for _ in range(n))
    several = gapped = 0
    for _ in range(4000):
        alpha = rnd.choice(["ACGT", "AC", "A", "ACG", "AAAC"])
        t = rs(rnd.randint(16, 27), alpha)
        kind = rnd.random()
        if kind < 0.4:
            s = list(t[rnd.randint(0, 5):len(t) - rnd.randint(0, 4)])
            for _k in range(rnd.randint(0, 2)):
                s[rnd.randrange(len(s))] = rnd.choice("ACGT")
            s = "".join(s) + rs(rnd.randint(0, 3), "ACGT")
        elif kind < 0.7:
            k = rnd.randint(5, len(t) - 5)
            s = t[:k] + t[k + 1:]
        elif kind < 0.85:
            k = rnd.randint(5, len(t) - 5)
            s = t[:k] + rnd.choice("ACGT") + t[k:]
        else:
            s = rs(rnd.randint(16, 26), alpha)
        listed = make_golden.pairwise2_localms(t, s, 2, -1, -20, -20)
        several += len(listed) > 1
        gapped += "-" in listed[0][0].strip("-") or "-" in listed[0][1].strip("-")
        assert a2i.local_pair(t, s) == listed[0], (t, s)
    assert several > 300 and gapped > 20


@pytest.mark.gpu
def test_report_with_gpu_genome_filters(golden, native_lib, oracle_lib, tmp_path):
    """The two genome runs on the GPU (both strands, best-stratum multiplicity) give the same
    retained / removed sets as the exhaustive scan, hence the same report."""
    from mirge_amd.engine import Engine
    from mirge_amd.index import FmIndex
    from oracle import model
    eng = Engine(0)
    names, seqs = golden["libraries"]["genome"]
    keys = []
    for n, s in zip(names, seqs):           # one library per chromosome
        eng.add_library("genome:" + n, FmIndex.build([n], [s]))
        keys.append("genome:" + n)
    gpu = a2i.EngineGenome(eng, keys)
    scan = model.ScanGenome(model.Library(names, seqs))
    reads = [s for s, r in golden["state"]["seqDic"].items() if r["annot"][1] != "" or r["annot"][9] != ""]
    reads += ["ACGT" * 6, "ACGTACGTACGTACGTACGTAC", "TTTTTTTTTTTTTTTTTTTTTT"]
    assert gpu.unique_best(reads) == scan.unique_best(reads)
    assert gpu.exact_hit(reads) == scan.exact_hit(reads)
    assert 0 < len(scan.unique_best(reads)) < len(set(reads))
    run_report(golden, gpu, tmp_path)
    check_files(golden, tmp_path)


@pytest.mark.gpu
def test_cli_ai_from_fastq_matches_reference_files(golden, native_lib, tmp_path):
    """`annotate -ai` end to end from FASTQ: collapse, cascade, tally and the two genome filters on
    the GPU, genome split into parts (build_index --max-bases) -> the reference's a2IEditing files."""
    import types
    from mirge_amd import build_index, cli, synth
    ns = types.SimpleNamespace(libs={k: tuple(v) for k, v in golden["libraries"].items()},
                               merges=golden["merges"])
    root = str(tmp_path / "libs")
    synth.SynthLibraries.write_layout(ns, root, species="syn", db="miRBase")
    gfa = os.path.join(root, "syn", "index.Libs", "syn_genome.fa")
    longest = max(len(s) for s in golden["libraries"]["genome"][1])
    assert build_index.main([gfa, "--max-bases", str(longest + 10)]) == 0
    os.remove(gfa)
    assert len([f for f in os.listdir(os.path.dirname(gfa)) if ".part" in f]) >= 2
    with open(os.path.join(root, "syn", "annotation.Libs", "syn_miRNAs_in_repetitive_element_miRBase.csv"), "w") as fh:
        for n in golden["removedMiRNAList"]:
            fh.write("%s,x\n" % n)
    fastqs = []
    for name, reads in zip(golden["sample_list"], golden["samples"]):
        p = str(tmp_path / name)
        with open(p, "w") as fh:
            for k, r in enumerate(reads):
                fh.write("@r%d\n%s\n+\n%s\n" % (k, r, "I" * len(r)))
        fastqs.append(p)
    out = cli.annotate_main(cli.build_parser().parse_args(
        ["annotate", "-s"] + fastqs + ["-lib", root, "-sp", "syn", "-o", str(tmp_path), "-ai"]), materialize=True)
    st = golden["state"]
    assert {s: r["annot"] for s, r in out["seqDic"].items()} == {s: r["annot"] for s, r in st["seqDic"].items()}
    check_files(golden, out["outdir"])


@pytest.mark.gpu
def test_columnar_path_gives_the_same_a2i_report(golden, native_lib, tmp_path):
    """mirge_amd.columnar: arrays instead of seqDic through cascade, tally, merge and filter; only
    the reads the -ai block groups become dict records -> the reference's files."""
    import types
    import numpy as np
    from mirge_amd import columnar, ingest, pack, synth
    from mirge_amd.engine import Engine
    from mirge_amd.index import FmIndex
    ns = types.SimpleNamespace(libs={k: tuple(v) for k, v in golden["libraries"].items()}, merges=golden["merges"])
    root = str(tmp_path / "libs")
    prefix = synth.SynthLibraries.write_layout(ns, root, species="syn", db="miRBase")
    eng = Engine(0)
    for key, stem in (("mirna", "mirna_miRBase"), ("hairpin", "hairpin_miRBase"), ("mature_trna", "mature_trna"),
                      ("pre_trna", "pre_trna"), ("snorna", "snorna"), ("rrna", "rrna"),
                      ("ncrna_others", "ncrna_others"), ("mrna", "mrna")):
        eng.add_library(key, FmIndex.open_prefix(prefix + stem))
    keys = []
    for n, s in zip(*golden["libraries"]["genome"]):
        eng.add_library("genome:" + n, FmIndex.build([n], [s]))
        keys.append("genome:" + n)
    # raw reads of both samples -> packed -> GPU collapse (arrays, no dict)
    reads, sample = [], []
    for si, rs in enumerate(golden["samples"]):
        reads += rs
        sample += [si] * len(rs)
    w, l, nm = pack.pack_reads(reads)
    col = ingest.collapse(eng, w, l, nm, np.array(sample, dtype=np.uint16), n_samples=2, max_len=int(l.max()))
    cols = columnar.annotate_columns(eng, col["words"], col["lens"], col["nmask"], col["quant"])
    mir_dic, log_dic, name_seq = columnar.tables_from_columns(
        eng, cols, golden["sample_list"], os.path.join(root, "syn", "annotation.Libs", "syn_merges_miRBase.csv"),
        os.path.join(root, "syn", "fasta.Libs", "syn_mirna_SNP_pseudo_miRBase.fa"), "0.1")
    st = golden["state"]
    assert mir_dic == st["mirDic"]
    for a, b in zip(log_dic["quantStats"], st["quantStats"]):
        for k, v in b.items():
            assert a[k] == v, k
    sub = columnar.mirna_read_subset(eng, cols, col["words"], col["lens"], col["nmask"], col["quant"], log_dic,
                                     for_a2i=True)
    full = {s: r for s, r in st["seqDic"].items() if r["annot"][1] != "" or r["annot"][9] != ""}
    assert 0 < len(sub) <= len(full) and all(full[s]["annot"] == r["annot"] and full[s]["quant"] == r["quant"]
                                              for s, r in sub.items())
    a2i.a_to_i_report(str(tmp_path), golden["sample_list"], log_dic, sub, mir_dic, name_seq,
                      golden["mirMergedNameDic"], golden["removedMiRNAList"], a2i.EngineGenome(eng, keys))
    check_files(golden, tmp_path)
