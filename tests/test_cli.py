"""The annotate command line: flag surface (CPU) and an end-to-end run from FASTQ files on
the GPU, checked table by table against the oracle pipeline."""
import copy
import os

import numpy as np
import pytest

from mirge_amd import cli


def test_annotate_flags_match_reference_parser():
    """parseArgument.py:31-53: same flags, string-typed -cpu / -ex defaults."""
    ap = cli.build_parser()
    a = ap.parse_args(["annotate", "-s", "a.fastq", "b.fastq.gz", "-lib", "/L", "-sp", "human", "-pb", "/x"])
    assert a.sampleList == ["a.fastq", "b.fastq.gz"] and a.cpu == "1" and a.canoRatio == "0.1"
    assert a.miRNA_database == "miRBase" and a.adapter == "none" and not a.gff_output
    a = ap.parse_args(["annotate", "-s", "x.fastq", "-lib", "/L", "-sp", "mouse", "-d", "MirGeneDB", "-ex", "0.2",
                       "-di", "-gff", "-tcf", "-cpu", "4", "-phred64", "-ai", "-trf", "-spikeIn"])
    assert (a.miRNA_database, a.canoRatio, a.cpu) == ("MirGeneDB", "0.2", "4")
    assert a.diff_isomirs and a.gff_output and a.trimmed_collapsed_fa and a.phred64 and a.a_to_i


def test_sample_list_resolution(tmp_path):
    f1, f2 = tmp_path / "a.fastq", tmp_path / "b.fastq.gz"
    f1.write_text("")
    f2.write_text("")
    assert cli.resolve_samples([str(f1), str(f2)]) == [str(f1), str(f2)]
    lst = tmp_path / "samples.txt"
    lst.write_text("%s\n%s\n%s\n" % (f1, f2, f1))
    assert cli.resolve_samples([str(lst)]) == [str(f1), str(f2)]
    with pytest.raises(SystemExit):
        cli.resolve_samples([str(tmp_path / "missing.fastq")])
    with pytest.raises(SystemExit):
        cli.resolve_samples([str(f1), str(lst)])


def write_fastq(path, reads, rng, adapter=""):
    with open(path, "w") as fh:
        for i, r in enumerate(reads):
            r = (r + adapter)[:max(len(r), 50)] if adapter else r
            q = rng.integers(25, 41, len(r))
            if rng.random() < 0.3:                       # a low-quality 3' tail that gets trimmed
                r = r + "ACGT"[int(rng.integers(0, 4))] * 3
                q = np.concatenate([q, [2, 2, 2]])
            fh.write("@r%d\n%s\n+\n%s\n" % (i, r, "".join(chr(int(x) + 33) for x in q)))


@pytest.mark.gpu
@pytest.mark.parametrize("adapter", ["none", "illumina"])
def test_cli_end_to_end_against_oracle(native_lib, oracle_lib, tmp_path, adapter):
    from mirge_amd import report, synth
    from oracle import cascade, ingest as oingest, model
    from tests.golden.make_golden import SHAPES
    rng = np.random.default_rng(8)
    libs = synth.SynthLibraries(seed=123, scale=1.0, n_paralogs=6, n_snp=8, shapes=SHAPES)
    libs.write_layout(str(tmp_path / "libs"), species="syn", db="miRBase")
    fastqs = []
    for si in range(2):
        reads = [synth.codes_to_str(c) for c in synth.synth_reads(libs, 1500, seed=40 + si, zipf_s=1.3)]
        reads += ["ACGTNACGTTAGCATCGATCGA", "TTTTTTTTTTTTTTTTTTTT", "ACGTACGTAC"]      # N, poly-T, too short
        p = str(tmp_path / ("s%d.fastq" % si))
        # raw small-RNA reads carry the 3' adapter: 50-cycle reads, insert + adapter prefix
        write_fastq(p, reads, rng, adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC" if adapter == "illumina" else "")
        fastqs.append(p)
    out = cli.annotate_main(cli.build_parser().parse_args(
        ["annotate", "-s"] + fastqs + ["-lib", str(tmp_path / "libs"), "-sp", "syn", "-o", str(tmp_path),
                                       "-di", "-tcf", "-ad", adapter, "-cpu", "3"]), materialize=True)
    # ---- the same run through the oracle ----
    from mirge_amd.ingest import resolve_adapter
    kept = [oingest.load_fastq(p, adapter=resolve_adapter(adapter))[0] for p in fastqs]
    seq_dic, len_dic = cascade.collapse(kept)
    olibs = {k: model.Library(*libs.libs[k]) for k in libs.libs}
    log = {"quantStats": [{} for _ in fastqs], "annotStats": []}
    cascade.run_annotation_pipeline(seq_dic, olibs, log)
    mir = {}
    cascade.summarize(seq_dic, ["s0.fastq", "s1.fastq"], log, mir, olibs["mirna"].names)
    cascade.mirna_merge(libs.merges, ["s0.fastq", "s1.fastq"], mir)
    cascade.filter_mirnas(mir, ["s0.fastq", "s1.fastq"], log, "0.1")
    assert {s: r["annot"] for s, r in out["seqDic"].items()} == {s: r["annot"] for s, r in seq_dic.items()}
    assert {s: r["quant"] for s, r in out["seqDic"].items()} == {s: r["quant"] for s, r in seq_dic.items()}
    assert out["mirDic"] == mir
    assert out["readLengthDic"] == len_dic
    for a, b in zip(out["logDic"]["quantStats"], log["quantStats"]):
        for k, v in b.items():
            assert a[k] == v, k
    for a, b in zip(out["logDic"]["annotStats"], log["annotStats"]):
        assert (a["readsProcessed"], a["readsAligned"]) == (b["readsProcessed"], b["readsAligned"])
    # ---- files: written by the (golden-tested) writers from identical state ----
    want_dir = tmp_path / "want"
    want_dir.mkdir()
    wl = copy.deepcopy(out["logDic"])
    report.writeDataToCSV(str(want_dir), cli.ANNOT_NAMES, ["s0.fastq", "s1.fastq"], True, False, wl, seq_dic, mir)
    for fn in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "isomirs.csv", "isomirs.samples.csv"):
        got = sorted(open(os.path.join(out["outdir"], fn)).read().split("\n"))
        assert got == sorted(open(str(want_dir / fn)).read().split("\n")), fn
    rep = open(os.path.join(out["outdir"], "annotation.report.csv")).read().split("\n")
    assert rep[1].split(",")[:3] == ["s0.fastq", "1503", str(len(kept[0]))]
    if adapter == "illumina":      # the adapter was found and cut: the kept reads are the inserts again
        assert sum(len(r) <= 26 for r in kept[0]) > 0.95 * len(kept[0])
    assert os.path.exists(os.path.join(out["outdir"], "s0.trim.collapse.fa"))


@pytest.mark.gpu
def test_cli_spikein_tcf_matches_reference_files(native_lib, tmp_path):
    """`annotate -spikeIn -tcf` from FASTQ: ten-pass cascade, spike-in column, `.trim.collapse.fa`
    -> the files of the reference's own run (tests/golden/flags.json)."""
    import json
    import types
    from mirge_amd import synth
    from tests.conftest import ROOT
    with open(os.path.join(ROOT, "tests", "golden", "flags.json")) as fh:
        golden = json.load(fh)
    ns = types.SimpleNamespace(libs={k: tuple(v) for k, v in golden["libraries"].items()}, merges=golden["merges"])
    root = str(tmp_path / "libs")
    synth.SynthLibraries.write_layout(ns, root, species="syn", db="miRBase")
    fastqs = []
    for name, reads in zip(golden["sample_list"], golden["samples"]):
        p = str(tmp_path / name)
        with open(p, "w") as fh:
            for k, r in enumerate(reads):
                fh.write("@r%d\n%s\n+\n%s\n" % (k, r, "I" * len(r)))
            for k in range(5):                      # five reads below the 16-nt minimum (totalReads - trimmedReads)
                fh.write("@short%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % k)
        fastqs.append(p)
    out = cli.annotate_main(cli.build_parser().parse_args(
        ["annotate", "-s"] + fastqs + ["-lib", root, "-sp", "syn", "-o", str(tmp_path), "-spikeIn", "-tcf"]),
        materialize=True)
    exp = golden["expected"]
    for a, b in zip(out["logDic"]["annotStats"], exp["annotStats"]):
        assert (a["readsProcessed"], a["readsAligned"]) == (b["readsProcessed"], b["readsAligned"])
    assert [q["spikeInReads"] for q in out["logDic"]["quantStats"]] == exp["spikeInReads"]
    for fn, want in exp["files"].items():
        got = open(os.path.join(out["outdir"], fn)).read().split("\n")
        if fn in ("mapped.csv", "unmapped.csv"):      # row order = dict order of the collapse
            assert got[0] == want[0] and sorted(got) == sorted(want), fn
        elif fn == "miR.RPM.csv":
            # Python 2 prints str(float) with 12 significant digits (reproduced by the product);
            # the golden was captured under Python 3 (17 digits)
            assert got[0] == want[0] and len(got) == len(want)
            for g, w in zip(got[1:], want[1:]):
                gf, wf = g.split(","), w.split(",")
                assert gf[0] == wf[0] and len(gf) == len(wf)
                for a, b in zip(gf[1:], wf[1:]):
                    assert abs(float(a) - float(b)) <= 1e-9 * max(1.0, abs(float(b))) and len(a) <= 14
        else:
            assert got == want, fn


@pytest.mark.gpu
@pytest.mark.parametrize("device_ingest", [False, True])
def test_cli_aligns_reads_of_every_length_as_the_reference_does(native_lib, tmp_path, device_ingest):
    """`annotate -ad none` on an untrimmed long-cycle run (tests/golden/long_reads.json: reads of 33..300 nt through
    the reference's own collapse + cascade): FASTQ -> host parser (the device parser hands a file with reads beyond
    255 nt over to it) -> device collapse of the packed reads, host collapse of the longer ones -> cascade +
    long-read lane -> tables.  EVERY read carries the reference's annotation, the per-pass counters and the
    read-length table are those of the reference's run, and the annotated long reads are rows of mapped.csv."""
    import json
    import types
    from mirge_amd import synth
    from tests.conftest import ROOT
    with open(os.path.join(ROOT, "tests", "golden", "long_reads.json")) as fh:
        golden = json.load(fh)
    ns = types.SimpleNamespace(libs={k: tuple(v) for k, v in golden["libraries"].items()}, merges=[])
    root = str(tmp_path / "libs")
    synth.SynthLibraries.write_layout(ns, root, species="syn", db="miRBase")
    p = str(tmp_path / golden["sample_list"][0])
    reads = golden["samples"][0]
    with open(p, "w") as fh:
        for k, r in enumerate(reads):
            fh.write("@r%d\n%s\n+\n%s\n" % (k, r, "I" * len(r)))
    out = cli.annotate_main(cli.build_parser().parse_args(
        ["annotate", "-s", p, "-lib", root, "-sp", "syn", "-o", str(tmp_path), "-ad", "none"] +
        (["--device-ingest"] if device_ingest else [])), materialize=True)
    exp = golden["expected"]
    assert set(out["seqDic"]) == set(reads)
    for s, rec in out["seqDic"].items():
        assert rec["quant"] == exp["seqDic"][s]["quant"], s
        assert rec["annot"] == exp["seqDic"][s]["annot"], s
    got_stats = [{k: a[k] for k in ("readsProcessed", "readsAligned")} for a in out["logDic"]["annotStats"]]
    assert got_stats == exp["annotStats"]
    assert out["readLengthDic"] == {int(k): v for k, v in exp["readLengthDic"].items()}
    mapped = open(os.path.join(out["outdir"], "mapped.csv")).read().split("\n")
    unmapped = open(os.path.join(out["outdir"], "unmapped.csv")).read().split("\n")
    for table, flag in ((mapped, 1), (unmapped, 0)):
        want = {s for s, r in exp["seqDic"].items() if r["annot"][0] == flag}
        rows = {l.split(",")[0]: l.split(",") for l in table[1:] if l}
        assert set(rows) == want
        for s, f in rows.items():
            assert f[1:11] == [str(x) for x in exp["seqDic"][s]["annot"]] and f[11:] == [str(q) for q in exp["seqDic"][s]["quant"]]
    assert sum(1 for l in mapped if len(l.split(",")[0]) > 255) == 28


@pytest.mark.gpu
def test_cli_runs_on_a_library_directory_that_holds_only_ebwt_files(native_lib, ebwt_writer, tmp_path):
    """The reference's own layout: index.Libs/<sp>_*.1.ebwt and nothing else (MAIN:262-281).  The
    `.1.ebwt` files here come from tests/helpers/ebwt_writer.cpp (no bowtie-build in the image), so this
    pins the wiring -- prefix resolution, names for the histogram bins (SUM:6), sequences for the
    cascade -- not bowtie's byte layout."""
    import glob
    from mirge_amd import synth
    from mirge_amd.index import FmIndex
    from tests.golden.make_golden import SHAPES
    rng = np.random.default_rng(21)
    libs = synth.SynthLibraries(seed=123, scale=1.0, n_paralogs=6, n_snp=8, shapes=SHAPES)
    fastq = str(tmp_path / "s.fastq")
    write_fastq(fastq, [synth.codes_to_str(c) for c in synth.synth_reads(libs, 1500, seed=5, zipf_s=1.3)], rng)
    outs = {}
    for kind in ("fa", "ebwt"):
        root = str(tmp_path / ("libs_" + kind))
        libs.write_layout(root, species="syn", db="miRBase")
        if kind == "ebwt":
            for fa in glob.glob(os.path.join(root, "syn", "index.Libs", "*.fa")):
                d = FmIndex.from_fasta(fa).name_seq_dict()
                ebwt_writer(fa[:-3], list(d), list(d.values()), ftab_chars=5)
                os.remove(fa)
            assert not glob.glob(os.path.join(root, "syn", "index.Libs", "*.fa"))
        outs[kind] = cli.annotate_main(cli.build_parser().parse_args(
            ["annotate", "-s", fastq, "-lib", root, "-sp", "syn", "-o", str(tmp_path / ("out_" + kind)), "-di"]))
    for fn in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "isomirs.csv", "annotation.report.csv"):
        assert open(os.path.join(outs["fa"]["outdir"], fn)).read() == open(os.path.join(outs["ebwt"]["outdir"], fn)).read(), fn
    # (what was read back from a `.1.ebwt` is never cached next to the library: the reader is unpinned)
    assert not glob.glob(os.path.join(str(tmp_path / "libs_ebwt"), "syn", "index.Libs", "*.mrgfm"))


@pytest.mark.gpu
def test_cli_two_gpus_over_rccl(native_lib, tmp_path):
    """`annotate --gpus 2` for real: two child processes, one GPU each, RCCL all-to-all / all-reduce /
    send-recv (skipped on a one-GPU box; the gloo tests of tests/test_dist_gloo.py cover the logic)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from mirge_amd import synth
    from tests.golden.make_golden import SHAPES
    rng = np.random.default_rng(8)
    libs = synth.SynthLibraries(seed=123, scale=1.0, n_paralogs=6, n_snp=8, shapes=SHAPES)
    root = str(tmp_path / "libs")
    libs.write_layout(root, species="syn", db="miRBase")
    fastqs = []
    for si in range(3):
        p = str(tmp_path / ("s%d.fastq" % si))
        write_fastq(p, [synth.codes_to_str(c) for c in synth.synth_reads(libs, 2000, seed=40 + si, zipf_s=1.3)], rng)
        fastqs.append(p)
    base = [sys.executable, "-m", "mirge_amd", "annotate", "-s"] + fastqs + ["-lib", root, "-sp", "syn", "-di"]
    outs = {}
    for n_gpu in (1, 2):
        o = str(tmp_path / ("out%d" % n_gpu))
        cp = subprocess.run(base + ["-o", o, "--gpus", str(n_gpu)], capture_output=True, text=True, timeout=900)
        assert cp.returncode == 0, cp.stderr[-2000:]
        outs[n_gpu] = os.path.join(o, os.listdir(o)[0])
    for fn in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "isomirs.csv", "annotation.report.csv"):
        a = open(os.path.join(outs[1], fn)).read()
        b = open(os.path.join(outs[2], fn)).read()
        assert sorted(a.split("\n")) == sorted(b.split("\n")), fn


@pytest.mark.gpu
def test_cli_device_ingest_writes_the_same_tables(native_lib, tmp_path):
    """`--device-ingest`: the FASTQ records are split, trimmed and packed on the GPU (one plain file, one
    gzip file; a third with blank lines between records falls back to the host parser on its own):
    every table equals the default run's."""
    from mirge_amd import synth
    from tests.golden.make_golden import SHAPES
    rng = np.random.default_rng(9)
    libs = synth.SynthLibraries(seed=123, scale=1.0, n_paralogs=6, n_snp=8, shapes=SHAPES)
    root = str(tmp_path / "libs")
    libs.write_layout(root, species="syn", db="miRBase")
    fastqs = []
    for si in range(3):
        p = str(tmp_path / ("s%d.fastq" % si))
        write_fastq(p, [synth.codes_to_str(c) for c in synth.synth_reads(libs, 1500, seed=60 + si, zipf_s=1.3)], rng)
        fastqs.append(p)
    import gzip
    with open(fastqs[1], "rb") as fi, gzip.open(fastqs[1] + ".gz", "wb") as fo:
        fo.write(fi.read())
    os.remove(fastqs[1])
    fastqs[1] += ".gz"
    text = open(fastqs[2]).read().split("\n@")
    open(fastqs[2], "w").write(text[0] + "\n\n@" + "\n@".join(text[1:]))
    outs = {}
    for mode in ("host", "device"):
        argv = ["annotate", "-s"] + fastqs + ["-lib", root, "-sp", "syn", "-o", str(tmp_path / mode), "-di", "-ad", "+1"]
        outs[mode] = cli.annotate_main(cli.build_parser().parse_args(argv + (["--device-ingest"] if mode == "device" else [])))
    for fn in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "isomirs.csv", "annotation.report.csv"):
        a = open(os.path.join(outs["host"]["outdir"], fn)).read()
        b = open(os.path.join(outs["device"]["outdir"], fn)).read()
        assert a == b, fn
    assert outs["host"]["logDic"]["quantStats"][0]["trimmedReads"] == outs["device"]["logDic"]["quantStats"][0]["trimmedReads"] > 1000
