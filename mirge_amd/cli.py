"""`python -m mirge_amd annotate ...` -- the annotate-mode command line of miRge2.0
(parseArgument.py:31-53) driving the GPU engine.

Same flags, library directory layout (MAIN:108-112, :262-281) and output tables
(`miRge.<timestamp>/annotation.report.csv`, `mapped.csv`, `unmapped.csv`, `miR.Counts.csv`,
`miR.RPM.csv`, `-di` isomiR tables, `-gff` per-sample GFF).  Differences:
  * `-pb` is accepted and ignored (no bowtie); `--gpu` picks the device;
  * index files are `<prefix>.mrgfm` or `<prefix>.fa` instead of `<prefix>.*.ebwt`
    (build with `python -m mirge_amd.build_index`);
  * `-ad illumina|ion|<sequence>|+N` is applied as trim_file.py does (3' quality trimming,
    cutadapt's 3' adapter search restated, 16-nt minimum);
  * `-ai` reads the genome from `<sp>_genome.mrgfm` / `.fa` or `<sp>_genome.partNNN.mrgfm`
    (`build_index --max-bases 500000000`) and answers the two genome bowtie runs on the GPU;
  * `-trf` writes `tRFs.potential.report.tsv`, `tRF.Counts.csv`, `tRF.RP100K.csv` and
    `discarded.reads.summary.assigningtRFs.csv`; the per-sample clustering reports
    (`tRFs.samples.tmp/`) and the PDF report are not produced.
Call order follows MAIN:346-389.
"""
import argparse
import os
import sys
import time

import numpy as np

ANNOT_NAMES = ["exact miRNA", "hairpin miRNA", "mature tRNA", "primary tRNA", "snoRNA", "rRNA",
               "ncrna others", "mRNA", "isomiR miRNA"]  # MAIN:335


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m mirge_amd", description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="command")
    p = sub.add_parser("annotate", help="annotate small-RNA reads (miRge2.0 annotate mode)")
    p.add_argument("-s", nargs="*", required=True, dest="sampleList", metavar="sample <required>")
    p.add_argument("-o", default=os.getcwd(), dest="output_dir", metavar="<dir>")
    p.add_argument("-d", default="miRBase", dest="miRNA_database", metavar="<string required>")
    p.add_argument("-pb", default=None, dest="bowtieBinary", metavar="<dir>", help="ignored (no bowtie)")
    p.add_argument("-lib", required=True, dest="libraryPath", metavar="<dir required>")
    p.add_argument("-sp", required=True, dest="species", metavar="<string required>")
    p.add_argument("-ex", default="0.1", dest="canoRatio", metavar="<float>")
    p.add_argument("-ad", default="none", dest="adapter", metavar="<string>")
    p.add_argument("-phred64", action="store_true")
    p.add_argument("-spikeIn", dest="spikeIn", action="store_true")
    p.add_argument("-tcf", dest="trimmed_collapsed_fa", action="store_true")
    p.add_argument("-di", dest="diff_isomirs", action="store_true")
    p.add_argument("-cpu", dest="cpu", metavar="<int>", default="1")
    p.add_argument("-ai", dest="a_to_i", action="store_true")
    p.add_argument("-gff", dest="gff_output", action="store_true")
    p.add_argument("-trf", dest="trf_output", action="store_true")
    p.add_argument("--gpu", type=int, default=0, help="device index (default 0)")
    return ap


def _die(msg):
    print(msg, file=sys.stderr)
    sys.exit(1)


def resolve_samples(sample_args):
    """MAIN:289-314: either *.fastq[.gz] paths or ONE file listing them."""
    def is_fastq(p):
        b = os.path.basename(p)
        return b.split(".")[-1] == "fastq" or ".".join(b.split(".")[-2:]) == "fastq.gz"
    if all(is_fastq(s) for s in sample_args):
        for s in sample_args:
            if not os.path.isfile(os.path.abspath(s)):
                print("%s can't be found in current directory. Please check it." % s)
                sys.exit(1)
        return list(sample_args)
    if len(sample_args) == 1:
        out = []
        try:
            with open(sample_args[0]) as fh:
                for line in fh:
                    line = line.strip()
                    if line and line not in out:
                        if not os.path.isfile(os.path.abspath(line)):
                            print("%s cannot be found, please check the path of the sample file." % line)
                            sys.exit(1)
                        out.append(line)
        except IOError:
            print("%s is not a file, please check it." % sample_args[0])
            sys.exit(1)
        return out
    print("The format of input argument '-s' is wrong, please check it.")
    sys.exit(1)


def annotate_main(args):
    from . import annotate, ingest, pack, report
    from .engine import Engine
    db = {"mirbase": "miRBase", "mirgenedb": "MirGeneDB"}.get(args.miRNA_database.lower())
    if db is None:
        _die("The value of parameter '-d' is invalid. Please check it")
    sp, lib = args.species, args.libraryPath
    if args.trf_output and sp != "human":  # MAIN:161-163
        _die("tRF detection is only supported for the species of human. Please check it.")
    trf_tables, trf_content = None, None
    if args.trf_output:
        from . import trf
        trf_tables = trf.load_trf_tables(lib, sp)
        trf_content = {}
    index_dir = os.path.join(lib, sp, "index.Libs")
    mirna_fa = os.path.join(lib, sp, "fasta.Libs", "%s_mirna_SNP_pseudo_%s.fa" % (sp, db))
    merge_file = os.path.join(lib, sp, "annotation.Libs", "%s_merges_%s.csv" % (sp, db))
    merged_name = {}
    with open(merge_file) as fh:  # MAIN:114-120
        for line in fh:
            f = line.strip().split(",")
            for item in f[1:]:
                merged_name[item] = f[0]
    pre_name, content = None, None
    if args.gff_output:
        from . import isomir
        pre_name = isomir.extract_premir_name(
            os.path.join(lib, sp, "annotation.Libs", "%s_%s.gff3" % (sp, db)), db)
        content = {}
    kinds = ["mirna_" + db, "hairpin_" + db, "mrna", "mature_trna", "pre_trna", "snorna", "rrna", "ncrna_others"]
    if args.spikeIn:
        kinds.append("spike-in")
    prefix = {}
    for kind in kinds:  # MAIN:262-267, with our index formats
        p = os.path.join(index_dir, "%s_%s" % (sp, kind))
        if not (os.path.isfile(p + ".mrgfm") or os.path.isfile(p + ".fa")):
            print("The index file of %s_%s (.mrgfm or .fa) is not located at %s, please check it."
                  % (sp, kind, index_dir))
            sys.exit(1)
        prefix[kind] = p
    outdir = os.path.join(os.path.abspath(args.output_dir),
                          "miRge." + time.strftime("%Y-%m-%d_%H-%M-%S", time.localtime()))
    os.makedirs(outdir)
    raw = resolve_samples(args.sampleList)
    sample_list = [os.path.basename(s)[:-3] if s.endswith(".gz") else os.path.basename(s) for s in raw]
    S = len(sample_list)

    engine = Engine(args.gpu)
    genome, removed_ai = None, []
    if args.a_to_i:  # MAIN:136-146, :277
        from . import a2i
        from .index import FmIndex
        rep = os.path.join(lib, sp, "annotation.Libs", "%s_miRNAs_in_repetitive_element_%s.csv" % (sp, db))
        if os.path.isfile(rep):
            with open(rep) as fh:
                for line in fh:
                    name = line.strip().split(",")[0]
                    if name not in removed_ai:
                        removed_ai.append(name)
        try:
            parts = FmIndex.open_prefix_parts(os.path.join(index_dir, "%s_genome" % sp))
        except FileNotFoundError:
            print("The index file of %s_genome (.mrgfm, .fa or .partNNN.mrgfm) is not located at %s, "
                  "please check it." % (sp, index_dir))
            sys.exit(1)
        keys = []
        for k, ix in enumerate(parts):
            engine.add_library("genome:%d" % k, ix)
            keys.append("genome:%d" % k)
        genome = a2i.EngineGenome(engine, keys)
    log_dic = {"quantStats": [], "annotStats": []}
    t0 = time.time()
    words_all, lens_all, nmask_all, sample_all = [], [], [], []
    any_n, W = False, 1
    loaded = []
    # trim_file per sample (MAIN:346-372); `-cpu` threads are shared out over the samples, which are
    # read concurrently (one inflate/record-splitting thread each plus its trimming workers)
    from concurrent.futures import ThreadPoolExecutor
    n_cpu = max(1, int(args.cpu))
    n_jobs = max(1, min(len(raw), n_cpu))

    def load_one(i):
        t1 = time.time()
        fq = ingest.load_fastq(os.path.abspath(raw[i]), adapter=args.adapter, threads=max(1, n_cpu // n_jobs))
        return fq, time.time() - t1

    for name in sample_list:
        print("Performing quantitation analysis of %s..." % name)
    with ThreadPoolExecutor(max_workers=n_jobs) as pool:
        results = list(pool.map(load_one, range(len(raw))))
    for i, (fq, dt) in enumerate(results):
        loaded.append(fq)
        W = max(W, fq["words"].shape[0])
        any_n = any_n or fq["nmask"] is not None
        log_dic["quantStats"].append({"filename": sample_list[i], "totalReads": fq["total"],
                                      "trimmedReads": fq["kept"], "cpuTime-trim": dt,
                                      "cpuTime-uniq": 0.0})
    t1 = time.time()
    for i, fq in enumerate(loaded):
        w = np.zeros((W, fq["kept"]), dtype=np.uint64)
        w[:fq["words"].shape[0]] = fq["words"]
        words_all.append(w)
        lens_all.append(fq["lens"])
        nm = np.zeros((W, fq["kept"]), dtype=np.uint64)
        if fq["nmask"] is not None:
            nm[:fq["nmask"].shape[0]] = fq["nmask"]
        nmask_all.append(nm)
        sample_all.append(np.full(fq["kept"], i, dtype=np.uint16))
    words = np.concatenate(words_all, axis=1)
    lens = np.concatenate(lens_all)
    nmask = np.concatenate(nmask_all, axis=1) if any_n else None
    sample = np.concatenate(sample_all)
    col = ingest.collapse(engine, words, lens, nmask, sample, n_samples=S,
                          max_len=max(fq["max_len"] for fq in loaded))
    for q in log_dic["quantStats"]:
        q["cpuTime-uniq"] = (time.time() - t1) / S
    seqs = pack.unpack_reads(col["words"], col["lens"], col["nmask"])
    spike = bool(args.spikeIn)
    width = 10 if spike else 9
    seq_dic = {s: {"quant": q, "annot": [0] + [""] * width, "length": len(s)}
               for s, q in zip(seqs, col["quant"].tolist())}
    read_len_dic = col["length_hist"]
    if args.trimmed_collapsed_fa:  # QNT:26-44
        for i, name in enumerate(sample_list):
            rows = sorted(((rec["quant"][i], s) for s, rec in seq_dic.items() if rec["quant"][i] > 0),
                          reverse=True)
            with open(os.path.join(outdir, os.path.splitext(name)[0] + ".trim.collapse.fa"), "w") as fh:
                for k, (c, s) in enumerate(rows):
                    fh.write(">seq%d_%d\n%s\n" % (k + 1, c, s))

    print("\nPerforming annotation for all of the collasped sequences...")
    t2 = time.time()
    names = list(ANNOT_NAMES) + (["spike-in"] if spike else [])
    annotate.runAnnotationPipeline(
        engine, seq_dic, args.cpu, args.phred64, names, outdir, log_dic, prefix["mirna_" + db],
        prefix["hairpin_" + db], prefix["mature_trna"], prefix["pre_trna"], prefix["snorna"], prefix["rrna"],
        prefix["ncrna_others"], prefix["mrna"], spike, prefix.get("spike-in"), args.gff_output, pre_name,
        content, db, args.trf_output, trf_tables["trnaStruDic"] if trf_tables else None, trf_content, sample_list)
    print("All annotation cycles completed (%.2f sec).\n" % (time.time() - t2))
    print("Summarizing and tabulating results...")
    t3 = time.time()
    mir_dic, name_seq = {}, {}
    annotate.summarize(seq_dic, sample_list, log_dic, mir_dic, prefix["mirna_" + db], outdir, spike, engine)
    annotate.miRNAmerge(merge_file, sample_list, mir_dic, mirna_fa, name_seq)
    annotate.filter(mir_dic, sample_list, log_dic, args.canoRatio)
    report.write_annotation_report_csv(os.path.join(outdir, "annotation.report.csv"), sample_list, log_dic, spike)
    report.writeDataToCSV(outdir, names, sample_list, args.diff_isomirs, args.a_to_i, log_dic, seq_dic, mir_dic,
                          name_seq, merged_name, spike, args.gff_output, content, db, args.trf_output,
                          genome=genome, removedMiRNAList=removed_ai, trfContentDic=trf_content,
                          trf_tables=trf_tables,
                          pretrnaNameSeqDic=engine.indexes["pre_trna"].name_seq_dict() if args.trf_output else None)
    print("Summary Complete (%.2f sec)" % (time.time() - t3))
    print("Annotation of miRge2.0 Completed (%.2f sec)" % (time.time() - t0))
    return dict(outdir=outdir, seqDic=seq_dic, mirDic=mir_dic, logDic=log_dic, readLengthDic=read_len_dic)


def main(argv=None):
    ap = build_parser()
    args = ap.parse_args(argv)
    if args.command != "annotate":
        ap.print_help()
        return 2
    annotate_main(args)
    return 0
