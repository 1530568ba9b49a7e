"""`python -m mirge_amd annotate ...` -- the annotate-mode command line of miRge2.0
(parseArgument.py:31-53) driving the GPU engine.

Same flags, library directory layout (MAIN:108-112, :262-281) and output tables
(`miRge.<timestamp>/annotation.report.csv`, `mapped.csv`, `unmapped.csv`, `miR.Counts.csv`,
`miR.RPM.csv`, `-di` isomiR tables, `-gff` per-sample GFF).  Differences:
  * `-pb` is accepted and ignored (no bowtie); `--gpu` picks the device, `--gpus N` runs one
    process per GPU over contiguous shards of the collapsed read set;
  * an index prefix resolves to `<prefix>.mrgfm` (this engine's format, also written next to the
    source on first use), `<prefix>.fa`, or the reference's own `<prefix>.1.ebwt` (read back into
    entry names + sequences as bowtie-inspect does);
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
    p.add_argument("--device-ingest", action="store_true",
                   help="split, trim and pack the FASTQ records on the GPU (raw text blocks are uploaded; `-ad none` / `-ad +N` "
                        "and plain four-line records only: anything else falls back to the host parser per file)")
    p.add_argument("--gpus", type=int, default=1,
                   help="annotate on N GPUs of this node: one process per GPU, the collapsed read set in N "
                        "contiguous shards, one RCCL all-reduce of the count vector (default 1)")
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


def _launch_ranks(argv, n_gpus):
    """`--gpus N`: one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE in its environment),
    started before anything in this process has touched a GPU.  The children are polled: the first
    one that fails (non-zero status, or killed by a signal) takes its siblings down with it -- they
    would otherwise sit in a collective until the RCCL watchdog fires -- and the launcher returns
    non-zero (128 + signal for a killed child)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MIRGE_AMD_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-m", "mirge_amd"] + list(argv), env=env))
    status = 0
    live = list(procs)
    while live and status == 0:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                status = 128 - rc if rc < 0 else rc
                break
    if status != 0:
        for p in live:
            p.terminate()
        deadline = time.time() + 10.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return status


def annotate_main(args, engine_factory=None, materialize=False):
    """MAIN:73-392 for `annotate`, columnar: FASTQ -> packed reads -> device collapse -> cascade +
    tally on this rank's shard of the collapsed set -> one all-reduce of the count vector -> (rank 0)
    merge / filter / tables.  No per-read Python object is built except for the reads the isomiR,
    GFF, tRF and A-to-I consumers look at (those claimed by passes 0 / 8 and 2 / 3).
    engine_factory(device_index) -> engine (default mirge_amd.engine.Engine; the multi-process CPU
    test passes an oracle-backed stand-in).  materialize: also return the whole seqDic (tests)."""
    from . import annotate, columnar, dist as mdist, ingest, report
    from .engine import CANON_PASS, ISOMIR_PASS, ReadSet
    import torch
    if engine_factory is None:
        from .engine import Engine as engine_factory
    # (RANK / WORLD_SIZE of a foreign launcher's environment do not make this a multi-rank run:
    # only the children of `--gpus N` -- or a test that says so -- take part in a process group)
    multi = os.environ.get("MIRGE_AMD_CHILD") == "1" or os.environ.get("MIRGE_AMD_DIST_BACKEND") is not None
    rank, local_rank, world = mdist.env_world() if multi else (0, 0, 1)
    if world > 1:
        mdist.init_process_group(os.environ.get("MIRGE_AMD_DIST_BACKEND", "nccl"), timeout_s=600)
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
    for kind in kinds:  # MAIN:262-267: the reference's .1.ebwt, or this engine's own formats
        p = os.path.join(index_dir, "%s_%s" % (sp, kind))
        if not any(os.path.isfile(p + ext) for ext in (".mrgfm", ".fa", ".fasta", ".1.ebwt")):
            print("The index file of %s_%s (.1.ebwt, .mrgfm or .fa) is not located at %s, please check it."
                  % (sp, kind, index_dir))
            sys.exit(1)
        prefix[kind] = p
    raw = resolve_samples(args.sampleList)
    sample_list = [os.path.basename(s)[:-3] if s.endswith(".gz") else os.path.basename(s) for s in raw]
    S = len(sample_list)
    outdir = None
    if rank == 0:
        outdir = os.path.join(os.path.abspath(args.output_dir),
                              "miRge." + time.strftime("%Y-%m-%d_%H-%M-%S", time.localtime()))
        os.makedirs(outdir)

    engine = engine_factory(local_rank if world > 1 else args.gpu)
    dev = engine.device
    spike = bool(args.spikeIn)
    files = {"mirna": prefix["mirna_" + db], "hairpin": prefix["hairpin_" + db], "mature_trna": prefix["mature_trna"],
             "pre_trna": prefix["pre_trna"], "snorna": prefix["snorna"], "rrna": prefix["rrna"],
             "ncrna_others": prefix["ncrna_others"], "mrna": prefix["mrna"]}
    if spike:
        files["spike-in"] = prefix["spike-in"]
    t_lib = time.time()
    annotate._ensure_libraries(engine, files, cache=(rank == 0))
    genome, removed_ai = None, []
    if args.a_to_i and rank == 0:  # MAIN:136-146, :277
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
    if rank == 0:
        print("Libraries resident on the device after %.2f sec" % (time.time() - t_lib))

    log_dic = {"quantStats": [], "annotStats": []}
    t0 = time.time()
    # ---- ingest (trim_file, MAIN:346-369): every rank its own files (file i -> rank i mod world) ----
    from concurrent.futures import ThreadPoolExecutor
    # (fewer files than ranks -- `--gpus 8` on ONE sample is the common case, MAIN:289-314 --: a file is read by several
    # ranks, each its own part of it: mrg_fastq_load_part)
    shares = mdist.file_shares(len(raw), world)          # per rank: [(file, part, n_parts)]
    mine = shares[rank]
    n_cpu = max(1, int(args.cpu))
    n_jobs = max(1, min(len(mine), n_cpu))

    device_ingest = bool(getattr(args, "device_ingest", False)) and hasattr(engine, "_lib")

    def load_one(share):
        i, part, n_parts = share
        t1 = time.time()
        fq = None
        if device_ingest and n_parts == 1:
            try:
                fq = ingest.load_fastq_device(engine, os.path.abspath(raw[i]), adapter=args.adapter)
            except ingest.DeviceIngestUnsupported:
                fq = None   # the host parser takes it (and words whatever is wrong with the file)
        if fq is None:
            # (every rank of the node runs its own loader: the thread budget is -cpu shared out over the ranks -- eight ranks
            # inflating one .gz with -cpu threads each were 8 x -cpu threads on one host: advisor, round 5)
            fq = ingest.load_fastq(os.path.abspath(raw[i]), adapter=args.adapter, threads=max(1, n_cpu // (n_jobs * max(1, world))), part=part,
                                   n_parts=n_parts)
        return fq, time.time() - t1

    if rank == 0:
        for name in sample_list:
            print("Performing quantitation analysis of %s..." % name)
    loaded = []
    if mine and device_ingest:
        loaded = [load_one(sh) for sh in mine]   # (one file at a time: each already fills the PCIe link)
    elif mine:
        with ThreadPoolExecutor(max_workers=n_jobs) as pool:
            loaded = list(pool.map(load_one, mine))
    quant_stats, long_counts = {}, {}
    for (i, _part, _n_parts), (fq, dt) in zip(mine, loaded):
        quant_stats[i] = {"filename": sample_list[i], "totalReads": fq["total"], "trimmedReads": fq["kept"],
                          "cpuTime-trim": dt, "cpuTime-uniq": 0.0}
        for r in fq["long_reads"]:   # beyond the one-byte length of the packed batch: collapsed here, annotated by cascade_long
            long_counts.setdefault(r, [0] * S)[i] += 1
    t1 = time.time()
    # the batch's shape must be the same on every rank: words per read, N mask or not, longest read
    W, any_n, max_len = mdist.allreduce_max([max([fq["words"].shape[0] for fq, _ in loaded] + [1]),
                                             int(any(fq["nmask"] is not None for fq, _ in loaded)),
                                             max([fq["max_len"] for fq, _ in loaded] + [0])])
    n_raw = sum(fq["packed"] for fq, _ in loaded)
    d_words = torch.zeros((W, n_raw), dtype=torch.int64, device=dev)
    d_lens = torch.empty(n_raw, dtype=torch.uint8, device=dev)
    d_nmask = torch.zeros((W, n_raw), dtype=torch.int64, device=dev) if any_n else None
    d_sample = torch.empty(n_raw, dtype=torch.int16, device=dev)
    at = 0
    for (i, _part, _n_parts), (fq, _) in zip(mine, loaded):
        m = fq["packed"]

        def on_dev(a, dt):   # a file's arrays: numpy from the host parser, device tensors from the device parser
            return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
        d_words[:fq["words"].shape[0], at:at + m] = on_dev(fq["words"], np.int64)
        d_lens[at:at + m] = on_dev(fq["lens"], np.uint8)
        if fq["nmask"] is not None:
            d_nmask[:fq["nmask"].shape[0], at:at + m] = on_dev(fq["nmask"], np.int64)
        d_sample[at:at + m] = i
        at += m
    del loaded
    if world > 1:
        # ---- partition by sequence: one all-to-all puts every copy of a sequence on one
        # rank, so the per-rank collapses below are disjoint and nothing is collapsed globally ----
        dest = mdist.sequence_destination(d_words, d_lens, world)
        got = mdist.exchange_by_destination(dest, [d_words.t().contiguous(), d_lens,
                                                   None if d_nmask is None else d_nmask.t().contiguous(), d_sample])
        d_words, d_lens, d_sample = got[0].t().contiguous(), got[1], got[3]
        d_nmask = None if got[2] is None else got[2].t().contiguous()
        del dest, got
    # ---- quantReads (QNT:3-24) on this rank's sequences ----
    if d_lens.numel():
        urs, hist = engine.collapse(d_words, d_lens, d_nmask, d_sample, S, max_len)
    else:
        urs = ReadSet.from_device(torch.empty((W, 0), dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.uint8, device=dev),
                                  torch.empty((W, 0), dtype=torch.int64, device=dev) if any_n else None,
                                  torch.empty((0, S), dtype=torch.int32, device=dev), 0, max_len or 255)
        hist = torch.zeros((256, S), dtype=torch.int64, device=dev)
    del d_words, d_lens, d_nmask, d_sample
    if hasattr(engine, "release_scratch"):
        engine.release_scratch()   # the collapse arena (40 B per raw read) is not needed again
    for q in quant_stats.values():
        q["cpuTime-uniq"] = (time.time() - t1) / max(1, len(mine))
    if world > 1:
        torch.distributed.all_reduce(hist)   # readLengthDic is a sum over reads
        boxes = mdist.gather_objects_to_rank0((quant_stats, long_counts))
        if rank == 0:
            quant_stats, long_counts = {}, {}
            for qs, lc in boxes:
                for i, q in qs.items():   # (the parts of a file several ranks read add up)
                    if i in quant_stats:
                        quant_stats[i]["totalReads"] += q["totalReads"]
                        quant_stats[i]["trimmedReads"] += q["trimmedReads"]
                        quant_stats[i]["cpuTime-trim"] = max(quant_stats[i]["cpuTime-trim"], q["cpuTime-trim"])
                    else:
                        quant_stats[i] = q
                for seq, q in lc.items():
                    row = long_counts.setdefault(seq, [0] * S)
                    for i in range(S):
                        row[i] += q[i]
    if rank == 0:
        log_dic["quantStats"] = [quant_stats[i] for i in range(len(raw))]
    shard = urs

    if rank == 0:
        print("\nPerforming annotation for all of the collasped sequences...")
    t2 = time.time()
    passes = engine.mirge_passes(spike_in=spike)
    n_pass = len(passes)
    M = engine.indexes["mirna"].n_ref
    try:
        res = engine.cascade(shard, passes)
        fused, ln = mdist.fused_buffer(engine.counts_len(M, S, n_pass), n_pass=n_pass, device=dev)
        engine.tally(shard, res, M, CANON_PASS, ISOMIR_PASS, counts=fused[:ln])
        fused[ln:] = res.pass_counts
        mdist.allreduce_counts(fused)          # the one collective of the data path
        stats = res.stats
    except Exception as e:   # RAP:661-663: message + exit status 1
        from ._native import MirgeAmdError
        if not isinstance(e, MirgeAmdError):
            raise
        print("Alignment to library %s exited with none-zero status.\n" % getattr(e, "library", "?"))
        print(str(e), file=sys.stderr)
        sys.exit(1)
    # per-read arrays go to the rank that writes the tables, nowhere else
    pass_id = mdist.gather_to_rank0(res.pass_id)
    ref_id = mdist.gather_to_rank0(res.ref_id)
    pos = mdist.gather_to_rank0(res.pos)
    mm = mdist.gather_to_rank0(res.mm)
    if world > 1:   # ... and so do the unique reads themselves (read-major for the transfer)
        g_words = mdist.gather_to_rank0(urs.words.t().contiguous())
        g_lens = mdist.gather_to_rank0(urs.lens)
        g_nmask = None if urs.nmask is None else mdist.gather_to_rank0(urs.nmask.t().contiguous())
        g_quant = mdist.gather_to_rank0(urs.quant)
    if rank != 0:
        mdist.barrier()
        return None
    if world > 1:
        h_words = np.ascontiguousarray(g_words.cpu().numpy().view(np.uint64).T)
        h_lens = g_lens.cpu().numpy()
        h_nmask = None if g_nmask is None else np.ascontiguousarray(g_nmask.cpu().numpy().view(np.uint64).T)
        h_quant = g_quant.cpu().numpy().view(np.uint32)
        del g_words, g_lens, g_nmask, g_quant
    else:
        h_words = urs.words.cpu().numpy().view(np.uint64)
        h_lens = urs.lens.cpu().numpy()
        h_nmask = None if urs.nmask is None else urs.nmask.cpu().numpy().view(np.uint64)
        h_quant = urs.quant.cpu().numpy().view(np.uint32)
    U = int(h_lens.size)
    wall = time.time() - t2
    print("All annotation cycles completed (%.2f sec).\n" % wall)

    # ---- rank 0: the M-sized tables and the output files ----
    print("Summarizing and tabulating results...")
    t3 = time.time()
    fused_h = fused.cpu().numpy()
    # ---- the reads beyond 255 nt (RAP:543-554 offers a read of any length to every pass): the same passes through
    # mrg_cascade_run_long, their counters and tallies added to the reduced vector, their rows to the tables ----
    long_seqs = list(long_counts)
    long_res = None
    if long_seqs:
        lp, lr, lo, lm, lstats = engine.cascade_long(long_seqs, passes)
        lq = np.array([long_counts[s] for s in long_seqs], dtype=np.uint32).reshape(len(long_seqs), S)
        for i, st in enumerate(lstats):
            fused_h[ln + 2 * i] += st["processed"]
            fused_h[ln + 2 * i + 1] += st["aligned"]
            stats[i]["ms"] += st["ms"]

        class _LongResult:
            pass
        lres = _LongResult()
        lres.pass_id, lres.ref_id, lres.n_pass = torch.from_numpy(lp).to(dev), torch.from_numpy(lr).to(dev), n_pass
        lrs = ReadSet(np.zeros((1, len(long_seqs)), np.uint64), np.zeros(len(long_seqs), np.uint8), None, lq, device=dev)
        fused_h[:ln] += engine.tally(lrs, lres, M, CANON_PASS, ISOMIR_PASS).cpu().numpy()
        long_res = (long_seqs, lq, lp, lr, lo, lm)
    gpu_ms = sum(s["ms"] for s in stats) or 1.0
    for i, s in enumerate(stats):   # RAP:640-705 (the reference stores wall seconds per bowtie run)
        log_dic["annotStats"].append({"cpuTime": wall * s["ms"] / gpu_ms, "readsProcessed": int(fused_h[ln + 2 * i]),
                                      "readsAligned": int(fused_h[ln + 2 * i + 1])})
    mir_dic, name_seq = {}, {}
    annotate.summarize_from_counts(fused_h[:ln], engine.indexes["mirna"].names, sample_list, log_dic, mir_dic, spike)
    annotate.miRNAmerge(merge_file, sample_list, mir_dic, mirna_fa, name_seq)
    annotate.filter(mir_dic, sample_list, log_dic, args.canoRatio)
    h_pass, h_ref, h_pos, h_mm = (t.cpu().numpy() for t in (pass_id, ref_id, pos, mm))
    hist_h = hist.cpu().numpy()
    read_len_dic = {int(L): [int(x) for x in hist_h[L]] for L in np.nonzero(hist_h.sum(axis=1))[0]}
    for seq, q in long_counts.items():
        row = read_len_dic.setdefault(len(seq), [0] * S)
        for i in range(S):
            row[i] += q[i]
    names = list(ANNOT_NAMES) + (["spike-in"] if spike else [])
    npp = columnar.names_by_pass(engine, spike)

    if args.trimmed_collapsed_fa:  # QNT:26-44: per sample, by count then sequence, descending
        from . import pack
        for i, name in enumerate(sample_list):
            idx = np.nonzero(h_quant[:, i])[0]
            seqs = pack.unpack_reads(np.ascontiguousarray(h_words[:, idx]), h_lens[idx],
                                     None if h_nmask is None else np.ascontiguousarray(h_nmask[:, idx]))
            rows = sorted(zip(h_quant[idx, i].tolist(), seqs), reverse=True)
            rows += sorted(((q[i], s) for s, q in long_counts.items() if q[i]), reverse=True)
            rows.sort(reverse=True)
            with open(os.path.join(outdir, os.path.splitext(name)[0] + ".trim.collapse.fa"), "w") as fh:
                for k, (c, s) in enumerate(rows):
                    fh.write(">seq%d_%d\n%s\n" % (k + 1, c, s))

    report.write_annotation_report_csv(os.path.join(outdir, "annotation.report.csv"), sample_list, log_dic, spike)
    columnar.write_read_tables(outdir, names, sample_list, h_words, h_lens, h_nmask, h_quant, h_pass, h_ref, npp,
                               extra=long_res)
    # the reads the remaining consumers look at: dict records only when the GFF, the tRF tables or the -ai report want them
    # (round 6: the isomiR tables alone come straight from the arrays, columnar.write_isomir_tables)
    want = {CANON_PASS, ISOMIR_PASS} | ({2, 3} if args.trf_output else set())
    long_mirna = long_res is not None and bool(np.isin(np.asarray(long_res[2]), (CANON_PASS, ISOMIR_PASS)).any())
    need_records = args.gff_output or args.trf_output or args.a_to_i or (args.diff_isomirs and long_mirna)
    sub, align = {}, {}
    if need_records:
        sub, align = columnar.read_subset(h_words, h_lens, h_nmask, h_quant, h_pass, h_ref, h_pos, h_mm, npp, want, spike)
        if long_res is not None:
            columnar.add_long_records(sub, align, long_res, npp, want, spike)
    if args.gff_output:   # RAP:609-619, :653-656
        from . import isomir
        hairpin_seqs = engine.indexes["hairpin"].name_seq_dict()
        mirna_seqs = engine.indexes["mirna"].name_seq_dict()
        for pass_index in (0, 8):
            trim = 0 if pass_index == 0 else 3  # -5 1 -3 2 shortens the aligned read
            hits = {s: (npp[pass_index][a[1]], a[2] + 1, "%dM" % (len(s) - trim))
                    for s, a in align.items() if a[0] == pass_index}
            isomir.build_isomir_content(content, hits, pass_index, pre_name, hairpin_seqs, mirna_seqs, db)
        isomir.write_isomir_gff(outdir, sample_list, content, sub, db)
    if args.trf_output:   # RAP:629-634, :657-660, :698-701; W2C:648
        from . import trf
        pre_seqs = engine.indexes["pre_trna"].name_seq_dict()
        trf.collect_trf_content(trf_content, sub, sample_list, trf_tables["trnaStruDic"], pre_seqs,
                                trf.engine_lister(engine))
        trf.write_trf_tables(outdir, sample_list, log_dic, trf_content, trf_tables, pre_seqs)
    if args.diff_isomirs and not long_mirna:
        columnar.write_isomir_tables(os.path.join(outdir, "isomirs.csv"), os.path.join(outdir, "isomirs.samples.csv"), sample_list,
                                     h_words, h_lens, h_nmask, h_quant, h_pass, h_ref, npp[CANON_PASS], log_dic)
    elif args.diff_isomirs:   # (a read beyond 255 nt claimed by a miRNA pass: the records, which hold it)
        report.write_isomir_tables(os.path.join(outdir, "isomirs.csv"), os.path.join(outdir, "isomirs.samples.csv"),
                                   sample_list, columnar.isomir_dic(sub, S), log_dic)
    report.write_counts_csv(os.path.join(outdir, "miR.Counts.csv"), sample_list, mir_dic, log_dic)
    report.write_rpm_csv(os.path.join(outdir, "miR.RPM.csv"), sample_list, mir_dic, log_dic)
    if args.a_to_i:   # W2C:1221
        from .a2i import a_to_i_report
        a_to_i_report(outdir, sample_list, log_dic, sub, mir_dic, name_seq, merged_name, removed_ai, genome)
    print("Summary Complete (%.2f sec)" % (time.time() - t3))
    print("Annotation of miRge2.0 Completed (%.2f sec)" % (time.time() - t0))
    mdist.barrier()
    out = dict(outdir=outdir, mirDic=mir_dic, logDic=log_dic, readLengthDic=read_len_dic, n_unique=U + len(long_counts))
    if materialize:
        seq_dic = columnar.full_seq_dic(h_words, h_lens, h_nmask, h_quant, h_pass, h_ref, npp, spike)
        width = 11 if spike else 10
        if long_res is not None:
            columnar.add_long_records(seq_dic, {}, long_res, npp, None, spike)
        out["seqDic"] = seq_dic
    return out


def main(argv=None):
    ap = build_parser()
    args = ap.parse_args(argv)
    if args.command != "annotate":
        ap.print_help()
        return 2
    if args.gpus > 1 and os.environ.get("MIRGE_AMD_CHILD") != "1":
        return _launch_ranks(sys.argv[1:] if argv is None else argv, args.gpus)
    annotate_main(args)
    return 0
