#!/usr/bin/env python3
"""Headline benchmark: M reads/s annotated by the full miRge cascade + tally on
MI355X, with the FM-index match kernel's roofline and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cascade|exact]
                    [--reads-per-gpu R] [--scale S]

A step = one pass of the hot path over one batch: the nine-pass cascade
(runAnnotationPipeline.py:636-705) and the count tally (summarize.py:34-66) over
the rank's shard of packed reads already resident in HBM, plus -- for N > 1 --
the single RCCL all-reduce of the fused count vector.  Synthetic libraries and
reads (mirge_amd.synth); weak scaling: every rank annotates
--reads-per-gpu reads of its own seeded shard.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["cascade", "exact", "varlen"], default="cascade",
                    help="cascade = headline (100 M x 22 nt, configs[2]/[3]); exact = configs[1]; varlen = "
                         "secondary run with lengths U{16..40} (hairpin pass, two-word reads; SURVEY.md 8d)")
    ap.add_argument("--reads-per-gpu", type=int, default=None)
    ap.add_argument("--scale", type=float, default=1.0, help="library size factor (1.0 = SURVEY 8d shapes)")
    ap.add_argument("--samples", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000,
                    help="reads of rank 0's shard the CPU port re-annotates (baseline + parity gate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wstop", type=int, default=None)
    ap.add_argument("--no-ftab", action="store_true")
    ap.add_argument("--sorted", action="store_true",
                    help="feed the reads in the order the GPU collapse emits uniques (sorted by packed key)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (repeatable)")
    ap.add_argument("--mix", action="append", default=[],
                    help="override a fraction of the read mixture, e.g. polyt=0 (experiments; not the headline workload)")
    args = ap.parse_args()

    import torch
    from mirge_amd import dist as mdist
    rank, local_rank, world = mdist.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    if world > 1:
        mdist.init_process_group("nccl")

    from mirge_amd import pack, synth
    from mirge_amd.engine import Engine, ReadSet, MIRGE_PASS_TABLE
    from mirge_amd.index import FmIndex

    n_reads = args.reads_per_gpu or {"cascade": 100_000_000, "exact": 10_000_000, "varlen": 20_000_000}[args.workload]
    keys = ["mirna"] if args.workload == "exact" else list(synth.LIB_KEYS)

    # ---- libraries + indexes (host; identical on every rank) ----
    # index construction (C++, releases the GIL) overlaps with read generation below
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.time()
    libs = synth.SynthLibraries(seed=20181, scale=args.scale)
    pool = ThreadPoolExecutor(max_workers=len(keys))
    futures = {k: pool.submit(FmIndex.build, *libs.libs[k]) for k in keys}

    # ---- this rank's shard of reads (seeded per rank), packed, moved to HBM once ----
    mix = None if args.workload == "cascade" else synth.EXACT_ONLY_MIX
    if args.mix:
        mix = dict(synth.DEFAULT_MIX if mix is None else mix)
        for kv in args.mix:
            k, v = kv.split("=")
            mix[k] = float(v)
    chunk = 10_000_000
    if args.workload == "varlen":
        words = np.empty((2, n_reads), dtype=np.uint64)
        lens = np.empty(n_reads, dtype=np.uint8)
        for lo in range(0, n_reads, chunk):
            m = min(chunk, n_reads - lo)
            words[:, lo:lo + m], lens[lo:lo + m] = synth.synth_reads_varlen(
                libs, m, seed=977 + 1000 * rank + lo // chunk)
    else:
        words = np.empty((1, n_reads), dtype=np.uint64)
        for lo in range(0, n_reads, chunk):
            m = min(chunk, n_reads - lo)
            words[0, lo:lo + m] = synth.synth_reads_packed(libs, m, seed=355 + 1000 * rank + lo // chunk, mix=mix)
        if args.sorted:
            words[0].sort()
        lens = np.full(n_reads, 22, dtype=np.uint8)
    quant = synth.synth_quant(n_reads, args.samples, seed=355 + rank)
    log(rank, "reads: %d (%s) generated+packed in %.1f s" %
        (n_reads, "16..40 nt" if args.workload == "varlen" else "22 nt", time.time() - t0))
    index = {k: f.result() for k, f in futures.items()}
    pool.shutdown()
    log(rank, "libraries + indexes (%s bp) ready after %.1f s" %
        (", ".join("%s %d" % (k, libs.total_bases(k)) for k in keys), time.time() - t0))

    eng = Engine(local_rank)
    for k in keys:
        eng.add_library(k, index[k])
    if args.wstop is not None:
        eng.set_option("wstop", args.wstop)
    if args.no_ftab:
        eng.set_option("ftab", 0)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    if args.workload in ("cascade", "varlen"):
        passes = eng.mirge_passes()
        table = MIRGE_PASS_TABLE[:9]
        canon, iso = 0, 8
    else:
        table = MIRGE_PASS_TABLE[:1]
        passes = eng.make_passes([dict(lib="mirna", min_len=0, max_len=25, seed_len=28, max_mm_seed=0,
                                       max_mm_total=2)])
        canon, iso = 0, -1
    n_pass = len(passes)
    M = index["mirna"].n_ref
    rs = ReadSet(words, lens, None, quant, device=eng.device)
    fused, ln = mdist.fused_buffer(eng.counts_len(M, args.samples, n_pass), n_pass=n_pass, device=eng.device)
    out = (torch.empty(n_reads, dtype=torch.int8, device=eng.device),
           torch.empty(n_reads, dtype=torch.int32, device=eng.device),
           torch.empty(n_reads, dtype=torch.int32, device=eng.device),
           torch.empty(n_reads, dtype=torch.uint8, device=eng.device),
           fused[ln:])

    def step():
        fused.zero_()
        res = eng.cascade(rs, passes, out=out)
        eng.tally(rs, res, M, canon, iso, counts=fused[:ln])
        mdist.allreduce_counts(fused)
        return res

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    # per-pass HIP-event times (recorded on the kernels' stream by mrg_cascade_run) of the last
    # timed step; reading them synchronises, so it is done after the timed region
    st = res.stats
    per_pass_ms = np.array([s["ms"] for s in st])
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_reads * world / (ms_per_step * 1e-3) / 1e6

    if rank != 0:
        return

    # ---- roofline of the dominant kernel (by instantiation, as rocprof names it) ----
    # Algorithmic bytes per unit (DESIGN.md "Measurement"): 16 B per read offered to a
    # pass (8 B packed read + 4 B index/count in + 4 B assignment out), 64 B per LF step
    # (two rank queries, canonical 32-byte occ block each -- SURVEY.md 8d), 8 B per
    # jump-table load, 16 B per verified candidate (8 B suffix-array row + 8 B text window).
    def alg_bytes(s):
        return 16.0 * s["processed"] + 64.0 * s["steps"] + 8.0 * s["lookups"] + 16.0 * s["candidates"]

    def survey_bytes(s):
        return 16.0 * s["processed"] + 64.0 * s["steps"]

    groups = {}
    # a pass whose length window excludes the whole batch is not launched (host length hint)
    skipped = [s["processed"] == 0 and s["lds_bytes"] == 0 and per_pass_ms[i] < 0.05 and i + 1 < len(st)
               for i, s in enumerate(st)]
    for i, s in enumerate(st):
        if skipped[i]:
            continue
        # <W, occ blocks in LDS, text in LDS, stratum-first (the 2-mismatch policy), row context
        # (libraries of >= 2^20 bases, never with the text in LDS)>
        # ..., 9-mer bitmap in LDS (libraries of at most 190 000 bases)>
        nb = index[table[i][0]].info.n_bases
        has_ctx = nb >= (1 << 20) and s["lds_mode"] in (0, 1)
        if s["lds_mode"] == 4:
            name = "mrg::fused_kernel<%d>" % rs.W
        else:
            name = "mrg::match_kernel<%d, %s, %s, %s, %s>" % (
                rs.W, {0: "false, false", 1: "true, false", 2: "true, true", 3: "false, true"}[s["lds_mode"]],
                "true" if table[i][4] == 2 else "false", "true" if has_ctx else "false",
                "true" if nb <= 190000 else "false")
        g = groups.setdefault(name, dict(ms=0.0, bytes=0.0, sbytes=0.0, launches=0, passes=[]))
        g["ms"] += per_pass_ms[i]
        g["bytes"] += alg_bytes(s)
        g["sbytes"] += survey_bytes(s)
        g["launches"] += 1 if s["group"] == i else 0
        g["passes"].append(i)
    dom_name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    achieved = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(args.workload, {}).get(dom_name)
            if ent and ent.get("reads_per_gpu") == n_reads:
                traffic = ent["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, kernel=dom_name,
                    launches_per_step=dom["launches"],
                    avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
                    algorithmic_bytes_per_launch=int(dom["bytes"] / dom["launches"]),
                    achieved_survey_8d_formula=round(dom["sbytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                    note="bytes = 16*reads + 64*LF steps + 8*jump-table loads + 16*candidates; "
                         "libraries staged in LDS serve most of them on-chip, so this is an "
                         "HBM-normalised rate, not HBM traffic (see traffic)")
    passes_report = []
    for i, s in enumerate(st):
        passes_report.append(dict(
            lib=table[i][0], ms=round(float(per_pass_ms[i]), 4), processed=s["processed"],
            aligned=s["aligned"], steps=s["steps"], candidates=s["candidates"], lookups=s["lookups"],
            lds_bytes=s["lds_bytes"],
            kernel="not launched" if skipped[i] else
                   "match_kernel<%d,%s%s>" % (rs.W, ["hbm", "blocks", "blocks+text", "text", "fused"][s["lds_mode"]],
                                             ",strata" if table[i][4] == 2 else ""),
            group=s["group"], kbits_log2=s["kbits_log2"],
            alg_gbs=round(alg_bytes(s) / max(per_pass_ms[i], 1e-9) / 1e6, 1)))

    # ---- CPU baseline: the oracle's port on a bounded sample, all host cores ----
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N=1 only
        from oracle import model
        m = min(args.cpu_sample, n_reads)
        cores = os.cpu_count() or 1
        pd = [dict(lib=keys.index(k), min_len=a, max_len=b, seed_len=s_, max_mm_seed=ms, max_mm_total=mt,
                   trim5=t5, trim3=t3, poly_t=pt) for (k, a, b, s_, ms, mt, t5, t3, pt) in table]
        views = [index[k].view() for k in keys]
        from mirge_amd.engine import DEFAULT_WSTOP
        wst = DEFAULT_WSTOP if args.wstop is None else args.wstop
        t1 = time.perf_counter()
        ref = model.fm_cascade(views, pd, words[:, :m], lens[:m], None, wstop=wst, threads=cores,
                               ftab=not args.no_ftab)
        cnt = model.tally(ref["pass_id"], ref["ref_id"], quant[:m], M, n_pass, canon, iso)
        dt = time.perf_counter() - t1
        # parity gate on the same sample: identical assignments before any number is reported
        got = [t[:m].cpu().numpy() for t in out[:4]]
        for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
            if not np.array_equal(a, ref[name]):
                raise SystemExit("PARITY FAILURE on %s: GPU and CPU port disagree" % name)
        cpu = dict(value=round(m / dt / 1e6, 4), unit="M reads/s", cores=cores, kind="port",
                   sample="first %d reads of rank 0's shard, full %d-pass cascade + tally, "
                          "oracle/fm_cpu.c with OpenMP on %d threads (%.1f s)" % (m, n_pass, cores, dt),
                   parity="assignments identical on the sample")

    line = {
        "metric": "M reads/s annotated (whole node)",
        "value": round(value, 3),
        "unit": "M reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64/u32 (2-bit packed bases, integer rank/popcount)",
        "data": "synthetic (seeded libraries + %s reads, SURVEY.md 8d)" %
                ("16..40 nt" if args.workload == "varlen" else "22 nt"),
        "config": {
            "workload": {
                "cascade": "100 M x 22 nt reads per GPU, full 9-pass cascade over 8 synthetic human-sized "
                           "libraries + isomiR/count tally (BASELINE configs[2]/[3])",
                "exact": "10 M x 22 nt unique reads per GPU, exact match vs miRNA library (BASELINE configs[1])",
                "varlen": "SECONDARY (not the headline): reads of 16..40 nt, two words per read, full 9-pass "
                          "cascade incl. the hairpin pass (SURVEY.md 8d)"}[args.workload],
            "reads_per_gpu": n_reads, "libraries_scale": args.scale, "samples": args.samples,
            "parallelism": "read shards x%d, one RCCL all-reduce of the count vector" % world,
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "passes": passes_report,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
