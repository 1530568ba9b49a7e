#!/usr/bin/env python3
"""Headline benchmark: M reads/s annotated by the full miRge cascade + tally on MI355X, with the
roofline of the dominant kernel, a CPU baseline and parity gates.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cascade|exact|varlen|a2i]
                    [--scaling strong|weak] [--reads R] [--scale S]

A step = one pass of the hot path over one batch already resident in HBM: the nine-pass cascade
(runAnnotationPipeline.py:636-705), the count tally (summarize.py:34-66) and -- for N > 1 -- the
single RCCL all-reduce of the fused count vector.  `--workload a2i` adds the A-to-I position
tally (writeDataToCSV.py:145-229) to the step and uses the mouse-seeded libraries.

Scaling: `strong` (default) = ONE seeded read set of --reads reads (default 100 M = BASELINE
configs[2]; with N GPUs configs[3]) cut into contiguous shards with mirge_amd.dist.shard_bounds,
every rank generating only its own shard; `weak` = every rank annotates --reads reads of its own.

Besides the contract keys the JSON line carries
  roofline      dominant kernel by time; achieved / frac = the STRICT SURVEY.md 8d reading: 16 B per
                read a launch walks (each read once per launch, however many passes the launch runs)
                + 64 B per LF step, / HIP-event time; peak 8 TB/s.  The round-2 reading (16 B per
                read offered to each pass of the launch) rides along as `per_pass_offered`
  legs          (N = 1, default workload) `repeats` and `varlen` (round 6), the `exact` (BASELINE configs[1]) and `a2i` (configs[4],
                1-GPU form) workloads, each a run of this script with its own roofline and parity
  cpu_baseline  the oracle's CPU port on a bounded sample (kind "port"), or the reference-shaped
                bowtie cascade when a real bowtie 1 is on the box (kind "reference")
  parity        what was compared with what before any number was printed
  e2e           (N = 1) the same step with the reads starting in pinned host memory and the
                assignments ending there: H2D and D2H of double-buffered chunks overlap the kernels
  collapsed     (N = 1, cascade) raw reads -> mrg_collapse_run -> cascade over the unique reads
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

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_ACHIEVABLE_GBS = 6300.0  # what a streaming kernel reaches (same guide); used for the floors only

DEFAULT_READS = {"cascade": 100_000_000, "exact": 10_000_000, "varlen": 20_000_000, "a2i": 50_000_000, "repeats": 20_000_000}


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def kernels_sha16():
    """Hash of the device sources: tells whether profiles/traffic.json was collected on this tree's kernels."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mirge_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def git_head():
    """Commit the tree was built from: `git rev-parse` where .git exists (this container), else what
    __graft_entry__.build() recorded next to the library (the GPU box gets a snapshot without .git)."""
    import subprocess
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        if out.returncode == 0 and out.stdout.strip():
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                                   text=True, timeout=20).stdout.strip() != ""
            return out.stdout.strip()[:12] + ("+dirty" if dirty else "")
    except Exception:
        pass
    try:
        info = json.load(open(os.path.join(ROOT, "mirge_amd", "lib", "build_info.json")))
        return info.get("git_head", "unknown") + ("" if info.get("kernels_sha16") == kernels_sha16() else "+changed")
    except Exception:
        return "unknown"


def survey_bytes(processed, steps):
    """16 B of streaming I/O per read offered to a pass (8 B packed read + 4 B count
    in + 4 B assignment out) + 64 B per LF step (two rank queries on the canonical 32-byte block)."""
    return 16.0 * processed + 64.0 * steps


def extended_bytes(s):
    """Secondary accounting: + 8 B per jump-table load + 16 B per verified candidate."""
    return survey_bytes(s["processed"], s["steps"]) + 8.0 * s["lookups"] + 16.0 * s["candidates"]


def kernel_name(W, table_row, s, n_bases):
    """The instantiation rocprofv3 names for the launch pass `s` ran in."""
    if s["lds_mode"] == 4:
        return "mrg::fused_kernel<%d, false>" % W   # (launch_table switches to <W, true> when a member carries pair tables)
    if s["lds_mode"] in (5, 6):
        return "mrg::stratum_kernel<%d>" % W
    if s["lds_mode"] == 7:
        # (FIRST = the pass streams the whole read set; KBITS = its library has the 9-mer bitmap)
        # (third argument, round 6: one contiguous stretch of the batch per workgroup -- variant + 16: a batch whose 4096-read
        # chunks would not go round the workgroups evenly, e.g. the 10 M reads of configs[1])
        return "mrg::exact_dict_kernel<%s, %s, %s>" % ("true" if s.get("first_pass") else "false", "true" if n_bases <= 190000 else "false",
                                                       "true" if s.get("variant", 0) & 16 else "false")
    if s["lds_mode"] == 11:
        return "mrg::pair_wave_kernel<%s>" % ("true" if s.get("variant", 0) & 8 else "false")
    if s["lds_mode"] in (8, 9):
        v, fat = s.get("variant", 0) & 3, "true" if s.get("variant", 0) & 4 else "false"
        if v:
            # (third argument, round 6: the instantiation that carries the second word of reads of 33..63 nt -- variant + 8)
            return "mrg::wave_seed_kernel<%s, %s>" % ({(8, 1): "false, 8", (9, 1): "true, 6", (8, 2): "false, 6", (9, 2): "true, 5"}[(s["lds_mode"], v)],
                                                      "true" if s.get("variant", 0) & 8 else "false")
        return "mrg::seed_kernel<%s, %s, %s>" % ("false, 8" if s["lds_mode"] == 8 else "true, 6", fat, "true" if s.get("variant", 0) & 8 else "false")
    has_ctx = n_bases >= (1 << 20) and s["lds_mode"] in (0, 1)
    return "mrg::match_kernel<%d, %s, %s, %s, %s>" % (
        W, {0: "false, false", 1: "true, false", 2: "true, true", 3: "false, true"}[s["lds_mode"]],
        "true" if table_row[4] == 2 else "false", "true" if has_ctx else "false",
        "true" if n_bases <= 190000 else "false")


def launch_table(st, per_pass_ms, table, index, W, n_reads=0):
    first_launched = next((i for i, s in enumerate(st) if s.get("n_launches", 0)), 0)
    for i, s in enumerate(st):
        s["first_pass"] = i == first_launched
    """One entry per kernel launch of the last step (a fused launch covers several passes).
    `walked` = reads on the launch's input list = the batch minus what earlier passes claimed."""
    launches = []
    claimed_before = 0
    for i, s in enumerate(st):
        claimed_here = claimed_before
        claimed_before += s["aligned"]
        skipped = s["processed"] == 0 and s["lds_bytes"] == 0 and per_pass_ms[i] < 0.05 and i + 1 < len(st) \
            and s["lds_mode"] not in (4, 8, 9)
        if skipped:
            continue
        if s["group"] != i and launches and launches[-1]["first"] == s["group"]:
            L = launches[-1]
        else:
            L = dict(first=i, passes=[], ms=0.0, processed=0, steps=0, ext=0.0, n=max(1, s.get("n_launches", 1)),
                     walked=max(0, n_reads - claimed_here),
                     kernel=kernel_name(W, table[i], s, index[table[i][0]].info.n_bases))
            launches.append(L)
        L["passes"].append(i)
        if s["lds_mode"] == 4 and s.get("pair_anchor"):
            L["kernel"] = "mrg::fused_kernel<%d, true>" % W
        L["ms"] += float(per_pass_ms[i])
        L["processed"] += s["processed"]
        L["steps"] += s["steps"]
        L["ext"] += extended_bytes(s)
    return launches



def kernel_time_frac(tpath, wl, kernel, n_reads, bytes_per_launch):
    """roofline.frac by the kernel's OWN duration as rocprofv3 measured it (profiles/traffic.json: kernel_avg_ms, kept
    with the counter passes of the same tree) -- a HIP-event bracket adds ~3 us of idle queue to a 50 us launch; null when
    the profile has no such figure or was collected on other device sources."""
    try:
        tj = json.load(open(tpath))
        ent = tj.get(wl, {}).get(kernel)
        if not ent or ent.get("reads_per_gpu") != n_reads or "kernel_avg_ms" not in ent:
            return None
        return dict(kernel_avg_ms=ent["kernel_avg_ms"], frac=round(bytes_per_launch / (ent["kernel_avg_ms"] * 1e6) / HBM_PEAK_GBS, 4),
                    source="rocprofv3 --kernel-trace --stats (profiles/traffic.json)",
                    stale=tj.get("_meta", {}).get("kernels_sha16") != kernels_sha16())
    except Exception:
        return None


def run_cold(eng, passes, words, lens, quant, M, n_pass, canon, iso, ln, ln_tally, fused, log, n_sets=6, rounds=4):
    """BASELINE configs[1] with its inputs COLD (round-5 verdict: the leg's whole working set -- 173 MB of reads, lengths,
    counts, outputs + 85 MB for the tally -- fits the 256 MiB memory-side cache and the timed steps replay the same buffers,
    so nothing showed that those bytes came from HBM).  n_sets read sets of the same size in their own allocations (the
    batch rotated by a different offset each: other addresses, other order), with their own output arrays, used in turn:
    between two uses of a set the others move (n_sets - 1) x 170 MB through the memory system.  Per-pass HIP events
    (fused_step off) give the kernel's launch duration for every step; `warm` is the same measurement replaying ONE set."""
    import torch
    from mirge_amd.engine import ReadSet
    n = len(lens)
    sets = []
    for j in range(n_sets):
        sh = (j * n) // n_sets
        w_ = np.ascontiguousarray(np.roll(words, sh, axis=1))
        rs_ = ReadSet(w_, np.roll(lens, sh), None, np.roll(quant, sh, axis=0), device=eng.device)
        sets.append((rs_, torch.empty(n, dtype=torch.int32, device=eng.device), torch.zeros_like(fused)))
    eng.set_option("fused_step", 0)

    def one(k):
        rs_, po, f = sets[k]
        f.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        res = eng.cascade_packed(rs_, passes, out=(po, f[ln:]))
        eng.tally(rs_, res, M, canon, iso, counts=f[:ln_tally])
        e1.record()
        torch.cuda.synchronize()
        return float(res.stats[0]["ms"]), e0.elapsed_time(e1)
    for k in range(n_sets):
        one(k)
    cold_k, cold_s = [], []
    for _ in range(rounds):
        for k in range(n_sets):
            a, b = one(k)
            cold_k.append(a)
            cold_s.append(b)
    warm_k, warm_s = [], []
    one(0)
    for _ in range(rounds * n_sets):
        a, b = one(0)
        warm_k.append(a)
        warm_s.append(b)
    per_set = int(sets[0][0].words.numel() * 8 + n + sets[0][0].quant.numel() * 4 + n * 4)
    del sets
    torch.cuda.empty_cache()
    alg = 16.0 * n
    med = lambda v: float(np.median(v))
    out = dict(protocol="%d read sets of %d reads in their own allocations used in turn (%.0f MB each: %.0f MB between two uses "
                        "of a set, the memory-side cache holds 256 MiB), %d rounds; kernel ms = the launch's HIP-event bracket, "
                        "step ms = events around cascade + tally" % (n_sets, n, per_set / 1e6, (n_sets - 1) * per_set / 1e6, rounds),
               cold=dict(kernel_ms=round(med(cold_k), 4), kernel_ms_min=round(min(cold_k), 4), kernel_ms_max=round(max(cold_k), 4),
                         step_ms=round(med(cold_s), 4), frac=round(alg / (med(cold_k) * 1e6) / HBM_PEAK_GBS, 4)),
               warm=dict(kernel_ms=round(med(warm_k), 4), kernel_ms_min=round(min(warm_k), 4), kernel_ms_max=round(max(warm_k), 4),
                         step_ms=round(med(warm_s), 4), frac=round(alg / (med(warm_k) * 1e6) / HBM_PEAK_GBS, 4)))
    out["meets_0.40_cold"] = bool(out["cold"]["frac"] >= 0.40)
    log(0, "cold protocol: kernel %.4f ms cold (frac %.4f) / %.4f ms warm (frac %.4f)" %
        (out["cold"]["kernel_ms"], out["cold"]["frac"], out["warm"]["kernel_ms"], out["warm"]["frac"]))
    return out

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["cascade", "exact", "varlen", "a2i", "repeats"], default="cascade",
                    help="cascade = headline (100 M x 22 nt, configs[2]/[3]); exact = configs[1]; varlen = secondary "
                         "run with lengths U{16..40}; a2i = configs[4] (mouse-seeded libraries, 50 M reads with "
                         "A->G edits, cascade + tally + A-to-I position tally); repeats = the headline's read mixture and "
                         "cascade on UNFRIENDLY libraries (poly-A/T tails, tandem repeats, paralog families, an element whose "
                         "core occurs ~10^5 times: synth.decorate_repeats) -- what the hash chains, seed buckets and the FM "
                         "fallback cost when the libraries are not i.i.d.-uniform")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--reads", type=int, default=None, help="reads of the whole job (strong) / per GPU (weak)")
    ap.add_argument("--reads-per-gpu", type=int, default=None, help="= --scaling weak --reads R")
    ap.add_argument("--scale", type=float, default=1.0, help="library size factor (1.0 = SURVEY 8d shapes)")
    ap.add_argument("--samples", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000,
                    help="reads of rank 0's shard the CPU port re-annotates (baseline + parity gate)")
    ap.add_argument("--scan-sample", type=int, default=4000,
                    help="distinct reads re-annotated by exhaustive scan of the library texts (no index at all), "
                         "split evenly over the claiming passes and the unclaimed reads")
    ap.add_argument("--bowtie-sample", type=int, default=100_000, help="reads for the bowtie probe, if bowtie exists")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the e2e and collapsed legs (and the workload legs)")
    ap.add_argument("--keep-lens", default=None, help="experiments: keep only the reads of lo,hi nt of the generated batch (N = 1)")
    ap.add_argument("--no-cold", action="store_true", help="--workload exact: skip the cache-cold protocol (rotating read sets)")
    ap.add_argument("--no-legs", action="store_true", help="skip the exact / a2i child runs of the default workload")
    ap.add_argument("--legs-reads", type=int, default=None, help="reads of each leg (default: the workloads' own sizes)")
    ap.add_argument("--wstop", type=int, default=None)
    ap.add_argument("--no-ftab", action="store_true")
    ap.add_argument("--sorted", action="store_true",
                    help="feed the reads in the order the GPU collapse emits uniques (sorted by packed key)")
    ap.add_argument("--e2e-wire", choices=["compact", "arrays"], default="compact",
                    help="upload form of the e2e leg: the compact wire form (6.5 B per 22-nt read; one-word reads) or the arrays (13 B)")
    ap.add_argument("--e2e-chunks", default="8,12,16", help="chunks of the e2e leg (a comma list: each is run, the best is reported)")
    ap.add_argument("--sort-key", choices=["word", "seed0", "seed1"], default="word",
                    help="--sorted by the packed word (the collapse's order) or by the first / second 11 bases (what a "
                         "partition by seed would give the large-library launch: an experiment)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (repeatable)")
    ap.add_argument("--big-dict", type=int, default=1,
                    help="0: no exact-match dictionary for the large libraries searched without seed mismatch (mRNA): A/B knob")
    ap.add_argument("--outputs", choices=["packed", "arrays"], default="packed",
                    help="per-read output of the timed steps: one 4-byte packed word per read (mrg_cascade_run_packed, "
                         "SURVEY.md 8d's unit) or the four arrays pass_id / ref_id / pos / mm (10 B per read)")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: the step (zeroing, cascade, tally) is captured into a hipGraph per count vector and replayed "
                         "(needs --fused-step 1: no event may live inside a captured step)")
    ap.add_argument("--fused-step", type=int, default=1,
                    help="1 (default): the timed steps run with the context option fused_step (no per-pass events, the per-pass "
                         "counters exported by the tally launch); the per-pass times of the line come from ONE extra untimed step "
                         "with the option off")
    ap.add_argument("--mix", action="append", default=[],
                    help="override a fraction of the read mixture, e.g. polyt=0 (experiments; not the headline workload)")
    args = ap.parse_args()
    if args.reads_per_gpu is not None:
        args.scaling, args.reads = "weak", args.reads_per_gpu

    import torch
    from mirge_amd import dist as mdist
    rank, local_rank, world = mdist.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # MRG_BENCH_SHARE_GPU=1: a dry run of the N > 1 path on a box with fewer GPUs than ranks (ranks
    # share devices, gloo instead of RCCL) -- exercises sharding, the all-reduce and the rank-0 line;
    # its timings mean nothing
    share = os.environ.get("MRG_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if share else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        mdist.init_process_group("gloo" if share else "nccl")

    from mirge_amd import synth
    from mirge_amd.engine import Engine, ReadSet, MIRGE_PASS_TABLE, DEFAULT_WSTOP
    from mirge_amd.index import FmIndex

    wl = args.workload
    n_job = args.reads or DEFAULT_READS[wl]
    keys = ["mirna"] if wl == "exact" else list(synth.LIB_KEYS)

    # ---- libraries + indexes (host; identical on every rank) ----
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.time()
    if wl == "a2i":
        libs = synth.SynthLibraries(seed=synth.MOUSE_SEED, scale=args.scale, shapes=synth.MOUSE_SHAPES)
    else:
        libs = synth.SynthLibraries(seed=20181, scale=args.scale, repeats=(wl == "repeats"))
    pool = ThreadPoolExecutor(max_workers=len(keys))
    futures = {k: pool.submit(FmIndex.build, *libs.libs[k]) for k in keys}

    # ---- this rank's reads ----
    mix = None
    if wl == "exact":
        mix = synth.EXACT_ONLY_MIX
    elif wl == "a2i":
        mix = synth.A2I_MIX
    if args.mix:
        mix = dict(synth.DEFAULT_MIX if mix is None else mix)
        for kv in args.mix:
            k, v = kv.split("=")
            mix[k] = float(v)
    if args.scaling == "strong":
        lo, hi = mdist.shard_bounds(n_job, rank, world)
        n_total = n_job
        seed0 = 355
    else:
        lo, hi = 0, n_job
        n_total = n_job * world
        seed0 = 355 + 1000 * rank
    n_reads = hi - lo
    words, lens, quant = synth.global_read_slice(
        libs, n_job, lo, hi, workload=("varlen" if wl == "varlen" else "cascade"),
        seed0=seed0 + (4000 if wl == "a2i" else 0), mix=mix, n_samples=args.samples)
    if args.keep_lens:
        # (experiments: the reads of one length class only, e.g. 33,40 -- what a lane of their own would see)
        lo_, hi_ = (int(x) for x in args.keep_lens.split(","))
        sel_ = (lens >= lo_) & (lens <= hi_)
        words, lens, quant = np.ascontiguousarray(words[:, sel_]), np.ascontiguousarray(lens[sel_]), np.ascontiguousarray(quant[sel_])
        n_reads = n_total = n_job = int(lens.shape[0])
    if args.sorted:
        key = words[0] if args.sort_key == "word" else (words[0] >> np.uint64(22 * int(args.sort_key[-1]))) & np.uint64((1 << 22) - 1)
        order = np.argsort(key, kind="stable")
        words, lens, quant = np.ascontiguousarray(words[:, order]), lens[order], quant[order]
    log(rank, "reads: %d of %d (%s, %s scaling) generated+packed in %.1f s" %
        (n_reads, n_total, "16..40 nt" if wl == "varlen" else "22 nt", args.scaling, time.time() - t0))
    index = {k: f.result() for k, f in futures.items()}
    pool.shutdown()
    log(rank, "libraries + indexes (%s bp) ready after %.1f s" %
        (", ".join("%s %d" % (k, libs.total_bases(k)) for k in keys), time.time() - t0))

    eng = Engine(dev_index)
    for k in keys:
        eng.add_library(k, index[k], exact_dict=None if args.big_dict else False)
    if args.wstop is not None:
        eng.set_option("wstop", args.wstop)
    if args.no_ftab:
        eng.set_option("ftab", 0)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    if wl == "exact":
        table = MIRGE_PASS_TABLE[:1]
        passes = eng.make_passes([dict(lib="mirna", min_len=0, max_len=25, seed_len=28, max_mm_seed=0, max_mm_total=2)])
        canon, iso = 0, -1
    else:
        passes = eng.mirge_passes()
        table = MIRGE_PASS_TABLE[:9]
        canon, iso = 0, 8
    n_pass = len(passes)
    M = index["mirna"].n_ref
    S = args.samples
    rs = ReadSet(words, lens, None, quant, device=eng.device)
    n_edit = eng.edit_counts_len("mirna", S) if wl == "a2i" else 0
    fused, ln = mdist.fused_buffer(eng.counts_len(M, S, n_pass) + n_edit, n_pass=n_pass, device=eng.device)
    ln_tally = ln - n_edit
    out = (torch.empty(n_reads, dtype=torch.int8, device=eng.device),
           torch.empty(n_reads, dtype=torch.int32, device=eng.device),
           torch.empty(n_reads, dtype=torch.int32, device=eng.device),
           torch.empty(n_reads, dtype=torch.uint8, device=eng.device),
           fused[ln:])

    packed_out = torch.empty(n_reads, dtype=torch.int32, device=eng.device) if args.outputs == "packed" else None

    # N > 1: the all-reduce of step k runs beside the cascade of step k + 1 (RCCL's stream; two count vectors used in
    # turn, a vector is not touched again before its own all-reduce has finished); every step's reduced vector is
    # complete when the timed region ends
    bufs = [fused, torch.zeros_like(fused)] if world > 1 else [fused]
    pending = [None] * len(bufs)
    state = dict(k=0)

    def body(f):
        f.zero_()
        if packed_out is not None:
            res = eng.cascade_packed(rs, passes, out=(packed_out, f[ln:]))
        else:
            res = eng.cascade(rs, passes, out=out[:4] + (f[ln:],))
        eng.tally(rs, res, M, canon, iso, counts=f[:ln_tally])
        if wl == "a2i":
            eng.edit_tally(rs, res, "mirna", canon, iso, counts=f[ln_tally:ln])
        return res

    graphs = [None] * len(bufs)

    def step():
        b = state["k"] % len(bufs)
        state["k"] += 1
        f = bufs[b]
        if pending[b] is not None:
            pending[b].wait()          # (the current stream waits; the host does not, on RCCL)
            pending[b] = None
        if graphs[b] is not None:
            graphs[b].replay()
            res = state["res"]
        else:
            res = body(f)
        pending[b] = mdist.allreduce_counts(f, async_op=True)
        state["last"] = f
        return res

    def drain():
        for b, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[b] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # whatever the launch plan derives lazily (concatenated small libraries, anchor-pair tables) is built
    # here, not inside a step
    eng.prepare(passes, rs.W, rs.min_len, rs.max_len)
    # the timed steps carry no HIP event and no launch but the kernels' own ("fused_step": the tally launch exports the
    # per-pass counters); the per-pass times of the line are those of one extra untimed step with the option off
    fused_step = bool(args.fused_step)
    if fused_step:
        eng.set_option("fused_step", 1)
    graph_note = None
    if args.graph and fused_step:
        try:
            state["res"] = body(bufs[0])      # (everything lazy is built, the result object of a step exists)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            for b_, f_ in enumerate(bufs):
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, stream=side):
                    body(f_)
                graphs[b_] = g_
            graph_note = "zeroing + cascade + tally of a step captured once per count vector, replayed"
        except Exception as e:   # (a runtime that cannot capture these launches: plain steps)
            graphs = [None] * len(bufs)
            graph_note = "capture failed (%r): plain launches" % (e,)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    if fused_step:
        # per-pass times, launch plan and counters: one more step, untimed, with its events
        eng.set_option("fused_step", 0)
        keep = state.get("last")
        res = body(torch.zeros_like(bufs[0]))
        torch.cuda.synchronize()
        state["last"] = keep
    fused = state.get("last", fused)   # the last step's (reduced) count vector
    reduce_check = None
    if world > 1:
        # every read falls into exactly one category: the reduced category totals sum to the reads of ALL ranks
        tot = torch.tensor([int(quant.sum())], dtype=torch.int64, device=eng.device)
        torch.distributed.all_reduce(tot)
        got_total = int(fused[2 * M * S:2 * M * S + (n_pass + 1) * S].sum().item())
        if got_total != int(tot.item()):
            raise SystemExit("PARITY FAILURE: the reduced category totals sum to %d, the ranks hold %d reads" % (got_total, int(tot.item())))
        reduce_check = "category totals of the last step's all-reduced count vector sum to the %d reads (with their counts) of all %d ranks" % (got_total, world)
    # per-pass HIP-event times (recorded on the kernels' stream by mrg_cascade_run) of the last
    # timed step; reading them synchronises, so it is done after the timed region
    st = res.stats
    per_pass_ms = np.array([s["ms"] for s in st])
    cold = None   # (after the step's own counters are read: the protocol launches cascades of its own)
    if wl == "exact" and world == 1 and not args.no_cold:
        cold = run_cold(eng, passes, words, lens, quant, M, n_pass, canon, iso, ln, ln_tally, fused, log)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = n_total / (ms_per_step * 1e-3) / 1e6

    if rank != 0:
        return

    # ---- roofline of the dominant kernel (by instantiation, as rocprofv3 names it) ----
    launches = launch_table(st, per_pass_ms, table, index, rs.W, n_reads)
    by_kernel = {}
    for L in launches:
        g = by_kernel.setdefault(L["kernel"], dict(ms=0.0, sbytes=0.0, strict=0.0, ext=0.0, launches=0, passes=[], processed=0, walked=0))
        g["ms"] += L["ms"]
        g["sbytes"] += survey_bytes(L["processed"], L["steps"])
        g["strict"] += survey_bytes(L["walked"], L["steps"])
        g["ext"] += L["ext"]
        g["launches"] += L["n"]
        g["passes"] += L["passes"]
        g["processed"] += L["processed"]
        g["walked"] += L["walked"]
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    achieved = dom["strict"] / max(dom["ms"], 1e-9) / 1e6
    achieved_offered = dom["sbytes"] / max(dom["ms"], 1e-9) / 1e6
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(wl, {}).get(dom_name)
            if ent and ent.get("reads_per_gpu") == n_reads:
                traffic = ent["hbm_bytes_per_launch"]
                meta = tj.get("_meta", {})
                traffic_src = dict(file="profiles/traffic.json", collected_at=meta.get("git_head"),
                                   kernels_sha16=meta.get("kernels_sha16"), tree_kernels_sha16=kernels_sha16(),
                                   stale=meta.get("kernels_sha16") != kernels_sha16(),
                                   fetch_bytes_raw=ent.get("fetch_bytes_raw"),
                                   fetch_bytes_doubled=ent.get("fetch_bytes_doubled"),
                                   write_bytes=ent.get("write_bytes"), applies=ent.get("applies"))
        except Exception:
            traffic = None
    whole_strict = 16.0 * n_reads + 64.0 * sum(s["steps"] for s in st)
    whole_offered = sum(survey_bytes(s["processed"], s["steps"]) for s in st)
    roofline = dict(
        bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src, kernel=dom_name,
        launches_per_step=dom["launches"], passes=dom["passes"], avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
        algorithmic_bytes_per_launch=int(dom["strict"] / dom["launches"]),
        accounting="per_read_strict",
        formula="SURVEY.md 8d, strict: 16 B x reads a launch walks (each read once per launch) + 64 B x LF steps",
        per_pass_offered=dict(
            achieved=round(achieved_offered, 1), frac=round(achieved_offered / HBM_PEAK_GBS, 4),
            algorithmic_bytes_per_launch=int(dom["sbytes"] / dom["launches"]),
            formula="round 2's reading: 16 B x reads offered to EACH cascade pass the launch runs + 64 B x LF steps"),
        compulsory_floor_ms=round(16.0 * dom["walked"] / dom["launches"] / (HBM_ACHIEVABLE_GBS * 1e6), 4),
        achieved_extended=round(dom["ext"] / max(dom["ms"], 1e-9) / 1e6, 1),
        extended_formula="per_pass_offered + 8 B per jump-table / slot load + 16 B per verified candidate (most of them "
                         "served on chip or by L2: a rate, not HBM traffic)",
        frac_kernel_time=kernel_time_frac(tpath, wl, dom_name, n_reads, dom["strict"] / dom["launches"]),
        cold=cold,
        whole_step=dict(algorithmic_bytes=int(whole_strict), achieved=round(whole_strict / (ms_per_step * 1e6), 1),
                        frac=round(whole_strict / (ms_per_step * 1e6) / HBM_PEAK_GBS, 4),
                        frac_per_pass_offered=round(whole_offered / (ms_per_step * 1e6) / HBM_PEAK_GBS, 4),
                        compulsory_floor_ms=round(16.0 * n_reads / (HBM_ACHIEVABLE_GBS * 1e6), 3)),
        per_launch=[dict(kernel=L["kernel"].replace("mrg::", ""), passes=L["passes"], ms=round(L["ms"], 4), reads_walked=L["walked"],
                         frac_strict=round(survey_bytes(L["walked"], L["steps"]) / max(L["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                         frac_per_pass_offered=round(survey_bytes(L["processed"], L["steps"]) / max(L["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4))
                    for L in launches])
    passes_report = []
    for i, s in enumerate(st):
        L = next((x for x in launches if i in x["passes"]), None)
        passes_report.append(dict(
            lib=table[i][0], ms=round(float(per_pass_ms[i]), 4), processed=s["processed"], aligned=s["aligned"],
            steps=s["steps"], candidates=s["candidates"], lookups=s["lookups"], lds_bytes=s["lds_bytes"],
            kernel="not launched" if L is None else L["kernel"].replace("mrg::", ""),
            launch=None if L is None else L["first"], n_launches=s["n_launches"], kbits_log2=s["kbits_log2"],
            compulsory_floor_ms=round(16.0 * s["processed"] / (HBM_ACHIEVABLE_GBS * 1e6), 4)))
        if s.get("ms_rest", 0.0) > 0.0:
            # a split batch (long reads / reads with N / very short reads ran their own cascade through the FM
            # kernels first): `ms`, `kernel` and the roofline describe the cascade of the one-word reads
            passes_report[-1]["ms_rest"] = round(float(s["ms_rest"]), 4)

    # ---- parity gates + CPU baseline (rank 0, N = 1 only) ----
    cpu, parity = None, {}
    if reduce_check:
        parity["all_reduce"] = reduce_check
    got = None
    if world == 1 and not (args.no_cpu_baseline and args.no_extras):
        if packed_out is not None:
            # the timed steps wrote packed words only: one more (untimed) run with the four full arrays for the
            # gates below, and the packed words must be exactly what packing those arrays gives
            from mirge_amd.engine import PACKED_POS_SAT, PACKED_REF_SAT, unpack_assignments
            fused_chk = torch.zeros_like(fused)
            res_a = eng.cascade(rs, passes, out=out[:4] + (fused_chk[ln:],))
            eng.tally(rs, res_a, M, canon, iso, counts=fused_chk[:ln_tally])
            if wl == "a2i":
                eng.edit_tally(rs, res_a, "mirna", canon, iso, counts=fused_chk[ln_tally:ln])
            torch.cuda.synchronize()
            e_pass, e_ref, e_pos, e_mm = [t.cpu().numpy() for t in out[:4]]
            u_pass, u_ref, u_pos, u_mm = unpack_assignments(packed_out.cpu().numpy())
            same = np.array_equal(u_pass, e_pass) and np.array_equal(u_ref, np.where(e_ref >= 0, np.minimum(e_ref, PACKED_REF_SAT), -1)) \
                and np.array_equal(u_pos, np.where(e_pos >= 0, np.minimum(e_pos, PACKED_POS_SAT), -1)) \
                and np.array_equal(u_mm, np.minimum(e_mm, 3)) and bool(torch.equal(fused_chk[:ln], fused[:ln]))
            if not same:
                raise SystemExit("PARITY FAILURE: the packed-output run differs from the four-array run")
            parity["packed_outputs"] = "the timed steps' packed words = the four-array run's assignments packed (entries / offsets " \
                                       "saturating at %d / %d), count vectors identical" % (PACKED_REF_SAT, PACKED_POS_SAT)

    if not args.no_cpu_baseline and world == 1:
        from oracle import model
        from mirge_amd import pack
        counts_gpu = fused[:ln].cpu().numpy().astype(np.uint64)
        got = [t.cpu().numpy() for t in out[:4]]
        pd = [dict(lib=keys.index(k), min_len=a, max_len=b, seed_len=s_, max_mm_seed=ms, max_mm_total=mt,
                   trim5=t5, trim3=t3, poly_t=pt, kbits_log2=st[i]["kbits_log2"], pair_anchor=st[i]["pair_anchor"])
              for i, (k, a, b, s_, ms, mt, t5, t3, pt) in enumerate(table)]
        views = [index[k].view() for k in keys]
        wst = DEFAULT_WSTOP if args.wstop is None else args.wstop
        m = min(args.cpu_sample, n_reads)
        cores = os.cpu_count() or 1
        t1 = time.perf_counter()
        ref = model.fm_cascade(views, pd, words[:, :m], lens[:m], None, wstop=wst, threads=cores, ftab=not args.no_ftab)
        cnt = model.tally(ref["pass_id"], ref["ref_id"], quant[:m], M, n_pass, canon, iso)
        dt = time.perf_counter() - t1
        for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
            if not np.array_equal(a[:m], ref[name]):
                raise SystemExit("PARITY FAILURE on %s: GPU and CPU port disagree" % name)
        if m == n_reads and not np.array_equal(cnt, counts_gpu[:ln_tally]):
            raise SystemExit("PARITY FAILURE: the count vector differs from the CPU port's tally")
        parity["cpu_port"] = "pass_id/ref_id/pos/mm identical on %d reads%s (oracle/fm_cpu.c, which reads the " \
                             "product's index arrays)" % (m, "; count vector identical" if m == n_reads else "")
        # -- the arrays the port and the kernels share, pinned to the library strings at full size (a failure raises)
        t1 = time.perf_counter()
        rows = sum(model.check_index(v, libs.libs[k][1], threads=cores)["rows"] for k, v in zip(keys, views))
        parity["index_check"] = "text, segments, suffix array (permutation, ascending suffixes, row fields), BWT blocks, " \
                                "jump tables and 9-mer bitmaps of all %d libraries = the FM index of their strings by " \
                                "definition (oracle/index_check.c: %d rows, %.1f s on %d threads)" % (
                                    len(keys), rows, time.perf_counter() - t1, cores)
        # -- ... and the tables the DEVICE derived from them (libtables.hip; the host functions' output is what the check
        # above pinned): every word equal
        t1 = time.perf_counter()
        bad = {k: {t: v for t, v in eng.check_tables(k).items() if v} for k in keys}
        if any(bad.values()):
            raise SystemExit("PARITY FAILURE: tables filled on the device differ from the host's: %r" % bad)
        parity["device_tables"] = "jump tables, row context, wide rows and seed buckets the device filled from rows + text " \
                                  "(csrc/libtables.hip) = the host functions' word for word, all %d libraries (%.1f s)" % (
                                      len(keys), time.perf_counter() - t1)
        cpu = dict(value=round(m / dt / 1e6, 4), unit="M reads/s", cores=cores, kind="port",
                   sample="first %d reads of rank 0's shard, full %d-pass cascade + tally, oracle/fm_cpu.c with "
                          "OpenMP on %d threads (%.1f s)" % (m, n_pass, cores, dt),
                   parity="assignments identical on the sample")
        if wl == "a2i":
            t1 = time.perf_counter()
            want = model.edit_tally(index["mirna"], got[0], got[1], got[2], words, lens, quant, canon, iso)
            if not np.array_equal(want.reshape(-1), counts_gpu[ln_tally:ln]):
                raise SystemExit("PARITY FAILURE: the A-to-I position tally differs from the oracle's")
            parity["edit_tally"] = "substitution counts per (miRNA, position, type, sample) identical on all %d " \
                                   "reads (numpy restatement, %.1f s)" % (n_reads, time.perf_counter() - t1)

        # -- independent of the index: exhaustive scan of the library TEXTS on distinct reads sampled PER
        # CLAIMING PASS (and among the unclaimed), so that every library -- also the HBM-served ones that
        # claim a few percent of the reads -- gets hundreds of positive checks at full size
        if args.scan_sample > 0:
            from oracle import cascade as ocascade
            rng = np.random.default_rng(99)
            strata = [-1] + list(range(n_pass))
            per = max(1, args.scan_sample // len(strata))
            first, by_stratum = {}, {}
            for v in strata:
                cand = np.nonzero(got[0] == v)[0]
                if cand.size == 0:
                    continue
                pick = rng.choice(cand, size=min(cand.size, 4 * per), replace=False)
                seqs = pack.unpack_reads(np.ascontiguousarray(words[:, pick]), lens[pick], None)
                k = 0
                for j, sq in zip(pick, seqs):
                    if sq not in first:
                        first[sq] = int(j)
                        k += 1
                        if k >= per:
                            break
                by_stratum[v] = k
            t1 = time.perf_counter()
            olibs = {k: model.Library(*libs.libs[k]) for k in keys}
            seq_dic = {sq: ocascade.new_seq_record(sq, 1) for sq in first}
            log_dic = {"quantStats": [{}], "annotStats": []}
            align = {}
            ocascade.run_annotation_pipeline(seq_dic, olibs, log_dic, align_dic=align, n_passes=n_pass)
            bad = 0
            for sq, j in first.items():
                a = align.get(sq)
                mine = (int(got[0][j]), int(got[1][j]), int(got[2][j]), int(got[3][j]))
                bad += mine != (tuple(a) if a is not None else (-1, -1, -1, 0))
            if bad:
                raise SystemExit("PARITY FAILURE: %d of %d sampled reads differ from the exhaustive scan" % (bad, len(first)))
            parity["exhaustive_scan"] = "claiming pass, entry, offset and mismatches identical on %d distinct reads sampled per " \
                                        "claiming pass (%s; -1 = unclaimed) against the full-size library texts " \
                                        "(oracle/bowtie_model.c: no index, %.1f s on %d threads)" % (
                                            len(first), ", ".join("%d: %d" % kv for kv in sorted(by_stratum.items())),
                                            time.perf_counter() - t1, cores)

        # -- a real bowtie 1, if this box has one: the reference's own command lines
        from oracle import bowtie_probe
        found = bowtie_probe.find_bowtie()
        if found is None:
            parity["bowtie"] = "no bowtie / bowtie-build on this box (PATH probed): aligner parity stays unpinned"
        elif wl != "exact":
            mb = min(args.bowtie_sample, n_reads)
            all_seqs = pack.unpack_reads(np.ascontiguousarray(words[:, :mb]), lens[:mb], None)
            where = {}
            for j, sq in enumerate(all_seqs):
                where.setdefault(sq, j)
            seqs = list(where)
            def read_ebwt(prefix):  # the product's `.1.ebwt` reader on what THIS box's bowtie-build just wrote
                ix = FmIndex.from_ebwt(prefix)
                return ix.names, [ix.sequence(i) for i in range(ix.n_ref)]
            bref = bowtie_probe.reference_cascade(found[0], found[1], {k: libs.libs[k] for k in keys}, seqs, threads=cores,
                                                  ebwt_reader=read_ebwt)
            bad_e = {k: v["detail"] for k, v in (bref.get("ebwt") or {}).items() if not v["ok"]}
            parity["ebwt_reader"] = ("PINNED: every index bowtie-build wrote here (%s) read back by mrg_index_build_ebwt = the "
                                     "indexed FASTA" % ", ".join(sorted(bref["ebwt"]))) if not bad_e else "MISMATCH: %s" % bad_e
            if bad_e:
                raise SystemExit("PARITY FAILURE: the .1.ebwt reader differs from bowtie-build's files: %s" % bad_e)
            names = {k: index[k].names for k in keys}
            mine_pass = np.array([got[0][where[sq]] for sq in seqs], dtype=np.int8)
            mine_name = [names[table[got[0][where[sq]]][0]][got[1][where[sq]]] if got[0][where[sq]] >= 0 else ""
                         for sq in seqs]
            mine_pos = np.array([got[2][where[sq]] for sq in seqs], dtype=np.int32)
            d = bowtie_probe.compare(bref, mine_pass, mine_name, mine_pos)
            parity["bowtie"] = "real bowtie (%s): %d distinct reads, D1 (claiming pass) disagreements %d, D2 (miRNA " \
                               "entry, tie-break) %d, D3 (other entry/offset, tie-break) %d" % (
                                   found[0], len(seqs), d["D1"], d["D2"], d["D3"])
            cpu = dict(value=round(len(seqs) / bref["seconds"] / 1e6, 4), unit="M reads/s", cores=cores, kind="reference",
                       sample="%d distinct reads through the nine bowtie command lines of runAnnotationPipeline.py:577-599 "
                              "(--threads %d) + SAM parsing in Python (%.1f s; bowtie-build %.1f s not counted)" %
                              (len(seqs), cores, bref["seconds"], bref["build_seconds"]),
                       parity="D1 %d / D2 %d / D3 %d" % (d["D1"], d["D2"], d["D3"]), port=cpu)
            if d["D1"]:
                raise SystemExit("PARITY FAILURE vs bowtie: %d reads claimed by a different pass" % d["D1"])

    # ---- extras (N = 1): end-to-end with PCIe, and the collapsed pipeline ----
    extras = {}
    if world == 1 and not args.no_extras:
        try:
            runs_ = [run_e2e(eng, passes, words, lens, quant, M, n_pass, canon, iso, got, log, wire=args.e2e_wire, n_chunks=int(k))
                     for k in args.e2e_chunks.split(",")]
            extras["e2e"] = min(runs_, key=lambda r: r["ms_per_step"])
            if len(runs_) > 1:
                extras["e2e"]["chunk_sweep"] = {str(r["chunks"]): r["ms_per_step"] for r in runs_}
        except SystemExit:
            raise
        except Exception as e:  # the headline must not die with an optional leg
            extras["e2e"] = dict(error=repr(e))
        if wl == "cascade":
            try:
                extras["collapsed"] = run_collapsed(eng, passes, rs, out, M, n_pass, canon, iso, log)
            except SystemExit:
                raise
            except Exception as e:
                extras["collapsed"] = dict(error=repr(e))

    # ---- legs (N = 1, headline workload): the other single-GPU configurations of BASELINE.json, each a
    # child run of this script with its own parity gates, roofline and CPU baseline ----
    legs = {}
    if wl == "cascade" and world == 1 and not args.no_extras and not args.no_legs:
        import subprocess
        del rs
        torch.cuda.empty_cache()
        # (round 6: `repeats` -- libraries with interspersed elements, poly-A, tandem motifs -- and `varlen` -- reads of
        # 16..40 nt, the hairpin pass populated -- ride along too: the driver's one command times the unfriendly inputs)
        for leg, leg_args in (("exact", ["--steps", "20", "--warmup", "2"]), ("a2i", ["--steps", "5", "--warmup", "1"]),
                              ("repeats", ["--steps", "5", "--warmup", "1"]), ("varlen", ["--steps", "5", "--warmup", "1"])):
            t1 = time.perf_counter()
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", leg, "--no-extras", "--no-legs",
                   "--scale", str(args.scale)] + leg_args
            for kv in args.opt:
                cmd += ["--opt", kv]
            if args.legs_reads:
                cmd += ["--reads", str(args.legs_reads), "--cpu-sample", str(args.legs_reads), "--scan-sample", str(args.scan_sample)]
            try:
                cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            except subprocess.TimeoutExpired:
                legs[leg] = dict(error="timed out")
                continue
            if cp.returncode != 0:
                tail = (cp.stderr or "").strip().splitlines()[-1:] or ["?"]
                if "PARITY FAILURE" in (cp.stderr or ""):
                    raise SystemExit("leg %s: %s" % (leg, tail[0]))
                legs[leg] = dict(error=tail[0])
                continue
            d = json.loads(cp.stdout.strip().splitlines()[-1])
            legs[leg] = {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "cpu_baseline", "parity")}
            for k in ("split_batch", "repeats"):
                if k in d:
                    legs[leg][k] = d[k]
            legs[leg]["passes"] = [dict(lib=p_["lib"], ms=p_["ms"], kernel=p_["kernel"], processed=p_["processed"], aligned=p_["aligned"])
                                   for p_ in d["passes"]]
            legs[leg]["wall_s"] = round(time.perf_counter() - t1, 1)
            log(rank, "leg %s: %.3f ms per step, roofline.frac %.4f (%.0f s)" % (leg, d["ms_per_step"], d["roofline"]["frac"],
                                                                               time.perf_counter() - t1))

    what = {
        "cascade": "%d M x 22 nt reads, full 9-pass cascade over 8 synthetic human-sized libraries + isomiR/count tally "
                   "(BASELINE configs[2]%s)" % (n_total // 1_000_000,
                                                "; sharded over %d GPUs = configs[3]" % world if world > 1 else ""),
        "exact": "%d M x 22 nt unique reads, exact match vs miRNA library (BASELINE configs[1])" % (n_total // 1_000_000),
        "varlen": "SECONDARY (not the headline): reads of 16..40 nt, two words per read, full 9-pass cascade incl. "
                  "the hairpin pass (SURVEY.md 8d)",
        "repeats": "SECONDARY (not the headline): %d M x 22 nt reads of the headline's mixture, full 9-pass cascade + tally over "
                   "libraries with poly-A/T tails, tandem repeats, paralog families and a 10^5-copy element (synth.decorate_repeats)"
                   % (n_total // 1_000_000),
        "a2i": "%d M x 22 nt reads with seeded A->G edits, mouse-seeded libraries, 9-pass cascade (1-mismatch seed "
               "search) + count tally + A-to-I position tally (BASELINE configs[4])" % (n_total // 1_000_000)}[wl]
    line = {
        "metric": "M reads/s annotated (whole node)",
        "value": round(value, 3),
        "unit": "M reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u64/u32 (2-bit packed bases, integer rank/popcount)",
        "data": "synthetic (seeded libraries + %s reads, SURVEY.md 8d)" % ("16..40 nt" if wl == "varlen" else "22 nt"),
        "config": {
            "workload": what, "reads_total": n_total, "reads_per_gpu": n_reads, "libraries_scale": args.scale,
            "samples": S,
            "step": ("fused_step: no per-pass event and no export launch inside the timed steps (per-pass ms from one extra untimed step)"
                     if fused_step else "per-pass events recorded inside every timed step") +
                    ("; " + graph_note if graph_note else ""),
            "outputs": ("one 4-byte packed assignment per read (mrg_cascade_run_packed)" if args.outputs == "packed" else
                        "pass_id, ref_id, pos, mm arrays (10 B per read)"),
            "parallelism": ("one read set in %d contiguous shards, libraries replicated, one RCCL all-reduce of the "
                            "fused count vector per step" % world) if args.scaling == "strong" else
                           ("%d independent read sets (weak scaling), one RCCL all-reduce of the count vector" % world),
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity": parity,
        "passes": passes_report,
        "split_batch": None if not any("ms_rest" in p_ for p_ in passes_report) else dict(
            rest_ms=round(sum(p_.get("ms_rest", 0.0) for p_ in passes_report), 4),
            one_word_ms=round(sum(p_["ms"] for p_ in passes_report), 4),
            note="the batch was split on the device: reads of 20..32 nt without N ran the cascade through the dictionary kernels "
                 "(passes[].ms, kernel, roofline), the rest through the FM kernels first (passes[].ms_rest); counters are the sums"),
    }
    if wl == "repeats":
        line["repeats"] = repeats_report(libs, index, eng, st, n_reads)
    line.update(extras)
    # SURVEY.md 8d's throughput region (H2D of the packed reads + counts -> cascade -> tally -> D2H) beside the resident
    # `value`: the task contract keeps the PCIe-inclusive rate out of `value`, the reader should still see it first
    if isinstance(extras.get("e2e"), dict) and "value" in extras["e2e"]:
        line["value_e2e"] = extras["e2e"]["value"]
        line["value_e2e_note"] = ("M reads/s with the reads starting in pinned HOST memory and the assignments returning there "
                                  "(SURVEY 8d's timed region; PCIe-bound) -- `value` is the rate with the batch resident in HBM")
    line["git_head"] = git_head()
    line["kernels_sha16"] = kernels_sha16()
    if legs:
        line["legs"] = legs
    print(json.dumps(line))


def run_e2e(eng, passes, words, lens, quant, M, n_pass, canon, iso, expect, log, n_chunks=16, reps=3, n_bufs=3, wire="compact"):
    """SURVEY.md 8d "Timed region", throughput figure: H2D of packed reads + counts -> cascade ->
    tally -> D2H of the assignment arrays and the count vector.  The host side is pinned; the read
    set goes through in `n_chunks` chunks on three streams (copy in / compute / copy out) with two
    device buffers, so PCIe transfers overlap the kernels.
    wire = "compact": the host-resident collapsed set travels in the compact wire form (mrg_expand_compact:
    reads grouped by length as a bit stream of 2 L bits each, one-byte counts + escapes: 6.5 B per 22-nt read
    instead of 13) when its reads allow (one word, no N); the grouping and bit packing are the host
    collapse's output format, prepared before the timed region like the arrays are."""
    import torch
    from mirge_amd import pack
    from mirge_amd.engine import ReadSet
    dev = eng.device
    W, n = words.shape
    S = quant.shape[1]
    t0 = time.perf_counter()
    compact = wire == "compact" and W == 1 and n > 0 and int(lens.max()) <= 32
    if compact:
        order = None if not np.any(lens[1:] < lens[:-1]) else np.argsort(lens, kind="stable")
        if order is not None:
            words, lens, quant = np.ascontiguousarray(words[:, order]), lens[order], quant[order]
            if expect is not None:
                expect = tuple(a[order] for a in expect)
    h_words = torch.from_numpy(words.view(np.int64)).pin_memory()
    h_lens = torch.from_numpy(lens).pin_memory()
    h_quant = torch.from_numpy(quant.view(np.int32)).pin_memory()
    h_packed = torch.empty(n, dtype=torch.int32).pin_memory()   # one 4-byte word per read (mrg_pack_assignments)
    ln = eng.counts_len(M, S, n_pass)
    h_counts = torch.empty(ln, dtype=torch.int64).pin_memory()
    pin_s = time.perf_counter() - t0
    n_chunks = max(1, min(n_chunks, n // 1024 or 1))
    cuts = [(n * c // n_chunks) & ~3 for c in range(n_chunks)] + [n]
    bounds = [(cuts[c], cuts[c + 1]) for c in range(n_chunks)]
    cap = max(b - a for a, b in bounds)
    if compact:
        # one bit stream per chunk (already grouped by length: no permutation), back to back in one pinned buffer
        streams = [pack.compact_read_set(words[:, a:e], lens[a:e])["bits"] for a, e in bounds]
        s_off = np.concatenate(([0], np.cumsum([len(x) for x in streams]))).astype(np.int64)
        h_bits = torch.from_numpy(np.concatenate(streams).view(np.int64)).pin_memory()
        bits_cap = max(len(x) for x in streams)
        del streams
        q8, _ = pack.compact_counts(quant)
        h_q8 = torch.from_numpy(q8).pin_memory()
        c_runs = [pack.compact_runs(lens[a:e]) for a, e in bounds]
        c_esc = []
        for a, e in bounds:
            _, esc = pack.compact_counts(quant[a:e])
            c_esc.append(torch.from_numpy(esc.view(np.int32)).pin_memory())
        esc_cap = max(1, max(int(t.shape[0]) for t in c_esc))
        esc_total = sum(int(t.shape[0]) for t in c_esc)
    bufs = []
    for _ in range(n_bufs):
        bufs.append(dict(
            words=torch.empty((W, cap), dtype=torch.int64, device=dev), lens=torch.empty(cap, dtype=torch.uint8, device=dev),
            quant=torch.empty((cap, S), dtype=torch.int32, device=dev),
            out=(torch.empty(cap, dtype=torch.int8, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                 torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.uint8, device=dev)),
            packed=torch.empty(cap, dtype=torch.int32, device=dev),
            bits=torch.empty(bits_cap, dtype=torch.int64, device=dev) if compact else None,
            q8=torch.empty((cap, S), dtype=torch.uint8, device=dev) if compact else None,
            esc=torch.empty((esc_cap, 2), dtype=torch.int32, device=dev) if compact else None,
            ev_in=torch.cuda.Event(), ev_done=torch.cuda.Event(), ev_free=torch.cuda.Event()))
    counts = torch.zeros(ln, dtype=torch.int64, device=dev)
    pc = torch.zeros(2 * n_pass, dtype=torch.int64, device=dev)
    # HIP streams share a few hardware queues: which stream pair really overlaps H2D, D2H and the
    # kernels differs from box to box (scripts/pcie_overlap.py), so a few assignments are tried once
    pool = [torch.cuda.Stream(dev) for _ in range(6)]
    cur = torch.cuda.current_stream(dev)
    candidates = [(pool[0], pool[2], cur), (pool[0], pool[2], pool[4]), (pool[1], pool[3], pool[5]), (pool[0], pool[1], cur)]
    s_in, s_out, s_comp = candidates[0]
    min_len, max_len = int(lens.min()), int(lens.max())
    # the length array always travels (13 B per read up): the kernels read it whatever the hints say
    send_lens = True

    def one_pass():
        with torch.cuda.stream(s_comp):
            _one_pass()
        torch.cuda.synchronize()

    host_s = dict(copy=0.0, expand=0.0, cascade=0.0, tally=0.0, out=0.0)
    send_back = [True]   # False: only the count vector comes back (callers that write no per-read table)

    def _one_pass():
        tp = time.perf_counter
        counts.zero_()
        for b in bufs:
            b["ev_free"].record(s_comp)
        for c, (a, e) in enumerate(bounds):
            b = bufs[c % n_bufs]
            m = e - a
            t_a = tp()
            with torch.cuda.stream(s_in):
                s_in.wait_event(b["ev_free"])
                if compact:
                    nw = int(s_off[c + 1] - s_off[c])
                    b["bits"][:nw].copy_(h_bits[s_off[c]:s_off[c + 1]], non_blocking=True)
                    b["q8"][:m].copy_(h_q8[a:e], non_blocking=True)
                    k = int(c_esc[c].shape[0])
                    if k:
                        b["esc"][:k].copy_(c_esc[c], non_blocking=True)
                else:
                    for w in range(W):
                        b["words"][w, :m].copy_(h_words[w, a:e], non_blocking=True)
                    if send_lens:
                        b["lens"][:m].copy_(h_lens[a:e], non_blocking=True)
                    b["quant"][:m].copy_(h_quant[a:e], non_blocking=True)
                b["ev_in"].record(s_in)
            s_comp.wait_event(b["ev_in"])
            t_b = tp()
            if compact:
                rsc = eng.expand_compact(b["bits"][:int(s_off[c + 1] - s_off[c])], c_runs[c], b["q8"][:m], b["esc"][:int(c_esc[c].shape[0])], n_samples=S,
                                         out=(b["words"][:, :m], b["lens"][:m], b["quant"][:m]))
            else:
                # the SoA stride of the cascade is the row length of the words array it is handed
                wv = b["words"] if m == cap else b["words"][:, :m].contiguous()
                rsc = ReadSet.from_device(wv, b["lens"][:m], None, b["quant"][:m], min_len, max_len)
            t_c = tp()
            res = eng.cascade_packed(rsc, passes, out=(b["packed"][:m], pc))
            t_d = tp()
            eng.tally(rsc, res, M, canon, iso, counts=counts)
            b["ev_done"].record(s_comp)
            t_e = tp()
            with torch.cuda.stream(s_out):
                s_out.wait_event(b["ev_done"])
                if send_back[0]:
                    h_packed[a:e].copy_(b["packed"][:m], non_blocking=True)
                b["ev_free"].record(s_out)
            t_f = tp()
            for key, dt in (("copy", t_b - t_a), ("expand", t_c - t_b), ("cascade", t_d - t_c), ("tally", t_e - t_d), ("out", t_f - t_e)):
                host_s[key] += dt
        with torch.cuda.stream(s_out):
            s_out.wait_event(bufs[(len(bounds) - 1) % n_bufs]["ev_done"])
            h_counts.copy_(counts, non_blocking=True)

    one_pass()  # warm-up (also the run whose output is checked)
    trial = []
    for cand in candidates:
        s_in, s_out, s_comp = cand
        one_pass()
        t0 = time.perf_counter()
        one_pass()
        trial.append(time.perf_counter() - t0)
    s_in, s_out, s_comp = candidates[int(np.argmin(trial))]
    ok = None
    if expect is not None:
        from mirge_amd.engine import unpack_assignments, PACKED_REF_SAT, PACKED_POS_SAT
        u_pass, u_ref, u_pos, u_mm = unpack_assignments(h_packed.numpy())
        e_pass, e_ref, e_pos, e_mm = expect
        ok = np.array_equal(u_pass, e_pass) and np.array_equal(u_ref, np.where(e_ref >= 0, np.minimum(e_ref, PACKED_REF_SAT), -1)) \
            and np.array_equal(u_pos, np.where(e_pos >= 0, np.minimum(e_pos, PACKED_POS_SAT), -1)) \
            and np.array_equal(u_mm, np.minimum(e_mm, 3))
        if not ok:
            raise SystemExit("PARITY FAILURE: chunked end-to-end run differs from the resident run")
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    send_back[0] = False
    one_pass()
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    ms_counts = (time.perf_counter() - t0) * 1e3 / reps
    send_back[0] = True
    for key in host_s:
        host_s[key] = 0.0
    one_pass()
    log(0, "e2e: host time issuing one pass of %d chunks: %s ms" % (n_chunks, ", ".join("%s %.2f" % (k, v * 1e3) for k, v in host_s.items())))
    # the two transfers alone, for the record
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c, (a, e) in enumerate(bounds):
        b = bufs[c % n_bufs]
        if compact:
            b["bits"][:int(s_off[c + 1] - s_off[c])].copy_(h_bits[s_off[c]:s_off[c + 1]], non_blocking=True)
            b["q8"][:e - a].copy_(h_q8[a:e], non_blocking=True)
            if int(c_esc[c].shape[0]):
                b["esc"][:int(c_esc[c].shape[0])].copy_(c_esc[c], non_blocking=True)
            continue
        for w in range(W):
            b["words"][w, :e - a].copy_(h_words[w, a:e], non_blocking=True)
        if send_lens:
            b["lens"][:e - a].copy_(h_lens[a:e], non_blocking=True)
        b["quant"][:e - a].copy_(h_quant[a:e], non_blocking=True)
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for c, (a, e) in enumerate(bounds):
        b = bufs[c % n_bufs]
        h_packed[a:e].copy_(b["packed"][:e - a], non_blocking=True)
    torch.cuda.synchronize()
    d2h_ms = (time.perf_counter() - t0) * 1e3
    h2d_bytes = n * (8 * W + (1 if send_lens else 0) + 4 * S)
    if compact:
        h2d_bytes = 8 * int(s_off[-1]) + n * S + 8 * esc_total
    d2h_bytes = n * 4
    log(0, "e2e: %.2f ms per %d reads (H2D alone %.2f ms, D2H alone %.2f ms; pinning the host arrays took %.1f s)" %
        (ms, n, h2d_ms, d2h_ms, pin_s))
    return dict(ms_per_step=round(ms, 3), value=round(n / ms / 1e3, 3), unit="M reads/s", chunks=n_chunks,
                ms_per_step_counts_only=round(ms_counts, 3),   # the same without the D2H of the per-read words
                h2d_ms=round(h2d_ms, 3), d2h_ms=round(d2h_ms, 3), h2d_bytes=h2d_bytes, d2h_bytes=d2h_bytes,
                h2d_gbs=round(h2d_bytes / h2d_ms / 1e6, 1), d2h_gbs=round(d2h_bytes / d2h_ms / 1e6, 1),
                parity=None if ok is None else "packed assignments (pass, mismatches, entry, offset; the last two saturating at "
                                               "262143 / 255) identical to the resident run's",
                note="SURVEY.md 8d timed region: pinned host arrays -> H2D (packed reads + counts%s) -> "
                     "cascade (mrg_cascade_run_packed: 4-byte packed assignment per read) -> tally -> D2H (packed words, count "
                     "vector); %d chunks through %d device buffers on three HIP streams (copy in / compute / copy out); "
                     "H2D-bound: %s, 4 B down"
                     % (" + lengths" if send_lens else "; one read length: no length array", n_chunks, n_bufs,
                        ("compact wire form (mrg_expand_compact: reads in length groups as a bit stream of 2 L bits each, one-byte counts + %d escapes), "
                         "%.2f B per read up" % (esc_total, h2d_bytes / max(n, 1))) if compact else "13 B per read up"),
                wire="compact" if compact else "arrays")


def repeats_report(libs, index, eng, st, n_reads):
    """What the unfriendly libraries did to the structures the dictionary kernels answer from: positions an exact-match
    dictionary left to the FM index because their home slot's chain overflowed (reads whose key is one of those take
    fm_exact_fallback: a jump-table interval verified row by row), 11-mers whose seed bucket overflowed (more than 8
    rows: the wave kernel then asks the jump table), the widest 11-mer interval, and per pass the candidates (slots /
    rows compared) per lookup -- 1-3 on i.i.d. libraries."""
    out = dict(decorations=libs.repeat_stats, libraries={}, passes=[])
    for k in index:
        d = {}
        try:
            nk, nov = eng.library_dict_stats(k)
            if nk or nov:
                d.update(dict_positions=nk, dict_homes_overflowed=nov)
        except Exception as e:   # (not fatal for a report)
            d["dict_stats_error"] = repr(e)
        v = index[k].view()
        ks = v["ftab_ks"]
        if 11 in ks:
            # the jump tables lie back to back, largest k first, 4^k + 1 words each
            base = sum((4 ** kk + 1) for kk in ks[:ks.index(11)] if kk)
            tab = np.asarray(v["ftab"][base:base + 4 ** 11 + 1]).astype(np.int64)
            w = np.diff(tab)
            d.update(kmers11_present=int((w > 0).sum()), kmers11_bucket_overflow=int((w > 8).sum()), widest_11mer_rows=int(w.max()))
        out["libraries"][k] = d
    for i, s in enumerate(st):
        out["passes"].append(dict(i=i, lookups=int(s["lookups"]), candidates=int(s["candidates"]),
                                  candidates_per_lookup=round(s["candidates"] / max(s["lookups"], 1), 2)))
    return out


def run_collapsed(eng, passes, rs, out, M, n_pass, canon, iso, log, reps=3):
    """The reference's own order of work (quantReads.py:3-24 before runAnnotationPipeline): the
    resident records are treated as RAW reads, collapsed on the GPU (mrg_collapse_run) and the
    cascade + tally run over the unique reads with their multiplicities."""
    import torch
    from mirge_amd.engine import CascadeResult, ReadSet
    dev = eng.device
    n = rs.n
    max_len = rs.max_len

    def one():
        urs, _hist = eng.collapse(rs.words, rs.lens, None, None, 1, max_len)
        urs.min_len = rs.min_len
        counts = torch.zeros(eng.counts_len(M, 1, n_pass), dtype=torch.int64, device=dev)
        res = eng.cascade(urs, passes)
        eng.tally(urs, res, M, canon, iso, counts=counts)
        return urs, res, counts

    torch.cuda.synchronize()
    urs, res, counts = one()
    torch.cuda.synchronize()
    # parity: a unique read's assignment is the one its records got in the resident run, and the
    # multiplicity-weighted tally equals the record-level tally with every record counting once
    ok = None
    if rs.W == 1 and rs.min_len == rs.max_len:
        rng = np.random.default_rng(5)
        pick_t = torch.from_numpy(np.sort(rng.choice(n, size=min(n, 200_000), replace=False))).to(dev)
        w_rec = rs.words[0, pick_t]
        uw = urs.words[0]
        where = torch.searchsorted(uw, w_rec)
        ok = bool(torch.equal(uw[where], w_rec))
        for a_u, a_r in zip((res.pass_id, res.ref_id, res.pos, res.mm), out[:4]):
            ok = ok and bool(torch.equal(a_u[where], a_r[pick_t]))
        ones = torch.ones((n, 1), dtype=torch.int32, device=dev)
        rec = CascadeResult(out[0], out[1], out[2], out[3], None, eng, n_pass)
        c_rec = eng.tally(ReadSet.from_device(rs.words, rs.lens, None, ones), rec, M, canon, iso)
        k = 2 * M + n_pass + 1   # everything but trimmedUniq, which counts records there and uniques here
        ok = ok and bool(torch.equal(c_rec[:k], counts[:k])) and int(counts[k]) == urs.n
        if not ok:
            raise SystemExit("PARITY FAILURE: collapsed pipeline differs from the record-level run")
    ts = []
    # (one set of output buffers for the timed repetitions: n-sized allocations are not the collapse)
    bufs = (torch.empty((rs.W, n), dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev), None,
            torch.empty((n, 1), dtype=torch.int32, device=dev))
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        urs2, _h = eng.collapse(rs.words, rs.lens, None, None, 1, max_len, out=bufs)   # synchronises (n_unique)
        urs2.min_len = rs.min_len
        t_c = time.perf_counter() - t0
        counts2 = torch.zeros(eng.counts_len(M, 1, n_pass), dtype=torch.int64, device=dev)
        res2 = eng.cascade(urs2, passes)
        eng.tally(urs2, res2, M, canon, iso, counts=counts2)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0, t_c))
    total, t_c = min(ts)
    log(0, "collapsed: %d raw -> %d unique reads; collapse %.2f ms + cascade/tally %.2f ms" %
        (n, urs.n, t_c * 1e3, (total - t_c) * 1e3))
    return dict(raw_reads=n, unique_reads=urs.n, collapse_ms=round(t_c * 1e3, 3),
                cascade_tally_ms=round((total - t_c) * 1e3, 3), ms_per_step=round(total * 1e3, 3),
                value=round(n / total / 1e6, 3), unit="M raw reads/s",
                parity=None if ok is None else "unique reads carry the assignment of their records (200 000 sampled); "
                                               "multiplicity-weighted count vector identical to the record-level tally",
                note="quantReads.py:3-24 then the cascade: what the reference does with 100 M raw reads")


if __name__ == "__main__":
    main()
