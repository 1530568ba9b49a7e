#!/usr/bin/env python3
"""Condense the rocprofv3 CSV output of one round (gpurun_out/<tag>_{stats,fetch,write,sq,tcp})
into the tracked summaries under profiles/:

  <tag>_kernel_stats.csv   the --kernel-trace --stats table, verbatim
  <tag>_summary.md         per-kernel time, HBM traffic per launch, SQ / TCP counters
  traffic.json             HBM bytes per launch of each kernel instantiation for bench.py's
                           `roofline.traffic` (FETCH_SIZE raw and doubled, WRITE_SIZE, which applies)

Rows are keyed by KERNEL NAME and dispatch id -- never by position: a cascade step launches
match_kernel / fused_kernel instantiations whose number depends on the fusion plan.  The pass a
dispatch belongs to is taken from the bench line of the same command (`passes[].launch`).

    python profiles/summarize_rocprof.py r02 [--workload cascade]
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CASCADE_KERNELS = ("match_kernel", "fused_kernel", "stratum_kernel", "exact_dict_kernel", "seed_kernel", "pair_wave_kernel")   # one launch per entry of the bench line's plan (seed_kernel also matches wave_seed_kernel)


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def counters(tag, kind):
    """{(dispatch id, kernel name): {counter: value summed over its instances}} of our kernels."""
    f = newest(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, kind), "**", "*_counter_collection.csv"))
    if not f:
        return None
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "mrg::" not in r["Kernel_Name"]:
            continue
        k = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]))
        agg.setdefault(k, collections.OrderedDict())
        agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return collections.OrderedDict(sorted(agg.items()))


def label_dispatches(keys, bench):
    """dispatch -> 'passes 2-5 (mature_trna..rrna)' using the launch plan of the bench line: the
    cascade kernels of one step appear in launch order."""
    if not bench:
        return {}
    launches = collections.OrderedDict()
    for i, p in enumerate(bench["passes"]):
        if p.get("launch") is None:
            continue
        launches.setdefault(p["launch"], []).append((i, p["lib"], "mrg::" + p["kernel"], p.get("n_launches", 1)))
    plan = []
    for g in launches.values():   # a 2-mismatch pass split into two launches appears twice
        n = max(1, g[0][3])
        for part in range(n):
            plan.append([(x[0], x[1] + (" part %d/%d" % (part + 1, n) if n > 1 else ""), x[2]) for x in g])
    out, at = {}, 0
    for k in keys:
        if not any(name in k[1] for name in CASCADE_KERNELS):
            continue
        group = plan[at % len(plan)]
        at += 1
        if group[0][2].split("<")[0] != k[1].split("<")[0]:
            out[k] = "?"
            continue
        out[k] = "pass %s (%s)" % ("-".join(str(g[0]) for g in (group[0], group[-1])) if len(group) > 1 else group[0][0],
                                   ", ".join(g[1] for g in group))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--workload", default="cascade")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    bench = None
    bj = os.path.join(ROOT, "gpurun_out", a.tag + "_stats.json")
    if os.path.exists(bj):
        try:
            bench = json.loads(open(bj).read().strip().splitlines()[-1])
            shutil.copy(bj, os.path.join(out, a.tag + "_bench_under_rocprof.json"))
        except Exception:
            bench = None
    reads = bench["config"]["reads_per_gpu"] if bench else None
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = "?"
    meta_f = os.path.join(ROOT, "gpurun_out", a.tag + "_meta.json")
    if not os.path.exists(meta_f) and "_" in a.tag:   # r05_exact, r05_a2i: one collection run, one meta file (r05_meta.json)
        meta_f = os.path.join(ROOT, "gpurun_out", a.tag.split("_")[0] + "_meta.json")
    kernels_sha = None
    if os.path.exists(meta_f):
        mj = json.load(open(meta_f))
        head = mj.get("git_head", head)
        kernels_sha = mj.get("kernels_sha16")   # bench.kernels_sha16() of the profiled tree
    lines = ["# rocprofv3 summary `%s` (%s workload, %s reads per GPU, tree %s)\n" % (a.tag, a.workload, reads, head)]

    stats = newest(os.path.join(ROOT, "gpurun_out", a.tag + "_stats", "**", "*_kernel_stats.csv"))
    kernel_avg_ms = {}   # rocprofv3's own duration per kernel (no event bracket around it): bench.py reports it beside its HIP-event figure
    if stats:
        for r in csv.DictReader(open(stats)):
            kernel_avg_ms[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e6
        shutil.copy(stats, os.path.join(out, a.tag + "_kernel_stats.csv"))
        lines.append("## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras`\n")
        lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(stats)))[:10]:
            lines.append("| `%s` | %s | %.4f | %.3f | %s |" % (
                short(r["Name"])[:80], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6,
                r["Percentage"]))
        lines.append("")
        if bench:
            lines.append("bench line of the same command: %.3f ms per step; dominant kernel `%s`, HIP-event average "
                         "%.4f ms per launch, roofline.frac %.4f\n" % (
                             bench["ms_per_step"], bench["roofline"]["kernel"], bench["roofline"]["avg_launch_ms"],
                             bench["roofline"]["frac"]))

    fetch, write = counters(a.tag, "fetch"), counters(a.tag, "write")
    traffic = {}
    if fetch and write:
        lab = label_dispatches(list(fetch), bench)
        lines.append("## HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, one step each)\n")
        lines.append("FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reads exactly half of a wide coalesced "
                     "stream (MI355X_MICROARCH.md, HBM), so the guide doubles it; for kernels dominated by narrow "
                     "gathers that correction over-counts (round 1: doubled traffic / time exceeded the achievable "
                     "HBM rate).  Both are listed; `applies` names the one bench.py reports: doubled for launches "
                     "that stream their input (identity walk of the read arrays), raw for the gather-dominated ones.\n")
        lines.append("| dispatch | kernel | what | FETCH KiB | fetch raw GB | fetch x2 GB | WRITE GB | applies | total GB |\n|---|---|---|---|---|---|---|---|---|")
        per_kernel = collections.defaultdict(list)
        wmap = {k[1]: [] for k in write}
        for k, v in write.items():
            wmap[k[1]].append(v.get("WRITE_SIZE", 0.0))
        seen = collections.Counter()
        for k, v in fetch.items():
            if not any(name in k[1] for name in CASCADE_KERNELS) and "tally" not in k[1]:
                continue
            f_kib = v.get("FETCH_SIZE", 0.0)
            wl = wmap.get(k[1], [])
            w_kib = wl[seen[k[1]]] if seen[k[1]] < len(wl) else 0.0
            seen[k[1]] += 1
            raw, dbl, wb = f_kib * 1024, 2 * f_kib * 1024, w_kib * 1024
            streams = "pass 0" in lab.get(k, "") or "tally" in k[1]
            applies = "doubled" if streams else "raw"
            tot = (dbl if streams else raw) + wb
            per_kernel[k[1]].append(dict(raw=raw, dbl=dbl, wb=wb, tot=tot, applies=applies))
            lines.append("| %d | `%s` | %s | %.0f | %.3f | %.3f | %.3f | %s | %.3f |" % (
                k[0], k[1].replace("mrg::", "")[:60], lab.get(k, ""), f_kib, raw / 1e9, dbl / 1e9, wb / 1e9, applies, tot / 1e9))
        for kn, vals in per_kernel.items():
            n = len(vals)
            traffic[kn] = dict(reads_per_gpu=reads, launches=n, hbm_bytes_per_launch=int(sum(v["tot"] for v in vals) / n),
                               fetch_bytes_raw=int(sum(v["raw"] for v in vals) / n),
                               fetch_bytes_doubled=int(sum(v["dbl"] for v in vals) / n),
                               write_bytes=int(sum(v["wb"] for v in vals) / n),
                               applies=vals[0]["applies"] if all(v["applies"] == vals[0]["applies"] for v in vals) else "mixed")
            if kn in kernel_avg_ms:
                traffic[kn]["kernel_avg_ms"] = round(kernel_avg_ms[kn], 5)
        lines.append("")
    for kind, title, keys in (
            ("sq", "SQ counters per launch (one step)",
             ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
              "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"]),
            ("tcp", "vector L1 / L2 counters per launch (one step)",
             ["TCP_TOTAL_ACCESSES", "TCP_TCC_READ_REQ", "TCP_TCC_READ_REQ_LATENCY", "TCP_PENDING_STALL_CYCLES"]),
            ("tcc", "L2 counters per launch (one step)", ["TCC_REQ", "TCC_HIT", "TCC_MISS"])):
        c = counters(a.tag, kind)
        if not c:
            continue
        lab = label_dispatches(list(c), bench)
        lines.append("## %s\n" % title)
        lines.append("| dispatch | kernel | what | " + " | ".join(keys) + " |\n|---|---|---|" + "---|" * len(keys))
        for k, v in c.items():
            if "export_pass" in k[1]:
                continue
            lines.append("| %d | `%s` | %s | " % (k[0], k[1].replace("mrg::", "")[:60], lab.get(k, "")) +
                         " | ".join("%.3g" % v.get(cn, float("nan")) for cn in keys) + " |")
        lines.append("")
    with open(os.path.join(out, a.tag + "_summary.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if traffic:
        tp = os.path.join(out, "traffic.json")
        tj = json.load(open(tp)) if os.path.exists(tp) else {}
        tj[a.workload] = traffic
        tj["_meta"] = dict(git_head=head, tag=a.tag, kernels_sha16=kernels_sha,
                           note="collected by scripts/profile_round.sh (separate --pmc passes); see <tag>_summary.md")
        json.dump(tj, open(tp, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
