#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<tag>_{stats,fetch,write,sq}) into the
tracked summaries under profiles/: <tag>_kernel_stats.csv (verbatim --stats table),
<tag>_summary.md and traffic.json (HBM bytes per launch of each match-kernel
instantiation, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).

    python profiles/summarize_rocprof.py r01 [--reads-per-gpu 100000000] [--workload cascade]
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASS_NAMES = ["mirna", "hairpin", "mature_trna", "pre_trna", "snorna", "rrna", "ncrna_others", "mrna",
              "mirna (isomiR)"]


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def counters(tag, kind):
    f = newest(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, kind), "*", "*_counter_collection.csv"))
    if not f:
        return None
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "match_kernel" not in r["Kernel_Name"] and "tally_kernel" not in r["Kernel_Name"]:
            continue
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].replace("void ", "").split("(")[0])
        agg.setdefault(k, collections.OrderedDict())
        agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--reads-per-gpu", type=int, default=100_000_000)
    ap.add_argument("--workload", default="cascade")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    lines = ["# rocprofv3 summary `%s` (%s workload, %d reads per GPU)\n" % (a.tag, a.workload, a.reads_per_gpu)]

    stats = newest(os.path.join(ROOT, "gpurun_out", a.tag + "_stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(out, a.tag + "_kernel_stats.csv"))
        lines.append("## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline`\n")
        lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(stats)))[:8]:
            lines.append("| `%s` | %s | %.4f | %.3f | %s |" % (
                r["Name"].replace("void ", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e6,
                float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        bj = os.path.join(ROOT, "gpurun_out", a.tag + "_stats.json")
        if os.path.exists(bj):
            shutil.copy(bj, os.path.join(out, a.tag + "_bench_under_rocprof.json"))
        lines.append("")

    fetch, write, sq = counters(a.tag, "fetch"), counters(a.tag, "write"), counters(a.tag, "sq")
    traffic = {}
    if fetch and write:
        lines.append("## HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, one step)\n")
        lines.append("FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on gfx950 "
                     "(MI355X_MICROARCH.md, HBM): it is doubled below; gathers are uncalibrated, so the "
                     "doubled figure is an upper bound for the HBM-served passes.\n")
        lines.append("| pass | kernel | FETCH_SIZE KiB | fetch x2 GB | WRITE_SIZE GB | total GB |\n|---|---|---|---|---|---|")
        per_kernel = collections.defaultdict(list)
        fk = [k for k in fetch if "match_kernel" in k[1]]
        wk = [k for k in write if "match_kernel" in k[1]]
        for i, (kf, kw) in enumerate(zip(fk, wk)):
            f_kib = fetch[kf].get("FETCH_SIZE", 0.0)
            w_kib = write[kw].get("WRITE_SIZE", 0.0)
            fb, wb = 2 * f_kib * 1024, w_kib * 1024
            per_kernel[kf[1]].append(fb + wb)
            lines.append("| %s | `%s` | %.0f | %.3f | %.3f | %.3f |" % (
                PASS_NAMES[i % len(PASS_NAMES)], kf[1].replace("mrg::", ""), f_kib, fb / 1e9, wb / 1e9,
                (fb + wb) / 1e9))
        for kn, vals in per_kernel.items():
            traffic[kn] = dict(reads_per_gpu=a.reads_per_gpu, launches=len(vals),
                               hbm_bytes_per_launch=int(sum(vals) / len(vals)))
        lines.append("")
    if sq:
        lines.append("## SQ counters per match launch (one step)\n")
        keys = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE",
                "SQ_LDS_BANK_CONFLICT"]
        lines.append("| pass | " + " | ".join(keys) + " |\n|---|" + "---|" * len(keys))
        for i, k in enumerate([k for k in sq if "match_kernel" in k[1]]):
            lines.append("| %s | " % PASS_NAMES[i % len(PASS_NAMES)] +
                         " | ".join("%.3g" % sq[k].get(c, 0) for c in keys) + " |")
        lines.append("")
    with open(os.path.join(out, a.tag + "_summary.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if traffic:
        tp = os.path.join(out, "traffic.json")
        tj = json.load(open(tp)) if os.path.exists(tp) else {}
        tj[a.workload] = traffic
        json.dump(tj, open(tp, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
