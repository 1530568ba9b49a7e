#!/usr/bin/env python3
"""profiles/<tag>_genome_summary.md: the kernels of one genome part (scripts/genome_scale_check.py: 300 Mbp, 4 M query rows,
`-n 1 -a -3 2` then `-n 0 -a -3 2`) by time (rocprofv3 --kernel-trace --stats) and by HBM bytes (separate FETCH_SIZE /
WRITE_SIZE passes), from gpurun_out/<tag>_genome_{stats,fetch,write}.

    python profiles/summarize_genome.py r06
"""
import csv
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import ROOT, counters, newest, short


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    out = os.path.join(ROOT, "profiles")
    lines = ["# rocprofv3 summary `%s_genome`: one 300 Mbp genome part, 4 M query rows (2 M reads x 2 strands, 20 nt)\n" % tag]
    part = os.path.join(ROOT, "gpurun_out", tag + "_genome_part.json")
    if os.path.exists(part):
        for ln in open(part):
            if ln.startswith("{"):
                lines.append("script line (under the profiler): `%s`\n" % ln.strip()[:900])
    stats = newest(os.path.join(ROOT, "gpurun_out", tag + "_genome_stats", "**", "*_kernel_stats.csv"))
    avg = {}
    if stats:
        shutil.copy(stats, os.path.join(out, tag + "_genome_kernel_stats.csv"))
        lines.append("## `rocprofv3 --kernel-trace --stats -- python3 scripts/genome_scale_check.py --reads 2000000`\n")
        lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(stats)))[:8]:
            lines.append("| `%s` | %s | %.4f | %.3f | %s |" % (short(r["Name"])[:80], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                            float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
            avg[short(r["Name"])] = float(r["AverageNs"]) / 1e6
        lines.append("")
    fetch, write = counters(tag + "_genome", "fetch"), counters(tag + "_genome", "write")
    if fetch and write:
        lines.append("## HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; KiB counters; FETCH raw: these "
                     "kernels gather 8- and 12-byte items from random 64-byte lines, the guide's doubling is for wide coalesced streams)\n")
        lines.append("| dispatch | kernel | FETCH raw GB | WRITE GB | algorithmic MB (16 B x 4 M rows) | traffic / algorithmic | avg ms | raw GB/s |\n|---|---|---|---|---|---|---|---|")
        wl = {}
        for k, v in write.items():
            wl.setdefault(k[1], []).append(v.get("WRITE_SIZE", 0.0))
        seen = {}
        for k, v in fetch.items():
            if "count" not in k[1]:
                continue
            i = seen.get(k[1], 0)
            seen[k[1]] = i + 1
            w = wl.get(k[1], [])
            fb, wb = v.get("FETCH_SIZE", 0.0) * 1024, (w[i] if i < len(w) else 0.0) * 1024
            ms = avg.get(k[1])
            lines.append("| %d | `%s` | %.3f | %.3f | 64 | %.1f | %s | %s |" % (
                k[0], k[1].replace("mrg::", "")[:60], fb / 1e9, wb / 1e9, (fb + wb) / 64e6,
                "%.3f" % ms if ms else "", "%.0f" % ((fb + wb) / (ms * 1e6)) if ms else ""))
        lines.append("")
    with open(os.path.join(out, tag + "_genome_summary.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
