"""Timeline of the kernels of the LAST repetition in a rocprofv3 kernel trace: start offset, duration, gap in front.
usage: python scripts/timeline.py <dir with *_kernel_trace.csv> [first-kernel-substring]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "prepass"
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[-1]
# (include the fills in front of it)
while i0 > 0 and "fillBuffer" in rows[i0 - 1]["Kernel_Name"]:
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
for r in rows[i0:i0 + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("mrg::(anonymous namespace)::", "").replace("void ", "")[:48]
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name))
    prev = e
