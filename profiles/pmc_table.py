#!/usr/bin/env python3
"""Per-dispatch counter table from a rocprofv3 `--pmc ... --output-format csv -d <dir>` run:
one row per dispatch of this repository's kernels (keyed by dispatch id and kernel name, summed
over the instances of each counter).

    python profiles/pmc_table.py gpurun_out/<dir> [counter ...]
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace("mrg::", "")
    return name[:60]


def load(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        raise SystemExit("no *_counter_collection.csv under %s" % d)
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(files[-1])):
        kn = r["Kernel_Name"]
        if "mrg::" not in kn:
            continue
        k = (int(r["Dispatch_Id"]), short(kn))
        agg.setdefault(k, collections.OrderedDict())
        agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return agg


def main():
    agg = load(sys.argv[1])
    want = sys.argv[2:]
    names = want or sorted({c for v in agg.values() for c in v})
    print("| dispatch | kernel | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for (did, kn), v in sorted(agg.items()):
        print("| %d | `%s` | " % (did, kn) + " | ".join("%.4g" % v.get(c, float("nan")) for c in names) + " |")


if __name__ == "__main__":
    main()
