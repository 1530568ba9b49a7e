#!/bin/bash
# rocprofv3 kernel stats of one command on the GPU box: scripts/prof_cmd.sh <tag> <python args...>
# prints the kernels by share of time; the csv lands under gpurun_out/<tag>_stats/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag="$1"; shift
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 "$@" > gpurun_out/${tag}.log 2> gpurun_out/${tag}.err
tail -2 gpurun_out/${tag}.log
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s_stats/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
for r in list(csv.DictReader(open(f[0])))[:24]:
    print("%-70s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
