#!/bin/bash
# collapse experiments on the GPU box: MIRGE_COLLAPSE_* knobs, kernel times per variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  echo "== $v"
  env MIRGE_COLLAPSE_DBG=8 $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x_stats -- python3 scripts/collapse_bench.py --reps 3 > gpurun_out/x.log 2> gpurun_out/x.err
  grep collapse_fast gpurun_out/x.err | head -2; tail -1 gpurun_out/x.log
  python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/x_stats/**/*kernel_stats.csv", recursive=True))
for r in list(csv.DictReader(open(f[-1])))[:8]:
    print("%-50s avg %10.1f us" % (r["Name"][26:76], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/x_stats
done
