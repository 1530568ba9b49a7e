#!/bin/bash
# same-box comparison of the step's fixed costs on a 12.5 M-read shard (what one of 8 ranks holds) and on the whole set
cd "$GRAFT_REPO_ROOT"
for v in "--reads 12500000 --fused-step 0" "--reads 12500000" "--reads 12500000 --graph 1" "--reads 12500000 --fused-step 0" "--reads 12500000" "--reads 12500000 --graph 1" "" "--graph 1"; do
  timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --scan-sample 0 $v > gpurun_out/sx.json 2> gpurun_out/sx.err || tail -3 gpurun_out/sx.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sx.json"))
print("%-40s %.4f ms  %s" % (sys.argv[1], d["ms_per_step"], d["config"]["step"][-60:]))
PY
done
