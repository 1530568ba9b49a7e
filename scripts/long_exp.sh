#!/bin/bash
# the reads of 33..40 nt of the varlen batch alone: the LONG instantiations of the seed kernels against the FM kernels
mkdir -p gpurun_out
i=0
run() {
  timeout 600 python bench.py --workload varlen --steps 5 --no-legs --no-extras --no-cpu-baseline --scan-sample 0 "$@" > gpurun_out/lexp_$i.json 2> gpurun_out/lexp_$i.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/lexp_$i.json"))
    print("$*", d["config"].get("reads_total"), d["ms_per_step"], d.get("split_batch", {}).get("rest_ms"), [(p["kernel"][:26], round(p["ms"], 3), p.get("ms_rest"), p["processed"], p["aligned"]) for p in d["passes"] if p["ms"] > 0.005 or (p.get("ms_rest") or 0) > 0.005])
except Exception as e:
    print("$*", "failed", e); print(open("gpurun_out/lexp_$i.err").read()[-500:])
PY
  i=$((i+1))
}
run --keep-lens 33,40 --opt long_lane=1

run --opt long_lane=1
