#!/bin/bash
# repeats experiments (one gpurun call): where the large launch's time goes
mkdir -p gpurun_out
i=0
run() {
  timeout 600 python bench.py --workload repeats --steps 5 --no-legs --no-extras --no-cpu-baseline --scan-sample 0 "$@" > gpurun_out/rexp_$i.json 2> gpurun_out/rexp_$i.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/rexp_$i.json"))
    print("$*", d["ms_per_step"], [(p["kernel"][:9], round(p["ms"], 3), p["processed"], p["aligned"], p.get("steps"), p["candidates"]) for p in d["passes"] if p["ms"] > 0.01 or p["lib"] in ("ncrna_others", "mrna")])
except Exception as e:
    print("$*", "failed", e); print(open("gpurun_out/rexp_$i.err").read()[-400:])
PY
  i=$((i+1))
}
run --opt walk_diag=1
run --opt walk_diag=1 --mix mrna=0 --mix random=0.12
run --opt walk_diag=1 --mix mrna=0 --mix rrna_ncrna=0 --mix random=0.17
