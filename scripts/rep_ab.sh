#!/bin/bash
# mid-stream drain of the walk records: tests, the buffer sizes, then the headline A/B (libA = before)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_repeats.py tests/test_gpu_dict.py tests/test_gpu_dict_edges.py tests/test_gpu_split.py tests/test_gpu_random_worlds.py -m gpu -x -q 2>&1 | tail -5
i=0
run() {
  timeout 600 python bench.py --workload repeats --steps 5 --no-legs --no-extras --no-cpu-baseline --scan-sample 0 "$@" > gpurun_out/rexp_$i.json 2> gpurun_out/rexp_$i.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/rexp_$i.json"))
    print("$*", d["ms_per_step"], [(p["kernel"][:9], round(p["ms"], 3), p["aligned"], p.get("steps"), p["candidates"]) for p in d["passes"] if p["ms"] > 0.01 or p["lib"] in ("ncrna_others", "mrna")])
except Exception as e:
    print("$*", "failed", e); print(open("gpurun_out/rexp_$i.err").read()[-400:])
PY
  i=$((i+1))
}
run
run --opt walk_diag=1
run --opt walk_diag=2
bash scripts/lib_ab.sh "--no-legs --no-extras --no-cpu-baseline" 2
bash scripts/lib_ab.sh "--workload repeats --no-legs --no-extras --no-cpu-baseline --scan-sample 0" 2
cp mirge_amd/lib/libB.so mirge_amd/lib/libmirge_amd.so
