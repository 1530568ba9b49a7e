#!/bin/bash
# exact_dict_kernel<FIRST>: contiguous stretches per workgroup against every gridDim-th chunk (libA = before)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dict.py tests/test_gpu_dict_edges.py tests/test_gpu_golden.py tests/test_gpu_split.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do
  for v in A B; do
    cp mirge_amd/lib/lib$v.so mirge_amd/lib/libmirge_amd.so
    timeout 900 python bench.py --workload exact --no-cpu-baseline --scan-sample 0 --steps 50 --warmup 5 > gpurun_out/ex_${v}_$i.json 2> gpurun_out/ex_${v}_$i.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ex_${v}_$i.json")); r = d["roofline"]; c = r.get("cold") or {}
    print("$v", $i, d["ms_per_step"], r["frac"], "cold", (c.get("cold") or {}).get("kernel_ms"), (c.get("cold") or {}).get("frac"), "warm", (c.get("warm") or {}).get("kernel_ms"), (c.get("warm") or {}).get("frac"))
except Exception as e:
    print("$v", $i, "failed", e); print(open("gpurun_out/ex_${v}_$i.err").read()[-600:])
PY
  done
done
bash scripts/lib_ab.sh "--no-legs --no-extras --no-cpu-baseline" 2
cp mirge_amd/lib/libB.so mirge_amd/lib/libmirge_amd.so
