#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_dict.py tests/test_gpu_dict_edges.py tests/test_gpu_random_worlds.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3
for v in "" "--mix mrna=0" "--mix rrna_ncrna=0" "--mix mrna=0 --mix rrna_ncrna=0"; do
  timeout 600 python bench.py --workload repeats --reads 2000000 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --scan-sample 0 $v > gpurun_out/rx.json 2> gpurun_out/rx.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/rx.json"))
print(sys.argv[1], d["ms_per_step"], [round(p["ms"], 3) for p in d["passes"]], [(p["candidates"], p["lookups"]) for p in d["passes"]][6:8])
PY
done
