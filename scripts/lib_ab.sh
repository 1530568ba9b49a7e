#!/bin/bash
# Same-box A/B of two builds of the library: mirge_amd/lib/libA.so and libB.so are copied over libmirge_amd.so in turn.
# usage (through gpurun): scripts/lib_ab.sh "<bench args>" [rounds]
mkdir -p gpurun_out
args="$1"; rounds=${2:-2}
for i in $(seq 1 $rounds); do
  for v in A B; do
    cp mirge_amd/lib/lib$v.so mirge_amd/lib/libmirge_amd.so
    timeout 900 python bench.py $args > gpurun_out/ab_${v}_$i.json 2> gpurun_out/ab_${v}_$i.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab_${v}_$i.json"))
    print("$v", $i, d["ms_per_step"], d["roofline"]["kernel"][5:30], d["roofline"]["frac"], [(p["kernel"][:10], round(p["ms"], 4)) for p in d["passes"] if p["ms"] > 0.01])
except Exception as e:
    print("$v", $i, "failed", e)
PY
  done
done
