#!/bin/bash
# Same-box comparison of several builds of the library: mirge_amd/lib/lib<NAME>.so are copied over libmirge_amd.so in turn.
# usage (through gpurun): scripts/lib_multi.sh "A B C" "<bench args>" [rounds]
mkdir -p gpurun_out
names="$1"; args="$2"; rounds=${3:-2}
for i in $(seq 1 $rounds); do
  for v in $names; do
    cp mirge_amd/lib/lib$v.so mirge_amd/lib/libmirge_amd.so
    timeout 900 python bench.py $args > gpurun_out/ab_${v}_$i.json 2> gpurun_out/ab_${v}_$i.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab_${v}_$i.json"))
    print("$v", $i, d["ms_per_step"], [(p["kernel"][:10], round(p["ms"], 4)) for p in d["passes"] if p["ms"] > 0.01])
except Exception as e:
    print("$v", $i, "failed", e)
PY
  done
done
