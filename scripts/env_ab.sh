#!/bin/bash
# Same-box A/B of one environment knob of the library: usage (through gpurun): scripts/env_ab.sh VAR "a b ..." "<bench args>" [rounds]
mkdir -p gpurun_out
var="$1"; vals="$2"; args="$3"; rounds=${4:-2}
for i in $(seq 1 $rounds); do
  for v in $vals; do
    export $var=$v
    timeout 900 python bench.py $args > gpurun_out/env_${v}_$i.json 2> gpurun_out/env_${v}_$i.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/env_${v}_$i.json"))
    ex = d.get("legs", {}).get("exact", {})
    print("$var=$v", $i, d["ms_per_step"], [(p["kernel"][:10], round(p["ms"], 4)) for p in d["passes"] if p["ms"] > 0.01], "exact leg", ex.get("ms_per_step"), (ex.get("roofline") or {}).get("frac"))
except Exception as e:
    print("$var=$v", $i, "failed", e)
PY
  done
done
