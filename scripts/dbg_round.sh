#!/bin/bash
mkdir -p gpurun_out
for d in 0 1 2 4; do
  MRG_DBG=$d python bench.py --no-cpu-baseline --no-extras --steps 3 > gpurun_out/dbg_$d.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/dbg_*.json')):
    d = json.load(open(f)); print(f.split('/')[-1], d['ms_per_step'], [round(p['ms'],3) for p in d['passes']], [p['lookups'] for p in d['passes']], [p['candidates'] for p in d['passes']])
PY
