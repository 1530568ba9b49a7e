#!/bin/bash
# Same-box A/B of two builds of libmirge_amd.so (both must share the C-ABI of the checked-out
# Python side): scripts/ab_bench.sh build_ab/libA.so build_ab/libB.so [bench args]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for v in A B; do
    src=$A; [ $v = B ] && src=$B
    cp "$src" mirge_amd/lib/libmirge_amd.so
    python bench.py --no-cpu-baseline "$@" 2>/dev/null > gpurun_out/ab_bench_${v}_${rep}.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_bench_*_*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], d['ms_per_step'], [round(p['ms'], 3) for p in d['passes']])
PY
