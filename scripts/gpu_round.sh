#!/bin/bash
# One gpurun call of the inner loop: the new tests, then same-box bench lines with option sets.
# usage: scripts/gpu_round.sh "<pytest args>" "<bench args A>" "<bench args B>" ...
mkdir -p gpurun_out
T="$1"; shift
if [ -n "$T" ]; then timeout 1500 python -m pytest $T -x -q 2>&1 | tail -25 > gpurun_out/pytest.log; fi
i=0
for args in "$@"; do
  timeout 900 python bench.py $args > gpurun_out/bench_$i.json 2> gpurun_out/bench_$i.err
  i=$((i+1))
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/bench_*.json')):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], d['ms_per_step'], [(p['kernel'][:14], round(p['ms'], 3)) for p in d['passes']])
    except Exception as e:
        print(f, 'failed', e)
PY
cat gpurun_out/pytest.log 2>/dev/null
