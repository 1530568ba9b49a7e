#!/bin/bash
# After collect_round.sh refreshed profiles/traffic.json: the complete -m gpu suite, smoke(), and the bench lines re-run so that
# they carry roofline.traffic / frac_kernel_time of THIS tree's kernels.  usage (through gpurun): scripts/final_lines.sh r06
tag=${1:-r06}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.txt 2>&1
python bench.py 2> gpurun_out/${tag}_bench_cascade.err > gpurun_out/${tag}_bench_cascade.json
for wl in exact a2i repeats varlen; do
  python bench.py --workload $wl 2> gpurun_out/${tag}_bench_$wl.err > gpurun_out/${tag}_bench_$wl.json
done
cat gpurun_out/${tag}_pytest_gpu.txt; tail -3 gpurun_out/${tag}_smoke.txt
python - <<PY
import json
for wl in ("cascade", "exact", "a2i", "repeats", "varlen"):
    try:
        d = json.load(open("gpurun_out/${tag}_bench_%s.json" % wl)); r = d["roofline"]
        print(wl, d["ms_per_step"], r["frac"], r.get("traffic"), r.get("frac_kernel_time"), (r.get("traffic_source") or {}).get("stale"))
    except Exception as e:
        print(wl, "failed", e)
PY
