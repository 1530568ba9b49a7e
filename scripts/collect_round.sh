#!/bin/bash
# After `gpurun -- scripts/profile_round.sh <tag> <head>`: condense gpurun_out/<tag>_* into the tracked files under profiles/.
# usage (here, from the repo root): scripts/collect_round.sh r05
tag=${1:-r05}
python profiles/summarize_rocprof.py $tag > /dev/null 2>&1
python profiles/summarize_rocprof.py ${tag}_exact --workload exact > /dev/null 2>&1
python profiles/summarize_rocprof.py ${tag}_a2i --workload a2i > /dev/null 2>&1
python profiles/summarize_collapse.py $tag > /dev/null
for n in bench_cascade bench_exact bench_a2i bench_varlen bench_sorted bench_repeats shard_12m5 shard_25m shard_50m shard_full; do cp gpurun_out/${tag}_$n.json profiles/${tag}_$n.json; done
cp gpurun_out/${tag}_lib_load.txt profiles/${tag}_lib_load.txt
cp gpurun_out/${tag}_collapse_plain.json profiles/${tag}_collapse_bench.json
cp "$(ls -t gpurun_out/${tag}_varlen_stats/*/*kernel_stats.csv | head -1)" profiles/${tag}_varlen_kernel_stats.csv
cp "$(ls -t gpurun_out/${tag}_collapse_stats/*/*kernel_stats.csv | head -1)" profiles/${tag}_collapse_kernel_stats.csv
grep -v amdgpu gpurun_out/${tag}_cli_scale_32m.txt | grep "resident\|cycles completed\|Summary\|Completed\|^{" > profiles/${tag}_cli_scale_32m.json
grep -v amdgpu gpurun_out/${tag}_cli_scale_32m_s02.txt | grep "resident\|cycles completed\|Summary\|Completed\|^{" > profiles/${tag}_cli_scale_32m_s02.json
python - <<PY
import json, bench
t = json.load(open("profiles/traffic.json"))
print("traffic.json kernels", t["_meta"]["kernels_sha16"], "tree", bench.kernels_sha16())
PY
