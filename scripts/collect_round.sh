#!/bin/bash
# After `gpurun -- scripts/profile_round.sh <tag> <head>`: condense gpurun_out/<tag>_* into the tracked files under profiles/.
# usage (here, from the repo root): scripts/collect_round.sh r05
tag=${1:-r06}
python profiles/summarize_rocprof.py $tag > /dev/null 2>&1
python profiles/summarize_rocprof.py ${tag}_exact --workload exact > /dev/null 2>&1
python profiles/summarize_rocprof.py ${tag}_a2i --workload a2i > /dev/null 2>&1
python profiles/summarize_collapse.py $tag > /dev/null
for n in bench_cascade bench_exact bench_a2i bench_varlen bench_sorted bench_repeats shard_12m5 shard_25m shard_50m shard_full; do cp gpurun_out/${tag}_$n.json profiles/${tag}_$n.json; done
cp gpurun_out/${tag}_lib_load.txt profiles/${tag}_lib_load.txt
cp gpurun_out/${tag}_shard_halves.txt profiles/${tag}_shard_halves.txt; cp gpurun_out/${tag}_genome_whole.json profiles/${tag}_genome_whole.json
cp gpurun_out/${tag}_collapse_plain.json profiles/${tag}_collapse_bench.json
cp "$(ls -t gpurun_out/${tag}_varlen_stats/*/*kernel_stats.csv | head -1)" profiles/${tag}_varlen_kernel_stats.csv
cp "$(ls -t gpurun_out/${tag}_collapse_stats/*/*kernel_stats.csv | head -1)" profiles/${tag}_collapse_kernel_stats.csv
grep -v amdgpu gpurun_out/${tag}_cli_scale_32m.txt | grep "resident\|cycles completed\|Summary\|Completed\|^{" > profiles/${tag}_cli_scale_32m.json
grep -v amdgpu gpurun_out/${tag}_cli_scale_32m_s02.txt | grep "resident\|cycles completed\|Summary\|Completed\|^{" > profiles/${tag}_cli_scale_32m_s02.json
python profiles/summarize_rocprof.py ${tag}_repeats --workload repeats > /dev/null 2>&1
python profiles/summarize_genome.py $tag > /dev/null
for n in bench_varlen_long_lane bench_long_only_lane bench_long_only_fm; do cp gpurun_out/${tag}_$n.json profiles/${tag}_$n.json 2>/dev/null; done
grep "^{" gpurun_out/${tag}_genome_part.json > profiles/${tag}_genome_part.json
# (the last workload summarised owns traffic.json's _meta: the cascade's again)
python profiles/summarize_rocprof.py $tag > /dev/null 2>&1
# round-5 verdict, item 10: the evidence must be of THIS tree -- a kernel change that forgot to re-collect fails here
python - <<PY || { echo "collect_round: profiles/traffic.json was collected on other device sources than the tree's: re-run scripts/profile_round.sh" >&2; exit 1; }
import json, sys, bench
t = json.load(open("profiles/traffic.json"))
have, want = t["_meta"]["kernels_sha16"], bench.kernels_sha16()
print("traffic.json kernels", have, "tree", want)
sys.exit(0 if have == want else 1)
PY
