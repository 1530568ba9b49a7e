#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo
# root):  scripts/profile_round.sh r01   -> gpurun_out/<tag>_*; then
# `python profiles/summarize_rocprof.py <tag>` and copy the bench lines into profiles/.
# Counters are collected in their own runs (never together with --kernel-trace/--stats).
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $B --steps 5 --warmup 1 > gpurun_out/${tag}_stats.json 2> gpurun_out/${tag}_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_write.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/${tag}_sq -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_sq.err
python bench.py 2> gpurun_out/bench_final.err > gpurun_out/bench_final.json
python bench.py --workload exact 2> gpurun_out/bench_exact.err > gpurun_out/bench_exact.json
python bench.py --sorted --no-cpu-baseline 2> /dev/null > gpurun_out/bench_sorted.json
