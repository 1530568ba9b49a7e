#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo
# root):  scripts/profile_round.sh r02 <git head>   -> gpurun_out/<tag>_*; then
# `python profiles/summarize_rocprof.py <tag>` here and copy the bench lines into profiles/.
# Counters are collected in their own runs (never together with --kernel-trace/--stats).
tag=${1:-r02}
head=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
sha=$(python3 -c "import bench; print(bench.kernels_sha16())")
echo "{\"git_head\": \"$head\", \"kernels_sha16\": \"$sha\"}" > gpurun_out/${tag}_meta.json
B="python3 bench.py --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $B --steps 5 --warmup 1 > gpurun_out/${tag}_stats.json 2> gpurun_out/${tag}_stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_write.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/${tag}_sq -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_sq.err
timeout 600 rocprofv3 --pmc TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES --output-format csv -d gpurun_out/${tag}_tcp -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_tcp.err
timeout 600 rocprofv3 --pmc TCC_REQ TCC_HIT TCC_MISS --output-format csv -d gpurun_out/${tag}_tcc -- $B --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_tcc.err
# the a2i workload's kernels (edit tally) by time
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_a2i_stats -- $B --workload a2i --steps 5 --warmup 1 > gpurun_out/${tag}_a2i_stats.json 2> gpurun_out/${tag}_a2i_stats.err
# ... and of the exact (BASELINE configs[1], the roofline configuration) and varlen workloads
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_exact_stats -- $B --workload exact --steps 20 --warmup 2 > gpurun_out/${tag}_exact_stats.json 2> gpurun_out/${tag}_exact_stats.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_varlen_stats -- $B --workload varlen --steps 5 --warmup 1 > gpurun_out/${tag}_varlen_stats.json 2> gpurun_out/${tag}_varlen_stats.err
# HBM traffic of the exact (BASELINE configs[1]: where the 40 % target is quoted) and a2i legs, for roofline.traffic of those lines
for wl in exact a2i; do
  st="--steps 1 --warmup 0"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_${wl}_fetch -- $B --workload $wl $st > /dev/null 2> gpurun_out/${tag}_${wl}_fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_${wl}_write -- $B --workload $wl $st > /dev/null 2> gpurun_out/${tag}_${wl}_write.err
done
# the bench lines proper (no profiler attached)
python bench.py 2> gpurun_out/${tag}_bench_cascade.err > gpurun_out/${tag}_bench_cascade.json
python bench.py --workload exact 2> gpurun_out/${tag}_bench_exact.err > gpurun_out/${tag}_bench_exact.json
python bench.py --workload a2i 2> gpurun_out/${tag}_bench_a2i.err > gpurun_out/${tag}_bench_a2i.json
python bench.py --workload varlen 2> gpurun_out/${tag}_bench_varlen.err > gpurun_out/${tag}_bench_varlen.json
python bench.py --sorted --no-cpu-baseline --no-extras 2> /dev/null > gpurun_out/${tag}_bench_sorted.json
# the shards a 2 / 4 / 8-GPU strong-scaling job gives every rank, on this one GPU (the all-reduce is not in them)
for pair in 12500000:12m5 25000000:25m 50000000:50m 100000000:full; do
  python bench.py --no-cpu-baseline --no-extras --reads ${pair%%:*} 2> /dev/null > gpurun_out/${tag}_shard_${pair##*:}.json
done
# ---- round 5 ----
# the unfriendly library set (repeats), the collapse kernels alone (stats + HBM bytes), library residency, the command line at scale
python bench.py --workload repeats --no-extras --no-legs 2> gpurun_out/${tag}_bench_repeats.err > gpurun_out/${tag}_bench_repeats.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_collapse_stats -- python3 scripts/collapse_bench.py > gpurun_out/${tag}_collapse.json 2> gpurun_out/${tag}_collapse.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_collapse_fetch -- python3 scripts/collapse_bench.py --reps 1 > /dev/null 2> gpurun_out/${tag}_collapse_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_collapse_write -- python3 scripts/collapse_bench.py --reps 1 > /dev/null 2> gpurun_out/${tag}_collapse_write.err
timeout 600 python scripts/collapse_bench.py > gpurun_out/${tag}_collapse_plain.json 2> /dev/null
timeout 600 python scripts/lib_load_timing.py 1.0 > /dev/null 2> gpurun_out/${tag}_lib_load.txt
timeout 900 python scripts/cli_scale_check.py 32000000 1.0 1 > gpurun_out/${tag}_cli_scale_32m.txt 2>&1
timeout 600 python scripts/cli_scale_check.py 32000000 0.2 1 > gpurun_out/${tag}_cli_scale_32m_s02.txt 2>&1
# ---- round 6 ----
# the repeats workload's kernels by time and by HBM bytes (the large launch answers wide seeds from position lists), the LONG
# lane against the default on the varlen batch, one genome part under the profiler (count_variants_kernel + count_kernel)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_repeats_stats -- $B --workload repeats --steps 5 --warmup 1 > gpurun_out/${tag}_repeats_stats.json 2> gpurun_out/${tag}_repeats_stats.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_repeats_fetch -- $B --workload repeats --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_repeats_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_repeats_write -- $B --workload repeats --steps 1 --warmup 0 > /dev/null 2> gpurun_out/${tag}_repeats_write.err
python bench.py --workload varlen --no-extras --no-legs --opt long_lane=1 2> /dev/null > gpurun_out/${tag}_bench_varlen_long_lane.json
python bench.py --workload varlen --no-extras --no-legs --no-cpu-baseline --scan-sample 0 --keep-lens 33,40 --opt long_lane=1 2> /dev/null > gpurun_out/${tag}_bench_long_only_lane.json
python bench.py --workload varlen --no-extras --no-legs --no-cpu-baseline --scan-sample 0 --keep-lens 33,40 --opt long_lane=0 2> /dev/null > gpurun_out/${tag}_bench_long_only_fm.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_genome_stats -- python3 scripts/genome_scale_check.py --reads 2000000 > gpurun_out/${tag}_genome_part.json 2> gpurun_out/${tag}_genome_stats.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_genome_fetch -- python3 scripts/genome_scale_check.py --reads 2000000 > /dev/null 2> gpurun_out/${tag}_genome_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_genome_write -- python3 scripts/genome_scale_check.py --reads 2000000 > /dev/null 2> gpurun_out/${tag}_genome_write.err
# a 12.5 M-read shard as two half-shards on two streams (round-5 verdict, item 9): cascade only, results not checked
timeout 900 python scripts/overlap_probe.py --reads 12500000 --chunks 1,2 --grid-pct 100,60,50 --opt fused_step=1 --reps 20 2> /dev/null > gpurun_out/${tag}_shard_halves.txt
timeout 2400 python scripts/genome_whole_check.py 2> /dev/null | grep "^{" > gpurun_out/${tag}_genome_whole.json
