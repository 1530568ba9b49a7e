#!/bin/bash
# The driver's N > 1 command lines at full size on a ONE-GPU box: all ranks on device 0, gloo instead of RCCL
# (MRG_BENCH_SHARE_GPU=1; the timings mean nothing, the path -- shards, gates, the one JSON line -- is what runs).
mkdir -p gpurun_out
export MRG_BENCH_SHARE_GPU=1
for n in 2 8; do
  t0=$(date +%s)
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
    bench.py --gpus $n --steps 5 --warmup 1 > gpurun_out/share_n$n.json 2> gpurun_out/share_n$n.err
  echo "N=$n rc=$? wall=$(( $(date +%s) - t0 )) s"
  python - <<PY
import json
try:
    d = [json.loads(l) for l in open("gpurun_out/share_n$n.json") if l.strip().startswith("{")][-1]
    print(d["n_gpus"], d["ms_per_step"], d["value"], d["scaling"], d["config"]["reads_per_gpu"], d["parity"].get("all_reduce"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/share_n$n.err").read()[-1500:])
PY
done
free -g | head -2
