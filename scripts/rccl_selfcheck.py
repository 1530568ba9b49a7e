#!/usr/bin/env python3
"""RCCL sanity check on whatever GPUs the box has (one here): process group over `nccl`,
int64 all-reduce of a fused-count-sized vector, barrier -- the collective bench.py issues per step.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
        --master-port 29533 scripts/rccl_selfcheck.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world)
v = torch.arange(5642, dtype=torch.int64, device="cuda") * (rank + 1)
dist.all_reduce(v, op=dist.ReduceOp.SUM)
want = torch.arange(5642, dtype=torch.int64, device="cuda") * (world * (world + 1) // 2)
assert torch.equal(v, want)
dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
if rank == 0:
    print("rccl ok: world %d, int64[5642] all-reduce %.1f us" % (world, (time.perf_counter() - t0) * 1e4))
dist.destroy_process_group()
