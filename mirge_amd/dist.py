"""Multi-GPU: shard the globally collapsed read set, replicate the libraries,
all-reduce ONE fused count vector.

The cascade outcome of a read depends only on that read, so ranks never
exchange reads.  The only collective is a sum over
[mir_quant | mir_iscan | category totals | trimmedUniq | per-pass processed,aligned]
(int64, a few tens of KB): latency-bound on xGMI, so it is one all-reduce, not
one per table.  `filter` (filter.py:7-13) is non-linear and must run on the
reduced vector, i.e. after this call.
"""
import os


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), \
        int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend=None):
    """One process per GPU; `nccl` is RCCL on ROCm, `gloo` for the CPU tests."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # the caller has already made LOCAL_RANK its current device (barrier() uses it)
        dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n, rank, world):
    """Contiguous, near-equal slices of the collapsed unique-read arrays.  The
    collapse must be global BEFORE sharding: a sequence present on two ranks would
    be double-counted in trimmedUniq (summarize.py:37) and readsProcessed/Aligned."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def fused_buffer(engine_or_len, n_mirna=None, n_samples=None, n_pass=None, device="cpu"):
    """int64 zeros: tally counts followed by 2*n_pass per-pass counters."""
    import torch
    if n_mirna is None:
        ln = int(engine_or_len)
    else:
        ln = 2 * n_mirna * n_samples + (n_pass + 2) * n_samples
    return torch.zeros(ln + 2 * n_pass, dtype=torch.int64, device=device), ln


def allreduce_counts(fused):
    """Sum the fused vector over all ranks in place (no-op for one process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(fused, op=dist.ReduceOp.SUM)
    return fused


def broadcast_from_rank0(obj):
    """A small picklable object (shapes, flags) from rank 0 to everyone."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def broadcast_tensor(t):
    """In place, from rank 0 (no-op for one process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=0)
    return t


def gather_shards(t, n_total):
    """The per-read arrays of the contiguous shards of `shard_bounds`, put back together: every rank
    passes its own shard (1-D tensor), every rank gets the whole array of n_total elements.  Shards
    differ by at most one element, so they are padded to a common length for one all_gather."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return t
    world = dist.get_world_size()
    cap = -(-n_total // world)
    pad = torch.zeros(cap, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.empty(n_total, dtype=t.dtype, device=t.device)
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out[lo:hi] = parts[r][:hi - lo]
    return out
