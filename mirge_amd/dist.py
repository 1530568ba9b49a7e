"""Multi-GPU: partition the reads by SEQUENCE, replicate the libraries, all-reduce ONE fused count
vector.

The cascade outcome of a read depends only on that read.  Two ways to give every rank its own
reads: contiguous shards of a globally collapsed set (`shard_bounds`: bench.py, whose synthetic
read set exists on every rank), or -- the command line, whose reads start in FASTQ files -- every
rank ingests its own files and the raw reads are hash-partitioned by sequence with one all-to-all
(`sequence_destination`, `exchange_by_destination`): all copies of a sequence land on one rank, the
per-rank collapses are disjoint, so trimmedUniq (summarize.py:37) and readsProcessed / readsAligned
(runAnnotationPipeline.py:648-650) add up without a global collapse and no rank ever holds the whole
read set.  After that the only collective of the data path is a sum over
[mir_quant | mir_iscan | category totals | trimmedUniq | per-pass processed,aligned]
(int64, a few tens of KB): latency-bound on xGMI, so it is one all-reduce, not
one per table.  `filter` (filter.py:7-13) is non-linear and must run on the
reduced vector, i.e. after this call.
"""
import os


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), \
        int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend=None, timeout_s=None):
    """One process per GPU; `nccl` is RCCL on ROCm, `gloo` for the CPU tests.  With nccl the rank's
    GPU (LOCAL_RANK) is made torch's current device first and bound to the process group, so that
    no collective can ever pick device 0 on every rank."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or "nccl"
        kw = {}
        if timeout_s:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        if backend == "nccl":
            import torch
            n_dev = torch.cuda.device_count()
            dev = local_rank % max(n_dev, 1) if os.environ.get("MRG_BENCH_SHARE_GPU") == "1" else local_rank
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def barrier():
    """Barrier on the rank's own device (no-op for one process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    if dist.get_backend() == "nccl":
        import torch
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


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


def allreduce_counts(fused, async_op=False):
    """Sum the fused vector over all ranks in place (no-op for one process).  async_op: returns the collective's
    work handle (None for one process) instead of making the current stream wait for it -- `handle.wait()` does
    that later, so the next batch's kernels can run beside the all-reduce (the caller must not touch `fused`
    before)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        work = dist.all_reduce(fused, op=dist.ReduceOp.SUM, async_op=bool(async_op))
        return work if async_op else fused
    return None if async_op else fused


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


# ---------------------------------------------------------------------------
# partition by sequence (the command line's multi-GPU data path)
# ---------------------------------------------------------------------------
def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def sequence_destination(words, lens, world):
    """Rank that owns each read: a hash of (length, packed bases) modulo the world size, computed
    where the reads live (words int64 [W, n], lens uint8 [n]).  Identical sequences -- whichever
    rank ingested them -- get the same destination; an N mask need not enter the hash."""
    import torch
    h = lens.to(torch.int64) * -7046029254386353131           # 0x9E3779B97F4A7C15 as int64
    for w in range(words.shape[0]):
        x = words[w]
        x = (x ^ (x >> 31)) * (-4658895280553007687 + 2 * w)   # odd multipliers, wrap-around arithmetic
        h = (h ^ x) * -7046029254386353131
        h = h ^ (h >> 29)
    return ((h >> 17) & 0x7FFFFFFF) % int(world)


def exchange_by_destination(dest, tensors):
    """One all-to-all per tensor: row i of every tensor (first dimension = reads) goes to rank
    dest[i].  Returns the rows this rank received, grouped by sending rank (rows of one sender keep
    their order).  No-op for one process."""
    import torch
    import torch.distributed as dist
    if not _active():
        return list(tensors)
    world = dist.get_world_size()
    order = torch.argsort(dest, stable=True)
    send_counts = torch.bincount(dest, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    sc, rc = [int(x) for x in send_counts.cpu()], [int(x) for x in recv_counts.cpu()]
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        # (as bytes: neither RCCL nor gloo moves every integer width -- int16 sample ids, for one)
        k = t.element_size()
        for d in t.shape[1:]:
            k *= int(d)
        src = t[order].contiguous().reshape(-1).view(torch.uint8)
        dst = torch.empty(sum(rc) * k, dtype=torch.uint8, device=t.device)
        dist.all_to_all_single(dst, src, [c * k for c in rc], [c * k for c in sc])
        out.append(dst.view(t.dtype).reshape((sum(rc),) + tuple(t.shape[1:])))
    return out


def gather_to_rank0(t):
    """Per-read arrays for the table writers: rank 0 gets the concatenation (in rank order) of every
    rank's tensor along the first dimension, the other ranks get None.  Point-to-point sends into
    place: nothing is replicated on ranks that do not write (no all_gather)."""
    import torch
    import torch.distributed as dist
    if not _active():
        return t
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([int(t.shape[0])], dtype=torch.int64, device=t.device)
    sizes = [torch.empty_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(x.item()) for x in sizes]
    if rank != 0:
        if sizes[rank]:
            dist.send(t.contiguous().reshape(-1).view(torch.uint8), dst=0)   # (bytes: see exchange_by_destination)
        return None
    out = torch.empty((sum(sizes),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:sizes[0]] = t
    at = sizes[0]
    row_bytes = t.element_size()
    for d in t.shape[1:]:
        row_bytes *= int(d)
    for r in range(1, world):
        if sizes[r]:
            buf = torch.empty(sizes[r] * row_bytes, dtype=torch.uint8, device=t.device)
            dist.recv(buf, src=r)
            out[at:at + sizes[r]] = buf.view(t.dtype).reshape((sizes[r],) + tuple(t.shape[1:]))
        at += sizes[r]
    return out


def gather_objects_to_rank0(obj):
    """Small picklable per-rank objects (ingest statistics, over-long reads): list on rank 0, None elsewhere."""
    import torch.distributed as dist
    if not _active():
        return [obj]
    box = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, box, dst=0)
    return box


def allreduce_max(values):
    """Element-wise maximum of a short list of ints over all ranks."""
    import torch
    import torch.distributed as dist
    if not _active():
        return list(values)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(x) for x in t.cpu()]


def file_shares(n_files, world):
    """Who ingests what: per rank a list of (file, part, n_parts).  At least as many files as ranks: file i goes to
    rank i mod world, whole.  Fewer files than ranks (`--gpus 8` on one sample): every file is read by
    world // n_files ranks, each its own part of it (mrg_fastq_load_part); the ranks left over ingest nothing (they
    still receive their share of the sequences in the exchange)."""
    shares = [[] for _ in range(world)]
    if n_files >= world:
        for i in range(n_files):
            shares[i % world].append((i, 0, 1))
        return shares
    per = max(1, world // max(n_files, 1))
    for i in range(n_files):
        for part in range(per):
            shares[i * per + part].append((i, part, per))
    return shares
