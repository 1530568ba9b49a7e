"""Multi-process path on CPU (gloo, world_size 2): sharding + the single fused
all-reduce of mirge_amd.dist give the same count vector as one process, and
`filter` applied AFTER the reduce equals `filter` on the whole read set
(it is non-linear, filter.py:7-13, so it must not run per shard)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import model
from tests.util import World


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, payload, out_path):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mirge_amd import dist as mdist
    r, _, w = mdist.init_process_group("gloo")
    assert (r, w) == (rank, world_size)
    pass_id, ref_id, quant, M, n_pass, stats_by_rank = payload
    lo, hi = mdist.shard_bounds(len(pass_id), rank, world_size)
    fused, ln = mdist.fused_buffer(None, n_mirna=M, n_samples=quant.shape[1], n_pass=n_pass)
    local = model.tally(pass_id[lo:hi], ref_id[lo:hi], quant[lo:hi], M, n_pass, 0, 8)
    fused[:ln] = torch.from_numpy(local.astype(np.int64))
    fused[ln:] = torch.from_numpy(stats_by_rank[rank].astype(np.int64))
    mdist.allreduce_counts(fused)
    if rank == 0:
        np.save(out_path, fused.numpy())
    torch.distributed.destroy_process_group()


def test_shard_bounds_cover_everything():
    from mirge_amd.dist import shard_bounds
    for n in (0, 1, 7, 100, 101):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_allreduce_equals_single_process(native_lib, oracle_lib, tmp_path):
    from mirge_amd import synth
    from mirge_amd.engine import split_counts
    world = World(scale=0.02, n_fixed=4000, n_var=300)
    ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask)
    n = len(world.lens)
    quant = synth.synth_quant(n, n_samples=2)
    M, n_pass = world.n_mirna, 9
    # per-rank per-pass (processed, aligned): rerun the port on each shard
    from mirge_amd.dist import shard_bounds
    stats = []
    for r in range(2):
        lo, hi = shard_bounds(n, r, 2)
        nm = None if world.nmask is None else world.nmask[:, lo:hi]
        part = model.fm_cascade(world.views, world.passes, world.words[:, lo:hi], world.lens[lo:hi], nm)
        assert np.array_equal(part["pass_id"], ref["pass_id"][lo:hi])  # a read's outcome is local
        stats.append(part["stats"][:, :2].reshape(-1))
    out = str(tmp_path / "fused.npy")
    mp.spawn(_worker, args=(2, _free_port(), (ref["pass_id"], ref["ref_id"], quant, M, n_pass, stats), out),
             nprocs=2, join=True)
    fused = np.load(out)
    whole = model.tally(ref["pass_id"], ref["ref_id"], quant, M, n_pass, 0, 8)
    ln = len(whole)
    assert np.array_equal(fused[:ln].astype(np.uint64), whole)
    assert np.array_equal(fused[ln:], ref["stats"][:, :2].reshape(-1).astype(np.int64))

    # filter after the reduce == filter on the whole set; per-shard filtering differs
    from mirge_amd import annotate
    names = world.index["mirna"].names

    def to_dic(counts):
        q, c, _, _ = split_counts(counts, M, 2, n_pass)
        return {nm_: {"quant": [int(x) for x in q[i]], "iscan": [int(x) for x in c[i]]}
                for i, nm_ in enumerate(names)}
    a, b = to_dic(fused[:ln]), to_dic(whole)
    la, lb = {"quantStats": [{}, {}]}, {"quantStats": [{}, {}]}
    annotate.filter(a, ["s0", "s1"], la, "0.1")
    annotate.filter(b, ["s0", "s1"], lb, "0.1")
    assert a == b and la == lb


def _cli_worker(rank, world_size, port, argv, out_path):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MIRGE_AMD_DIST_BACKEND="gloo")
    from mirge_amd import cli
    from tests.fake_engine import OracleEngine
    out = cli.annotate_main(cli.build_parser().parse_args(argv), engine_factory=OracleEngine)
    assert (out is None) == (rank != 0)
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write(out["outdir"])
    torch.distributed.destroy_process_group()


def _exchange_worker(rank, world_size, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mirge_amd import dist as mdist
    mdist.init_process_group("gloo")
    rng = np.random.default_rng(100 + rank)
    n = 500 + 37 * rank
    # a pool of 300 sequences shared by all ranks: every rank holds copies of most of them
    pool = np.random.default_rng(1).integers(0, 2 ** 40, (2, 300), dtype=np.int64)
    pick = rng.integers(0, 300, n)
    words = torch.from_numpy(np.ascontiguousarray(pool[:, pick]))
    lens = torch.from_numpy((20 + pick % 5).astype(np.uint8))
    tag = torch.from_numpy((rank * 10000 + np.arange(n)).astype(np.int32))
    dest = mdist.sequence_destination(words, lens, world_size)
    got_w, got_l, got_none, got_tag = mdist.exchange_by_destination(dest, [words.t().contiguous(), lens, None, tag])
    assert got_none is None and got_w.shape[0] == got_l.shape[0] == got_tag.shape[0]
    # everything this rank received is its own by the same rule, wherever it came from
    assert bool((mdist.sequence_destination(got_w.t().contiguous(), got_l, world_size) == rank).all())
    whole = mdist.gather_to_rank0(got_tag)
    assert (whole is None) == (rank != 0)
    objs = mdist.gather_objects_to_rank0({"rank": rank, "n": n})
    assert mdist.allreduce_max([rank, 7 - rank]) == [world_size - 1, 7]
    if rank == 0:
        assert sorted(o["rank"] for o in objs) == list(range(world_size))
        np.save(os.path.join(out_dir, "tags.npy"), whole.numpy())
        np.save(os.path.join(out_dir, "n.npy"), np.array([o["n"] for o in objs]))
    torch.distributed.destroy_process_group()


def test_partition_by_sequence_moves_every_read_exactly_once(tmp_path):
    """sequence_destination + exchange_by_destination + gather_to_rank0 on three gloo ranks: no read
    lost or duplicated, copies of one sequence end on one rank (asserted inside the workers)."""
    mp.spawn(_exchange_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    tags, ns = np.load(str(tmp_path / "tags.npy")), np.load(str(tmp_path / "n.npy"))
    want = np.concatenate([r * 10000 + np.arange(n) for r, n in enumerate(ns)])
    assert np.array_equal(np.sort(tags), np.sort(want))


@pytest.mark.parametrize("world_size", [2, 3])
def test_cli_ranks_write_the_same_tables_as_one(native_lib, oracle_lib, tmp_path, world_size):
    """`annotate --gpus N` as its children run it (here: gloo, CPU tensors, the oracle's CPU port in
    place of the GPU engine): every rank ingests its own files (with three ranks and two files one
    rank ingests nothing), the raw reads are partitioned by sequence with one all-to-all -- the same
    sequences occur in both files -- every rank collapses and annotates its own sequences, the count
    vector is all-reduced, the per-read arrays go to rank 0 only, and rank 0 writes: every table
    identical to the single-process run (filter after the reduce, trimmedUniq and the per-pass
    counters not double-counted)."""
    from mirge_amd import cli, synth
    from tests.fake_engine import OracleEngine
    from tests.golden.make_golden import SHAPES
    from tests.test_cli import write_fastq
    rng = np.random.default_rng(4)
    libs = synth.SynthLibraries(seed=123, scale=1.0, n_paralogs=6, n_snp=8, shapes=SHAPES)
    libs.write_layout(str(tmp_path / "libs"), species="syn", db="miRBase")
    fastqs = []
    for si in range(2):
        reads = [synth.codes_to_str(c) for c in synth.synth_reads(libs, 1200, seed=70 + si, zipf_s=1.3)]
        # an N, a read of eight packed words, and reads beyond the 255 nt of a packed batch (the long-read lane): one that
        # aligns nowhere, one cut from an mRNA entry that occurs in BOTH files (its copies meet on rank 0)
        long_mrna = max(libs.libs["mrna"][1], key=len)
        assert len(long_mrna) > 300
        reads += ["ACGTNACGTTAGCATCGATCGA", libs.libs["mrna"][1][si][3:160], "A" * 300, long_mrna[2:290]]
        p = str(tmp_path / ("s%d.fastq" % si))
        write_fastq(p, reads, rng)
        fastqs.append(p)
    base = ["annotate", "-s"] + fastqs + ["-lib", str(tmp_path / "libs"), "-sp", "syn", "-di", "-tcf"]
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    one = cli.annotate_main(cli.build_parser().parse_args(base + ["-o", str(tmp_path / "one")]),
                            engine_factory=OracleEngine)
    marker = str(tmp_path / "outdir.txt")
    mp.spawn(_cli_worker, args=(world_size, _free_port(), base + ["-o", str(tmp_path / "two")], marker), nprocs=world_size,
             join=True)
    two_dir = open(marker).read()
    files = sorted(os.listdir(one["outdir"]))
    assert files == sorted(os.listdir(two_dir)) and "mapped.csv" in files and "isomirs.csv" in files
    for fn in files:
        a = open(os.path.join(one["outdir"], fn)).read()
        b = open(os.path.join(two_dir, fn)).read()
        if fn == "annotation.report.csv":
            assert a == b, fn
        else:
            assert sorted(a.split("\n")) == sorted(b.split("\n")), fn
    # the reads beyond 255 nt: annotated by the long-read lane on rank 0, one row each, counts per sample
    assert any(line.startswith("A" * 300 + ",0,") for line in open(os.path.join(two_dir, "unmapped.csv")))
    long_mrna = max(libs.libs["mrna"][1], key=len)
    rows = [line.strip().split(",") for line in open(os.path.join(two_dir, "mapped.csv")) if line.startswith(long_mrna[2:290] + ",")]
    assert len(rows) == 1 and rows[0][1] == "1" and rows[0][9] != "" and rows[0][-2:] == ["1", "1"]
    assert any(line.startswith(libs.libs["mrna"][1][1][3:160] + ",1,") for line in open(os.path.join(two_dir, "mapped.csv")))
    rep = open(os.path.join(two_dir, "annotation.report.csv")).read().split("\n")[1].split(",")
    assert int(rep[1]) == 1204 and int(rep[2]) == one["logDic"]["quantStats"][0]["trimmedReads"]


def test_cli_one_file_three_ranks(native_lib, oracle_lib, tmp_path):
    """`annotate --gpus 3` on ONE sample (the common case: MAIN:289-314 takes one file per sample): the file is cut into
    three byte ranges at record starts, every rank ingests its own (mrg_fastq_load_part), the reads are partitioned by
    sequence, and rank 0 writes the tables of the single-process run -- totalReads / trimmedReads of the sample are
    the sums over the parts, a read beyond 255 nt that occurs in two parts is one row."""
    from mirge_amd import cli, dist as mdist, synth
    from tests.fake_engine import OracleEngine
    from tests.golden.make_golden import SHAPES
    from tests.test_cli import write_fastq
    assert mdist.file_shares(1, 3) == [[(0, 0, 3)], [(0, 1, 3)], [(0, 2, 3)]]
    assert mdist.file_shares(2, 5) == [[(0, 0, 2)], [(0, 1, 2)], [(1, 0, 2)], [(1, 1, 2)], []]
    assert mdist.file_shares(3, 2) == [[(0, 0, 1), (2, 0, 1)], [(1, 0, 1)]]
    rng = np.random.default_rng(6)
    libs = synth.SynthLibraries(seed=123, scale=1.0, n_paralogs=6, n_snp=8, shapes=SHAPES)
    libs.write_layout(str(tmp_path / "libs"), species="syn", db="miRBase")
    long_mrna = max(libs.libs["mrna"][1], key=len)
    reads = [synth.codes_to_str(c) for c in synth.synth_reads(libs, 3000, seed=71, zipf_s=1.3)]
    reads[100:100] = [long_mrna[2:290]]
    reads += ["ACGTNACGTTAGCATCGATCGA", long_mrna[2:290], "A" * 300]
    p = str(tmp_path / "only.fastq")
    write_fastq(p, reads, rng)
    base = ["annotate", "-s", p, "-lib", str(tmp_path / "libs"), "-sp", "syn", "-di"]
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    one = cli.annotate_main(cli.build_parser().parse_args(base + ["-o", str(tmp_path / "one")]), engine_factory=OracleEngine)
    marker = str(tmp_path / "outdir.txt")
    mp.spawn(_cli_worker, args=(3, _free_port(), base + ["-o", str(tmp_path / "three")], marker), nprocs=3, join=True)
    three_dir = open(marker).read()
    files = sorted(os.listdir(one["outdir"]))
    assert files == sorted(os.listdir(three_dir)) and "mapped.csv" in files
    for fn in files:
        a = open(os.path.join(one["outdir"], fn)).read()
        b = open(os.path.join(three_dir, fn)).read()
        if fn == "annotation.report.csv":
            assert a == b, fn
        else:
            assert sorted(a.split("\n")) == sorted(b.split("\n")), fn
    rep = open(os.path.join(three_dir, "annotation.report.csv")).read().split("\n")[1].split(",")
    assert int(rep[1]) == len(reads)
    rows = [l for l in open(os.path.join(three_dir, "mapped.csv")) if l.startswith(long_mrna[2:290] + ",")]
    assert len(rows) == 1 and rows[0].strip().endswith(",2")
