"""GPU parity (-m gpu) on RANDOM worlds: library sizes, cascades and context options drawn per seed, so that launch
plans nobody wrote a test for get exercised -- which passes share a launch, which libraries have dictionaries or seed
buckets, tile or wave kernel, split batches, trims and poly-T rules in odd places.  The yardstick is the CPU port
(oracle/fm_cpu.c, itself pinned to the exhaustive scan in tests/test_oracle_cpu.py) on every read, and the exhaustive
scan on a sample for the policies the reference's cascade does not contain."""
import numpy as np
import pytest

from oracle import model
from tests.util import LIB_ORDER, mixed_reads

pytestmark = pytest.mark.gpu

V = 1024   # "-v": the seed region is the whole read


def random_cascade(rng, n_libs):
    passes = []
    for _ in range(int(rng.integers(4, 11))):
        mode = ["n0", "n0", "n1", "n1", "n1", "v0", "v1", "v2", "n2"][int(rng.integers(0, 9))]
        seed_len = V if mode[0] == "v" else int(rng.choice([16, 20, 24, 28, 28, 28, 32]))
        mm_seed = int(mode[1])
        mm_total = mm_seed if mode[0] == "v" else int(rng.choice([mm_seed, 2, 2, 3]))
        lo = int(rng.choice([0, 0, 0, 18, 26]))
        hi = int(rng.choice([255, 255, 255, 25, 30]))
        if lo > hi:
            lo = 0
        passes.append(dict(lib=int(rng.integers(0, n_libs)), seed_len=seed_len, max_mm_seed=mm_seed, max_mm_total=max(mm_total, mm_seed),
                           trim5=int(rng.choice([0, 0, 0, 1, 2])), trim3=int(rng.choice([0, 0, 0, 1, 2, 3])), min_len=lo, max_len=hi,
                           poly_t=int(rng.random() < 0.15)))
    return passes


@pytest.mark.parametrize("seed", range(24))
def test_random_world_random_cascade(native_lib, oracle_lib, seed):
    from mirge_amd import pack, synth
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(4000 + seed)
    libs = synth.SynthLibraries(seed=700 + seed, scale=[0.01, 0.03, 0.06, 0.1][seed % 4], n_paralogs=int(rng.integers(0, 8)),
                                n_snp=int(rng.integers(0, 10)))
    index = {k: FmIndex.build(*libs.libs[k]) for k in LIB_ORDER}
    views = [index[k].view() for k in LIB_ORDER]
    one_word = seed % 2 == 1
    reads = mixed_reads(libs, n_fixed=8000, n_var=5000, seed=seed, with_n=(seed % 3 == 0), max_var_len=32 if one_word else 44)
    words, lens, nmask = pack.pack_reads(reads)
    passes = random_cascade(rng, len(LIB_ORDER))
    opts = dict(seed_impl=int(rng.choice([-1, -1, 0, 1, 2])), pair_impl=int(rng.integers(0, 2)), split_min_len=int(rng.choice([0, 16, 16, 20])),
                fuse=int(rng.random() < 0.8), seed_units=int(rng.random() < 0.8), wstop=int(rng.choice([0, 8])))
    # (round 6: the reads of 33..44 nt of the two-word batches ride the dictionary kernels' LONG instantiations -- off by
    # default, measured no faster than the FM kernels -- in every two-word world but each fourth; a separate generator, so
    # that the worlds of round 5 stay what they were)
    opts["long_lane"] = int(seed % 8 != 2)
    eng = Engine(0)
    big = rng.random() < 0.5
    for k in LIB_ORDER:
        if big:   # (a large library gets an exact-match dictionary too)
            eng.set_option("dict_max_bases", 1 << 30)
        eng.add_library(k, index[k], exact_dict=None if big else False)
    for k, v in opts.items():
        eng.set_option(k, v)
    res = eng.cascade(ReadSet(words, lens, nmask, None, device=eng.device), eng.make_passes(passes))
    ref = model.fm_cascade(views, passes, words, lens, nmask)
    got = res.to_host()
    stats = res.stats   # (counters and event times live in the context: read before the next cascade)
    if seed % 2 == 0:   # the packed output form of the same plan (one word per read instead of four arrays)
        rs2 = ReadSet(words, lens, nmask, None, device=eng.device)
        pk = eng.cascade_packed(rs2, eng.make_passes(passes))
        assert np.array_equal(pk.packed.cpu().numpy(), eng.pack_assignments(res).cpu().numpy()), (seed, opts, passes)
        assert np.array_equal(pk.pass_counts.cpu().numpy(), res.pass_counts.cpu().numpy())
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
        bad = np.flatnonzero(a != ref[name])
        assert bad.size == 0, (name, seed, opts, passes, reads[int(bad[0])], int(a[bad[0]]), int(ref[name][bad[0]]))
    for i, st in enumerate(stats):
        assert (st["processed"], st["aligned"]) == (int(ref["stats"][i][0]), int(ref["stats"][i][1])), (i, seed, opts, passes)
    # the port against the index-free scan for these policies, on a sample (per pass: the reads it claimed first)
    olibs = [model.Library(*libs.libs[k]) for k in LIB_ORDER]
    pick = [int(j) for j in rng.permutation(len(reads))[:400]]
    want = {}
    for pi, p in enumerate(passes):
        todo, qs = [], []
        for j in pick:
            r = reads[j]
            if j in want or not (p["min_len"] <= len(r) <= p["max_len"]):
                continue
            q = r
            if p["poly_t"]:
                tail = len(q) - len(q.rstrip("T"))
                if tail < 3 or len(q) - tail < 11:
                    continue
                q = q[:len(q) - tail]
            q = q[p["trim5"]:len(q) - p["trim3"]] if p["trim3"] else q[p["trim5"]:]
            if q:
                todo.append(j)
                qs.append(q)
        e, o, m = model.align_batch(olibs[p["lib"]], qs, p["seed_len"], p["max_mm_seed"], p["max_mm_total"])
        for k, j in enumerate(todo):
            if e[k] >= 0:
                want[j] = (pi, int(e[k]), int(o[k]), int(m[k]))
    for j in pick:
        mine = None if ref["pass_id"][j] < 0 else (int(ref["pass_id"][j]), int(ref["ref_id"][j]), int(ref["pos"][j]), int(ref["mm"][j]))
        assert mine == want.get(j), (reads[j], seed, passes)
    assert sum(int(st["aligned"]) for st in stats) > 0
