"""Size-independent properties of the GPU path at a size the oracle is not run at (millions of
reads): permutation invariance, idempotence, tally linearity and conservation, collapse round
trips.  (Equality with the CPU port at BASELINE's full 100 M reads is asserted by bench.py's
cpu_baseline leg before it prints a number.)"""
import numpy as np
import pytest

from mirge_amd import synth

pytestmark = pytest.mark.gpu

N = 4_000_000


@pytest.fixture(scope="module")
def big(native_lib):
    from mirge_amd.engine import Engine
    from mirge_amd.index import FmIndex
    libs = synth.SynthLibraries(seed=20181, scale=0.05)
    eng = Engine(0)
    for k in synth.LIB_KEYS:
        eng.add_library(k, FmIndex.build(*libs.libs[k]))
    words = synth.synth_reads_packed(libs, N, seed=99)[None, :]
    lens = np.full(N, 22, dtype=np.uint8)
    return eng, libs, words, lens


def run(eng, words, lens, quant=None):
    from mirge_amd.engine import ReadSet
    rs = ReadSet(words, lens, None, quant, device=eng.device)
    res = eng.cascade(rs, eng.mirge_passes())
    return rs, res


def test_permutation_invariance_and_idempotence(big):
    eng, libs, words, lens = big
    _, res = run(eng, words, lens)
    a = res.to_host()
    st = res.stats          # the counters live in the context: read them before the next run
    _, res2 = run(eng, words, lens)
    import copy
    stale = copy.copy(res)
    stale._stats = None
    with pytest.raises(RuntimeError):       # ... a later read would silently be the next run's
        stale.stats
    for x, y in zip(a, res2.to_host()):
        assert np.array_equal(x, y)                         # same input, same answer
    perm = np.random.default_rng(5).permutation(N)
    _, res3 = run(eng, np.ascontiguousarray(words[:, perm]), lens[perm])
    for x, y in zip(a, res3.to_host()):
        assert np.array_equal(x[perm], y)                   # a read's outcome does not depend on its neighbours
    order = np.argsort(words[0], kind="stable")             # the order mrg_collapse_run emits
    _, res4 = run(eng, np.ascontiguousarray(words[:, order]), lens[order])
    for x, y in zip(a, res4.to_host()):
        assert np.array_equal(x[order], y)
    assert sum(s["aligned"] for s in st) == int((a[0] >= 0).sum())
    assert st[0]["processed"] == N
    # every later pass is offered exactly what the earlier ones left (all reads are 22 nt)
    left = N
    for i, s in enumerate(st):
        if i in (1, 3):
            left -= s["aligned"]
            continue
        assert s["processed"] == left, i
        left -= s["aligned"]


def test_tally_linearity_and_conservation(big):
    from mirge_amd.engine import split_counts
    eng, libs, words, lens = big
    M = eng.indexes["mirna"].n_ref
    rng = np.random.default_rng(8)
    q1 = rng.integers(0, 50, (N, 2), dtype=np.uint32)
    q2 = rng.integers(0, 7, (N, 2), dtype=np.uint32)
    outs = []
    for q in (q1, q2, q1 + q2):
        rs, res = run(eng, words, lens, q)
        outs.append(eng.tally(rs, res, M).cpu().numpy().astype(np.int64))
    pass_id = res.to_host()[0]
    assert np.array_equal(outs[0][:-2] + outs[1][:-2], outs[2][:-2])     # all but trimmedUniq are sums
    quant, iscan, cat, uniq = split_counts(outs[2], M, 2, 9)
    tot = (q1 + q2).astype(np.int64)
    for p in range(9):
        want = tot[pass_id == p].sum(axis=0)
        assert np.array_equal(cat[p], want), p
    assert np.array_equal(quant.sum(axis=0), cat[0] + cat[8])            # miRNA bins = exact + isomiR reads
    assert np.array_equal(cat[9], tot[pass_id < 0].sum(axis=0))          # last row: unannotated reads
    assert int(cat.sum()) == int(tot.sum())


def test_collapse_round_trip(big):
    from mirge_amd import ingest
    eng, libs, words, lens = big
    sample = (np.arange(N) % 3).astype(np.uint16)
    col = ingest.collapse(eng, words, lens, None, sample, n_samples=3, max_len=22)
    u = col["words"][0]
    assert np.all(u[1:] > u[:-1])                                        # sorted, unique
    assert np.array_equal(col["quant"].sum(axis=0), np.bincount(sample, minlength=3))
    assert set(np.unique(words[0]).tolist()) == set(u.tolist())
    assert col["length_hist"] == {22: [int(x) for x in np.bincount(sample, minlength=3)]}
    again = ingest.collapse(eng, col["words"], col["lens"], None, None, n_samples=1, max_len=22)
    assert np.array_equal(again["words"], col["words"]) and int(again["quant"].sum()) == len(u)
    # annotating the uniques with their counts = annotating the raw reads
    M = eng.indexes["mirna"].n_ref
    rs_u, res_u = run(eng, col["words"], col["lens"], col["quant"])
    q_raw = np.zeros((N, 3), dtype=np.uint32)
    q_raw[np.arange(N), sample] = 1
    rs_r, res_r = run(eng, words, lens, q_raw)
    cu = eng.tally(rs_u, res_u, M).cpu().numpy()
    cr = eng.tally(rs_r, res_r, M).cpu().numpy()
    assert np.array_equal(cu[:-3], cr[:-3])                              # trimmedUniq counts uniques, the rest agrees


def test_pass_without_eligible_reads_is_skipped(big):
    """All reads are 22 nt, so the hairpin pass (len > 25) cannot be offered any: with the host's
    length hint it is not launched (zero counters, no LDS), and nothing else changes."""
    eng, libs, words, lens = big
    m = 500_000
    w, l = np.ascontiguousarray(words[:, :m]), lens[:m]
    _, res = run(eng, w, l)
    a = res.to_host()
    st = res.stats
    assert (st[1]["processed"], st[1]["aligned"], st[1]["lds_bytes"]) == (0, 0, 0) and st[1]["ms"] < 0.05
    # the same batch declared as "lengths unknown": the pass runs (and still claims nothing)
    from mirge_amd.engine import ReadSet
    rs = ReadSet(w, l, None, None, device=eng.device)
    rs.min_len, rs.max_len = 0, 255
    res2 = eng.cascade(rs, eng.mirge_passes())
    for x, y in zip(a, res2.to_host()):
        assert np.array_equal(x, y)
    st2 = res2.stats
    # (launched: as a match_kernel with its library staged in LDS, or as a unit of the seed launch)
    assert (st2[1]["lds_bytes"] > 0 or st2[1]["lds_mode"] in (8, 9)) and st2[1]["processed"] == 0
    for i in (0, 2, 3, 4, 5, 6, 7, 8):
        assert (st[i]["processed"], st[i]["aligned"]) == (st2[i]["processed"], st2[i]["aligned"])
