"""GPU (-m gpu): the long-read lane (mrg_cascade_run_long, csrc/long_reads.hip) -- the cascade for reads the packed
batches cannot describe (beyond 255 nt; runAnnotationPipeline.py:543-554 offers a read of ANY length to every pass) --
against the oracle's exhaustive scan (oracle.cascade.scan_cascade: no index at all) on every read, and against the
packed cascade on reads both lanes can take."""
import numpy as np
import pytest

from mirge_amd import pack
from mirge_amd.engine import V_MODE_SEED
from mirge_amd.index import FmIndex

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand_seq(rng, n):
    return ACGT[rng.integers(0, 4, n)].tobytes().decode()


def _world(seed, n_reads=700, n_in_lib=False):
    """Three libraries (entries of 40..3000 nt, a few paralogs, optionally entries with N) and reads of 20..1500 nt cut
    from them with 0..3 substitutions, some with N, some with poly-T tails, some random, some low-complexity."""
    rng = np.random.default_rng(seed)
    libs = {}
    for key, (n_entries, lo, hi) in (("short", (60, 40, 400)), ("mid", (40, 200, 1200)), ("long", (25, 800, 3000))):
        seqs = [_rand_seq(rng, int(rng.integers(lo, hi))) for _ in range(n_entries)]
        for _ in range(n_entries // 5):   # paralogs: a copy with a few substitutions
            s = list(seqs[int(rng.integers(0, n_entries))])
            for _ in range(int(rng.integers(1, 4))):
                s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
            seqs.append("".join(s))
        if n_in_lib:
            for k in range(0, len(seqs), 7):
                s = list(seqs[k])
                s[len(s) // 2] = "N"
                seqs[k] = "".join(s)
        seqs.append("A" * 700 + _rand_seq(rng, 50))   # low complexity: wide seed intervals
        libs[key] = (["%s_%d" % (key, i) for i in range(len(seqs))], seqs)
    reads = []
    keys = list(libs)
    for _ in range(n_reads):
        kind = rng.random()
        if kind < 0.1:
            r = _rand_seq(rng, int(rng.integers(20, 900)))
        elif kind < 0.13:
            r = "A" * int(rng.integers(256, 700))
        else:
            seqs = libs[keys[int(rng.integers(0, 3))]][1]
            s = seqs[int(rng.integers(0, len(seqs)))].replace("N", "A")
            ln = int(min(len(s), rng.choice([20, 30, 64, 200, 256, 257, 300, 500, 1000, 1500]) + rng.integers(0, 9)))
            o = int(rng.integers(0, len(s) - ln + 1))
            r = list(s[o:o + ln])
            for _ in range(int(rng.choice([0, 0, 1, 1, 2, 3]))):
                r[int(rng.integers(0, ln))] = "ACGTN"[int(rng.integers(0, 5 if rng.random() < 0.2 else 4))]
            r = "".join(r)
            if rng.random() < 0.2:
                r = r.rstrip("T") + "T" * int(rng.integers(2, 9))
            if rng.random() < 0.15:   # flanks the `-5` / `-3` passes cut off again
                r = _rand_seq(rng, int(rng.integers(0, 40))) + r + _rand_seq(rng, int(rng.integers(0, 4)))
        reads.append(r)
    return libs, list(dict.fromkeys(reads)), rng


def _random_passes(rng, keys, n_pass):
    out = []
    for _ in range(n_pass):
        mode_v = rng.random() < 0.5
        mms = int(rng.integers(0, 3))
        p = dict(lib=keys[int(rng.integers(0, len(keys)))], seed_len=V_MODE_SEED if mode_v else int(rng.choice([20, 28, 32, 40])),
                 max_mm_seed=mms, max_mm_total=mms if mode_v else int(rng.integers(mms, 4)),
                 trim5=int(rng.choice([0, 0, 1, 3, 35])), trim3=int(rng.choice([0, 0, 2, 3])),
                 min_len=int(rng.choice([0, 0, 26, 300])), max_len=int(rng.choice([255, 255, 255, 600, 25])),
                 poly_t=int(rng.random() < 0.2))
        if p["max_len"] == 600:
            p["max_len"] = 255   # (only "< 255" bounds bind; 255 and more is unbounded)
        out.append(p)
    # (a last pass every read is offered to: a drawn table may exclude every read by its length windows)
    out.append(dict(lib=keys[0], seed_len=28, max_mm_seed=1, max_mm_total=2, trim5=0, trim3=0, min_len=0, max_len=255, poly_t=0))
    out.insert(len(out) // 2, dict(lib=keys[-1], seed_len=V_MODE_SEED, max_mm_seed=2, max_mm_total=2, trim5=0, trim3=0, min_len=0,
                                  max_len=255, poly_t=0))
    return out


@pytest.mark.parametrize("seed", list(range(8)))
def test_long_lane_equals_exhaustive_scan(native_lib, oracle_lib, seed):
    """Random libraries x random pass tables (-n / -v, 0..2 seed mismatches, trims up to 35, poly-T rules, length
    windows) x reads of 20..1500 nt: assignments (pass, entry, offset, mismatches) and per-pass processed / aligned
    of mrg_cascade_run_long equal the exhaustive scan's on every read."""
    from mirge_amd.engine import Engine
    from oracle import cascade as ocas, model
    libs, reads, rng = _world(seed, n_in_lib=(seed % 2 == 1))
    eng = Engine(0)
    olibs = {}
    for k, (names, seqs) in libs.items():
        eng.add_library(k, FmIndex.build(names, seqs))
        olibs[k] = model.Library(names, seqs)
    rows = _random_passes(rng, list(libs), int(rng.integers(2, 9)))
    got = eng.cascade_long(reads, eng.make_passes(rows))
    want = ocas.scan_cascade(olibs, rows, reads)
    for name, a, b in zip(("pass_id", "ref_id", "pos", "mm"), got[:4], want[:4]):
        bad = np.nonzero(np.asarray(a) != np.asarray(b))[0]
        assert bad.size == 0, (name, rows, [(reads[i][:40], len(reads[i]), a[i], b[i]) for i in bad[:5]])
    assert [(s["processed"], s["aligned"]) for s in got[4]] == [tuple(c) for c in want[4]]
    assert sum(s["aligned"] for s in got[4]) > 50 and sum(1 for r, p in zip(reads, got[0]) if len(r) > 255 and p >= 0) > 20
    assert sum(s["steps"] for s in got[4]) > 0 and sum(s["candidates"] for s in got[4]) > 0


def test_long_lane_equals_the_packed_cascade_on_reads_both_take(native_lib):
    """Reads of 16..255 nt (with N, poly-T tails) through mrg_cascade_run_long and through mrg_cascade_run with the
    reference's nine passes: the same four arrays and the same per-pass processed / aligned; and the long lane's
    counters ADD to a count vector / stats list that is handed in."""
    import torch
    from mirge_amd.engine import Engine, ReadSet
    from tests.util import World
    w = World(scale=0.03, n_fixed=1500, n_var=1500, with_n=True, max_var_len=44)
    rng = np.random.default_rng(3)
    for key in ("ncrna_others", "mrna", "rrna", "snorna", "hairpin") * 60:   # reads of four and eight words too
        s = w.libs.libs[key][1][int(rng.integers(0, len(w.libs.libs[key][1])))]
        ln = int(min(len(s), rng.integers(33, 256)))
        o = int(rng.integers(0, len(s) - ln + 1))
        r = list(s[o:o + ln])
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, ln))] = "ACGTN"[int(rng.integers(0, 5))]
        w.reads.append("".join(r))
    w.reads = list(dict.fromkeys(w.reads))
    w.words, w.lens, w.nmask = pack.pack_reads(w.reads)
    eng = Engine(0)
    for k in w.index:
        eng.add_library(k, w.index[k])
    passes = eng.mirge_passes()
    res = eng.cascade(ReadSet(w.words, w.lens, w.nmask, None, device=eng.device), passes)
    a = res.to_host()
    stats = [dict(s) for s in res.stats]
    pc = res.pass_counts.clone()
    b = eng.cascade_long(w.reads, passes, pass_counts=pc, stats=stats)
    for x, y in zip(a, b[:4]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert [(s["processed"], s["aligned"]) for s in b[4]] == [(s["processed"], s["aligned"]) for s in res.stats]
    assert torch.equal(pc, 2 * res.pass_counts)
    assert all(s2["processed"] == 2 * s1["processed"] and s2["aligned"] == 2 * s1["aligned"] for s1, s2 in zip(res.stats, stats))
    assert (a[0] >= 0).sum() > 500 and w.lens.max() > 128


def test_long_lane_edge_cases(native_lib, oracle_lib):
    """Nothing to do, a read longer than every entry, a read that IS an entry, reads shorter than the seed mismatches,
    a read that is all T, an entry reached only across a reference N (never), 100 000-nt reads."""
    from mirge_amd.engine import Engine
    from oracle import cascade as ocas, model
    rng = np.random.default_rng(5)
    big = _rand_seq(rng, 120_000)
    names, seqs = ["e0", "e1", "e2", "e3"], [_rand_seq(rng, 300), big, _rand_seq(rng, 500) + "N" + _rand_seq(rng, 500), "ACGT" * 100]
    eng = Engine(0)
    eng.add_library("lib", FmIndex.build(names, seqs))
    olibs = {"lib": model.Library(names, seqs)}
    rows = [dict(lib="lib", seed_len=28, max_mm_seed=1, max_mm_total=2, trim5=0, trim3=0, min_len=0, max_len=255, poly_t=0),
            dict(lib="lib", seed_len=V_MODE_SEED, max_mm_seed=2, max_mm_total=2, trim5=1, trim3=2, min_len=0, max_len=255, poly_t=0),
            dict(lib="lib", seed_len=V_MODE_SEED, max_mm_seed=0, max_mm_total=0, trim5=0, trim3=0, min_len=0, max_len=255, poly_t=1)]
    passes = eng.make_passes(rows)
    got = eng.cascade_long([], passes)
    assert all(len(x) == 0 for x in got[:4]) and all(s["processed"] == 0 for s in got[4])
    mut = list(big[5:100_005])
    mut[70_000] = "A" if mut[70_000] != "A" else "C"
    mut[99_000] = "A" if mut[99_000] != "A" else "C"
    reads = [seqs[0], big, big[5:100_005], "".join(mut), big + "A", seqs[2][400:600], seqs[2][:500], "T" * 400, "A", "",
             "AC", seqs[3][3:303], big[1000:1300] + "TTTT", "G" + seqs[0] + "CA",
             "A" * 300, "A" * 28, "C" * 300]   # (a poly-A seed's interval starts at the sentinel's row, which lies in no segment)
    got = eng.cascade_long(reads, passes)
    want = ocas.scan_cascade(olibs, rows, reads)
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(np.asarray(a), np.asarray(b)), (a, b)
    assert [(s["processed"], s["aligned"]) for s in got[4]] == [tuple(c) for c in want[4]]
    assert list(got[0][:4]) == [0, 0, 0, 0] and list(got[3][:4]) == [0, 0, 0, 2] and list(got[2][:3]) == [0, 0, 5]
    assert got[0][4] == 1 and got[0][5] == -1 and got[0][13] == 1 and got[1][11] == 3
