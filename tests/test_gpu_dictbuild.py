"""GPU parity (-m gpu): the exact-match dictionary of a LARGE library filled on the device (csrc/dictbuild.hip) answers
every read as the host-built one does (csrc/dict_index.cpp) and as the exhaustive scan does -- on a library with what
makes dictionaries hard: an element with thousands of copies (chains overflow into the FM index), duplicated entries,
N runs (segments), keys that end at a segment's last base."""
import numpy as np
import pytest

from oracle import model

pytestmark = pytest.mark.gpu


def rnd(rng, n):
    return "".join("ACGT"[c] for c in rng.integers(0, 4, n))


def make_library(rng, n_entries=5200, lo=700, hi=1100):
    element = rnd(rng, 34)
    seqs = []
    for i in range(n_entries):
        s = rnd(rng, int(rng.integers(lo, hi)))
        if i % 2 == 0:        # the element, 2 600 copies: its 16-mers overflow every chain
            o = int(rng.integers(0, len(s) - 40))
            s = s[:o] + element + s[o + 34:]
        if i % 97 == 0:       # an N run: two segments
            o = int(rng.integers(100, len(s) - 100))
            s = s[:o] + "N" * int(rng.integers(1, 40)) + s[o:]
        seqs.append(s)
    seqs += [seqs[7], seqs[7], seqs[11][:300]]      # duplicated entries and a prefix copy
    names = ["big%d" % i for i in range(len(seqs))]
    return names, seqs, element


def make_reads(rng, seqs, element, n=12000):
    reads = []
    for _ in range(n):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(16, 33))
        what = rng.random()
        if what < 0.55:         # a substring somewhere
            o = int(rng.integers(0, len(s) - L))
            r = s[o:o + L]
        elif what < 0.65:       # ... ending at the entry's last base / starting at its first
            r = s[len(s) - L:] if rng.random() < 0.5 else s[:L]
        elif what < 0.80:       # inside / across the repeated element
            k = int(rng.integers(0, 12))
            r = (element[k:] + rnd(rng, 32))[:L] if rng.random() < 0.5 else element[k:k + min(L, 34 - k)]
        elif what < 0.90:       # one or two substitutions
            o = int(rng.integers(0, len(s) - L))
            r = list(s[o:o + L])
            for p in rng.integers(0, L, int(rng.integers(1, 3))):
                r[p] = "ACGT"[("ACGT".find(r[p]) + 1) % 4] if r[p] in "ACGT" else "A"
            r = "".join(r)
        else:
            r = rnd(rng, L)
        if "N" not in r and len(r) >= 16:
            reads.append(r)
    return list(dict.fromkeys(reads))


def test_device_built_dictionary_equals_the_host_built_one_and_the_scan(native_lib, oracle_lib):
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(41)
    names, seqs, element = make_library(rng)
    assert sum(map(len, seqs)) > (1 << 22)          # beyond kDictSmallBases: the parallel builds
    ix = FmIndex.build(names, seqs)
    decoy = FmIndex.build(["decoy"], ["GATTACAGATTACAGGCCTTAAGGCCTTAACGCGCGTATATA" * 3])
    reads = make_reads(rng, seqs, element)
    w, l, nm = pack.pack_reads(reads)
    assert w.shape[0] == 1 and nm is None
    results, stats = {}, {}
    for dev_tables in (1, 0):
        eng = Engine(0)
        eng.set_option("device_tables", dev_tables)
        eng.add_library("big", ix, exact_dict=True)
        eng.add_library("decoy", decoy)
        stats[dev_tables] = eng.library_dict_stats("big")
        # jump tables, row context, wide rows (and seed buckets, where a library has them): the device's = the host's
        chk = eng.check_tables("big")
        assert chk["jump_tables"] == 0 and chk["row_context"] == 0 and chk["wide_rows"] == 0 and chk["seed_buckets"] in (0, None), (dev_tables, chk)
        first = dict(lib="decoy", seed_len=28, max_mm_seed=0, max_mm_total=2)
        for (seed_len, mm_total, t5, t3) in ((28, 2, 0, 0), (1024, 0, 0, 0), (28, 2, 1, 2)):
            pol = dict(lib="big", seed_len=seed_len, max_mm_seed=0, max_mm_total=mm_total, trim5=t5, trim3=t3)
            for plan in ([pol], [first, pol]):      # exact_dict_kernel, and a dictionary unit of a seed launch
                res = eng.cascade(ReadSet(w, l, None, None, device=eng.device), eng.make_passes(plan))
                results[(dev_tables, seed_len, t5, len(plan))] = tuple(a.copy() for a in res.to_host())
        eng.close()
    # (positions stored, homes whose chain overflowed: the layout -- and so which home overflows first -- depends on the
    # insertion order, what a lookup finds does not)
    assert stats[1][1] > 0 and stats[0][1] > 0
    assert abs(stats[1][0] - stats[0][0]) < 0.01 * stats[0][0]
    for key, got in results.items():
        if key[0] != 1:
            continue
        want = results[(0,) + key[1:]]
        for a, b in zip(got, want):
            assert np.array_equal(a, b), key
    # ... and both equal the exhaustive scan (a sample: the scan reads 4.6 Mbp per read)
    lib = model.Library(names, seqs)
    pick = rng.choice(len(reads), 400, replace=False)
    sub = [reads[i] for i in pick]
    for (seed_len, mm_total, t5, t3) in ((28, 2, 0, 0), (1024, 0, 0, 0), (28, 2, 1, 2)):
        trimmed = [r[t5:len(r) - t3] if t3 else r[t5:] for r in sub]
        want_ref, want_pos, want_mm = model.align_batch(lib, trimmed, seed_len, 0, mm_total)
        for n_plan in (1, 2):
            pass_id, ref_id, pos, mm = results[(1, seed_len, t5, n_plan)]
            for j, i in enumerate(pick):
                got = (int(ref_id[i]), int(pos[i]), int(mm[i])) if pass_id[i] == n_plan - 1 else (-1, -1, -1)
                assert got == (int(want_ref[j]), int(want_pos[j]), int(want_mm[j])), (reads[i], seed_len, t5, n_plan)


def test_device_built_tables_of_a_library_with_seed_buckets(native_lib, oracle_lib):
    """An 11 Mbp-class library (here 2.2 Mbp .. the size where an 11-mer has 0.25-4 rows) gets seed buckets: jump tables
    with k = 11 / 12, row context, wide rows and buckets from libtables.hip equal the host's word for word; a library
    with N runs (text positions that sort out of code order) and a tiny one beside it."""
    from mirge_amd.engine import Engine
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(43)
    seqs = [rnd(rng, int(rng.integers(300, 3000))) for _ in range(1400)]
    for i in range(0, len(seqs), 50):          # N runs, also at the very start / end of an entry
        s = seqs[i]
        o = int(rng.integers(0, len(s) - 20))
        seqs[i] = ("NN" if i % 100 == 0 else "") + s[:o] + "N" * int(rng.integers(1, 30)) + s[o:] + ("N" if i % 150 == 0 else "")
    seqs += ["A" * 500, "ACGT" * 100, seqs[3]]
    names = ["e%d" % i for i in range(len(seqs))]
    assert (1 << 20) <= sum(map(len, seqs)) < (1 << 22)
    ix = FmIndex.build(names, seqs)
    eng = Engine(0)
    eng.add_library("mid", ix)
    eng.add_library("tiny", FmIndex.build(["t"], ["GATTACA" * 30]))
    chk = eng.check_tables("mid")
    assert chk == {"jump_tables": 0, "row_context": 0, "wide_rows": 0, "seed_buckets": 0}, chk
    assert eng.check_tables("tiny")["jump_tables"] == 0
    eng.close()


def test_device_built_dictionary_with_a_shorter_key(native_lib, oracle_lib):
    """`dict_key` 12 (the option's range is 8..16): positions whose 12-base keys collide far more often -- longer chains,
    more overflowed homes -- built on the device and on the host; same answers for 16..32-nt reads."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(47)
    names, seqs, element = make_library(rng, n_entries=4800, lo=850, hi=1000)
    assert sum(map(len, seqs)) > (1 << 22)
    ix = FmIndex.build(names, seqs)
    reads = make_reads(rng, seqs, element, n=6000)
    w, l, nm = pack.pack_reads(reads)
    got = {}
    for dev_tables in (1, 0):
        eng = Engine(0)
        eng.set_option("device_tables", dev_tables)
        eng.set_option("dict_key", 12)
        eng.add_library("big", ix, exact_dict=True)
        assert eng.library_dict_stats("big")[0] > 4_000_000
        res = eng.cascade(ReadSet(w, l, None, None, device=eng.device),
                          eng.make_passes([dict(lib="big", seed_len=28, max_mm_seed=0, max_mm_total=2)]))
        got[dev_tables] = tuple(a.copy() for a in res.to_host())
        eng.close()
    for a, b in zip(got[1], got[0]):
        assert np.array_equal(a, b)
    lib = model.Library(names, seqs)
    pick = rng.choice(len(reads), 200, replace=False)
    want_ref, want_pos, want_mm = model.align_batch(lib, [reads[i] for i in pick], 28, 0, 2)
    pass_id, ref_id, pos, mm = got[1]
    for j, i in enumerate(pick):
        g = (int(ref_id[i]), int(pos[i]), int(mm[i])) if pass_id[i] == 0 else (-1, -1, -1)
        assert g == (int(want_ref[j]), int(want_pos[j]), int(want_mm[j])), reads[i]


def test_seed_unit_over_a_megabase_library_without_device_tables(native_lib, oracle_lib):
    """Advisor, round 5: a one-mismatch pass on a library of 1..4 Mbp is searched through a seed unit whose index
    `mrg_cascade_run` builds itself; with `device_tables` = 0 that index must get its jump tables and row context from the
    host before the upload (they used to be left empty: out-of-bounds table reads).  Both settings answer alike, and as
    the exhaustive scan does."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(47)
    seqs = [rnd(rng, int(rng.integers(400, 2400))) for _ in range(900)]
    names = ["m%d" % i for i in range(len(seqs))]
    assert (1 << 20) <= sum(map(len, seqs)) < (1 << 22)
    ix = FmIndex.build(names, seqs)
    decoy = FmIndex.build(["decoy"], ["GATTACAGATTACAGGCCTTAAGGCCTTAACGCGCGTATATA" * 3])
    reads = []
    for _ in range(6000):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(18, 31))
        o = int(rng.integers(0, len(s) - L))
        r = list(s[o:o + L])
        for p in rng.integers(0, L, int(rng.integers(0, 3))):
            r[p] = "ACGT"[("ACGT".find(r[p]) + 1) % 4]
        reads.append("".join(r) if rng.random() < 0.85 else rnd(rng, L))
    reads = list(dict.fromkeys(reads))
    w, l, nm = pack.pack_reads(reads)
    assert w.shape[0] == 1 and nm is None
    plan = [dict(lib="decoy", seed_len=28, max_mm_seed=0, max_mm_total=2),
            dict(lib="mid", seed_len=28, max_mm_seed=1, max_mm_total=2)]
    results = {}
    for dev_tables in (0, 1):
        eng = Engine(0)
        eng.set_option("device_tables", dev_tables)
        eng.add_library("mid", ix)
        eng.add_library("decoy", decoy)
        res = eng.cascade(ReadSet(w, l, None, None, device=eng.device), eng.make_passes(plan))
        assert res.stats[1]["lds_mode"] in (8, 9, 10), res.stats[1]      # a seed launch served the pass
        results[dev_tables] = tuple(a.copy() for a in res.to_host())
        eng.close()
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
    lib = model.Library(names, seqs)
    pick = rng.choice(len(reads), 300, replace=False)
    want_ref, want_pos, want_mm = model.align_batch(lib, [reads[i] for i in pick], 28, 1, 2)
    pass_id, ref_id, pos, mm = results[0]
    for j, i in enumerate(pick):
        got = (int(ref_id[i]), int(pos[i]), int(mm[i])) if pass_id[i] == 1 else (-1, -1, -1)
        assert got == (int(want_ref[j]), int(want_pos[j]), int(want_mm[j])), reads[i]
    assert (pass_id == 1).sum() > 1000
