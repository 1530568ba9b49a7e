"""GPU parity (-m gpu): a one-mismatch pass on a library that looks like real ncRNA / mRNA sets, not like i.i.d.
text -- an interspersed element with 10^3 diverged copies and an exactly conserved core, poly-A tails, tandem
motifs -- where a seed's suffix interval holds 10^2..10^4 rows (`runAnnotationPipeline.py:581-584`: every unannotated
read is offered to every library, whatever the library holds).  Round 6: `wave_seed_kernel` answers such seeds from
POSITION LISTS (the rows of an overflowing seed bucket in text order: the walk stops at the first valid alignment)
plus a 64-ary search of the suffix-sorted rows for an exact occurrence, instead of verifying every row.  The answers
must not move: equal to the row-by-row path (`pos_scan` = 0), to the CPU port on every read, to the exhaustive scan
on a sample; the lists themselves equal the host restatement word for word."""
import numpy as np
import pytest

from oracle import model

pytestmark = pytest.mark.gpu


def rnd(rng, n):
    return "".join("ACGT"[c] for c in rng.integers(0, 4, n))


def mutate(rng, s, rate, keep=None):
    out = list(s)
    for i in range(len(out)):
        if keep and keep[0] <= i < keep[1]:
            continue
        if rng.random() < rate:
            out[i] = "ACGT"[("ACGT".index(out[i]) + 1 + int(rng.integers(0, 3))) % 4]
    return "".join(out)


ELEMENT_LEN, CORE = 220, (80, 140)


def make_library(rng, n_entries=6400):
    element = rnd(rng, ELEMENT_LEN)
    motifs = ["AC", "AGG", "TTAGGG", "CAG"]
    seqs = []
    for i in range(n_entries):
        s = rnd(rng, int(rng.integers(500, 1100)))
        if i % 3 == 0:          # ~2 100 copies of the element, 5 % divergence outside the core
            o = int(rng.integers(0, len(s) - ELEMENT_LEN))
            s = s[:o] + mutate(rng, element, 0.05, CORE) + s[o + ELEMENT_LEN:]
        if i % 29 == 0:         # poly-A tail
            s = s[:-int(rng.integers(20, 60))] + "A" * int(rng.integers(20, 60))
        if i % 31 == 0:         # tandem motif
            m = motifs[i % len(motifs)]
            span = int(rng.integers(60, 300))
            o = int(rng.integers(0, len(s) - span))
            s = s[:o] + (m * span)[:span] + s[o + span:]
        if i % 211 == 0:        # an N run: two segments
            o = int(rng.integers(100, len(s) - 100))
            s = s[:o] + "N" * int(rng.integers(1, 20)) + s[o:]
        seqs.append(s)
    seqs += [seqs[9], seqs[12][:400]]       # a duplicated entry, a prefix copy
    return ["rep%d" % i for i in range(len(seqs))], seqs, element


def make_reads(rng, seqs, element, n=16000):
    reads = []
    for _ in range(n):
        L = 22 if rng.random() < 0.7 else 23
        what = rng.random()
        if what < 0.25:         # from a FOREIGN copy of the element (as a read of another library would be): no exact hit, hundreds of one-mismatch ones
            src = mutate(rng, element, 0.05, CORE)
            o = int(rng.integers(0, ELEMENT_LEN - L))
            r = src[o:o + L]
        elif what < 0.35:       # inside / across the exactly conserved core: a thousand exact hits
            o = int(rng.integers(CORE[0] - 6, CORE[1] - L + 6))
            r = element[o:o + L]
        elif what < 0.50:       # from a copy that IS in the library (exact there, one mismatch in many others)
            s = seqs[3 * int(rng.integers(0, 2000))]
            o = int(rng.integers(0, len(s) - L))
            r = s[o:o + L]
        elif what < 0.62:       # one seed in the element, the other in unique sequence (one list walked, the other seed narrow)
            s = seqs[3 * int(rng.integers(0, 2000))]
            at = s.find(element[CORE[0]:CORE[0] + 11])
            if at < 12:
                continue
            r = s[at - 11:at - 11 + L] if rng.random() < 0.5 else (rnd(rng, 11) + element[CORE[0]:CORE[0] + L - 11])
        elif what < 0.70:       # poly-A and tandem reads
            r = ("A" * L) if rng.random() < 0.4 else (("TTAGGG" * 6)[int(rng.integers(0, 6)):][:L] if rng.random() < 0.5 else ("AC" * 12)[:L])
        elif what < 0.90:       # anywhere, 0..2 substitutions
            s = seqs[int(rng.integers(0, len(seqs)))]
            o = int(rng.integers(0, len(s) - L))
            r = list(s[o:o + L])
            for p in rng.integers(0, L, int(rng.integers(0, 3))):
                r[p] = "ACGT"[("ACGT".find(r[p]) + 1) % 4] if r[p] in "ACGT" else "A"
            r = "".join(r)
        else:
            r = rnd(rng, L)
        if "N" not in r and len(r) == L:
            reads.append(r)
    # one substitution in an element read, in either half (the clean seed is the other one)
    for _ in range(1500):
        o = int(rng.integers(0, ELEMENT_LEN - 23))
        r = list(element[o:o + 22])
        p = int(rng.integers(0, 22))
        r[p] = "ACGT"[("ACGT".index(r[p]) + 1) % 4]
        reads.append("".join(r))
    return list(dict.fromkeys(reads))


@pytest.fixture(scope="module")
def case(native_lib, oracle_lib):
    from mirge_amd import pack
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(2306)
    names, seqs, element = make_library(rng)
    total = sum(map(len, seqs))
    assert (1 << 22) < total < (1 << 24)           # beyond kDictSmallBases (its own index, not a seed unit's) and where a library gets seed buckets (k = 11)
    reads = make_reads(rng, seqs, element)
    w, l, nm = pack.pack_reads(reads)
    assert w.shape[0] == 1 and nm is None
    return dict(names=names, seqs=seqs, reads=reads, words=w, lens=l, ix=FmIndex.build(names, seqs),
                decoy=FmIndex.build(["decoy"], ["GATTACAGATTACAGGCCTTAAGGCCTTAACGCGCGTATATA" * 3]))


PLAN = [dict(lib="decoy", seed_len=28, max_mm_seed=0, max_mm_total=2),
        dict(lib="rep", seed_len=28, max_mm_seed=1, max_mm_total=2)]


def run(case, **opts):
    from mirge_amd.engine import Engine, ReadSet
    eng = Engine(0)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.add_library("rep", case["ix"])
    eng.add_library("decoy", case["decoy"])
    chk = eng.check_tables("rep")
    res = eng.cascade(ReadSet(case["words"], case["lens"], None, None, device=eng.device), eng.make_passes(PLAN))
    out = tuple(a.copy() for a in res.to_host()), [dict(s) for s in res.stats], chk, eng.library_dict_stats("rep")
    eng.close()
    return out


def test_position_lists_answer_as_the_row_by_row_path_the_port_and_the_scan(case):
    got, stats, chk, _ = run(case)
    assert chk["seed_buckets"] == 0 and chk["jump_tables"] == 0 and chk["wide_rows"] == 0, chk   # buckets, their headers, the lists
    assert stats[1]["lds_mode"] == 9 and stats[1]["variant"] in (1, 2, 5, 6), stats[1]             # wave_seed_kernel, a unit with buckets
    old, old_stats, _, _ = run(case, pos_scan=0)
    for a, b in zip(got, old):
        assert np.array_equal(a, b)
    assert (stats[1]["processed"], stats[1]["aligned"]) == (old_stats[1]["processed"], old_stats[1]["aligned"])
    # what the walk saves: rows looked at (the row-by-row path verifies every row of a wide interval)
    assert stats[1]["candidates"] * 2 < old_stats[1]["candidates"], (stats[1]["candidates"], old_stats[1]["candidates"])
    # without the lists at all (pos_lists = 0 before add_library): the headers stay empty, same answers
    bare, _, chk0, _ = run(case, pos_lists=0)
    assert chk0["seed_buckets"] == 0
    for a, b in zip(got, bare):
        assert np.array_equal(a, b)
    # the CPU port on every read
    views = [case["ix"].view(), case["decoy"].view()]
    pd = [dict(lib=1, min_len=0, max_len=255, seed_len=28, max_mm_seed=0, max_mm_total=2, trim5=0, trim3=0, poly_t=0),
          dict(lib=0, min_len=0, max_len=255, seed_len=28, max_mm_seed=1, max_mm_total=2, trim5=0, trim3=0, poly_t=0)]
    ref = model.fm_cascade(views, pd, case["words"], case["lens"], None, ftab=True)
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
        assert np.array_equal(a, ref[name]), name
    pass_id, ref_id, pos, mm = got
    assert (pass_id == 1).sum() > 6000 and (mm[pass_id == 1] == 1).sum() > 1500 and (mm[pass_id == 1] == 0).sum() > 1500
    # the exhaustive scan (no index at all) on a sample
    lib = model.Library(case["names"], case["seqs"])
    pick = np.random.default_rng(5).choice(len(case["reads"]), 500, replace=False)
    want_ref, want_pos, want_mm = model.align_batch(lib, [case["reads"][i] for i in pick], 28, 1, 2)
    for j, i in enumerate(pick):
        g = (int(ref_id[i]), int(pos[i]), int(mm[i])) if pass_id[i] == 1 else (-1, -1, -1)
        assert g == (int(want_ref[j]), int(want_pos[j]), int(want_mm[j])), case["reads"][i]


def test_position_lists_with_host_built_tables(case):
    """device_tables = 0: buckets filled on the host, lists and headers added on the device; same answers."""
    a, _, chk, _ = run(case)
    b, _, chk_h, _ = run(case, device_tables=0)
    assert chk_h["seed_buckets"] == 0, chk_h
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_a_full_walk_buffer_falls_back_to_the_row_by_row_path(case):
    """walk_cap = 64: a wave may leave ONE batch of reads behind its stream, every later wide seed is verified row by row
    (and a dictionary fallback's wide interval by its lane) as before round 6 -- on a grid of a hundredth of the
    workgroups, so that a wave certainly meets more than 64 such reads whatever order the pass in front leaves them in;
    walk_cap = 0: nothing is left behind at all.  Same answers."""
    a, st_a, _, _ = run(case)
    for cap, pct in ((64, 1), (64, 100), (0, 100)):
        ref_st = st_a if pct == 100 else run(case, grid_pct=pct)[1]
        b, st_b, _, _ = run(case, walk_cap=cap, grid_pct=pct)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (cap, pct)
        assert st_b[1]["candidates"] >= ref_st[1]["candidates"], (cap, pct)
        if (cap, pct) != (64, 100):
            assert st_b[1]["candidates"] > ref_st[1]["candidates"], (cap, pct)     # (more rows looked at: the fallback ran)
