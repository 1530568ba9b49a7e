"""GPU parity (-m gpu): the HIP path, called through the C-ABI, against the
oracle on the same seeded inputs.  Bit-exact: every array is integer."""
import os

import numpy as np
import pytest

from oracle import cascade, model
from mirge_amd.engine import DEFAULT_WSTOP
from tests.util import LIB_ORDER, World

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(native_lib, oracle_lib):
    return World(scale=0.05, n_fixed=20000, n_var=3000)


@pytest.fixture(scope="module")
def engine(world):
    from mirge_amd.engine import Engine
    eng = Engine(0)
    for k in LIB_ORDER:
        eng.add_library(k, world.index[k])
    return eng


def run_gpu(engine, world, quant=None, **opts):
    from mirge_amd.engine import ReadSet
    # the match_kernel variants are exercised one launch per pass; the fused launches have their
    # own test below (and are what every default-option test in this file runs)
    opts.setdefault("fuse", 0)
    opts.setdefault("pair_seeds", 0)   # (the anchor-pair search of the 2-mismatch pass: test_pair_seeds_*)
    opts.setdefault("dict", 0)         # (the dictionary kernels of one-word batches: tests/test_gpu_dict.py)
    for k, v in opts.items():
        engine.set_option(k, v)
    rs = ReadSet(world.words, world.lens, world.nmask, quant, device=engine.device)
    res = engine.cascade(rs, engine.mirge_passes())
    return rs, res


def assert_same(res, ref):
    got = res.to_host()
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
        assert np.array_equal(a, ref[name]), name
    for i, st in enumerate(res.stats):
        assert st["processed"] == int(ref["stats"][i][0])
        assert st["aligned"] == int(ref["stats"][i][1])
        assert st["steps"] == int(ref["stats"][i][2])
        assert st["candidates"] == int(ref["stats"][i][3])
        assert st["lookups"] == int(ref["stats"][i][4])


def test_fused_launches_match_cpu_port(engine, world):
    """fuse = 1/2/3: consecutive passes share one walk of the survivor list (fused_kernel); every
    assignment and all five per-pass counters equal the port's, which filters seed pieces with the
    same folded 9-mer bitmaps (mrg_pass_stats.kbits_log2) the launch staged."""
    base = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask,
                            wstop=DEFAULT_WSTOP, ftab=True)
    seen_groups = set()
    for fuse, wstop, ftab, wide in ((1, DEFAULT_WSTOP, 1, 64), (2, DEFAULT_WSTOP, 1, 64), (3, 0, 0, 64),
                                    (1, 2, 1, 1)):
        _, res = run_gpu(engine, world, fuse=fuse, wstop=wstop, ftab=ftab, wide_rows=wide)
        st = res.stats
        passes = [dict(p, kbits_log2=st[i]["kbits_log2"]) for i, p in enumerate(world.passes)]
        ref = model.fm_cascade(world.views, passes, world.words, world.lens, world.nmask, wstop=wstop,
                               ftab=bool(ftab))
        for k in ("pass_id", "ref_id", "pos", "mm"):
            assert np.array_equal(base[k], ref[k])  # the filter size never changes an assignment
        assert_same(res, ref)
        groups = [s["group"] for s in st]
        seen_groups.add(tuple(groups))
        assert groups[0] == 0 and groups[8] == 8            # first pass and the 2-mismatch pass run alone
        assert groups[2] == groups[3] == groups[4] == groups[5] == 1   # hairpin .. rRNA share a launch
        assert all(s["lds_mode"] == 4 for s in st[1:6])
        assert (groups[6] == 1) == (fuse == 2)
        assert (groups[7] == groups[6]) == (fuse != 3)
    assert len(seen_groups) == 3
    engine.set_option("fuse", 1)
    engine.set_option("wide_rows", 64)
    engine.set_option("wstop", DEFAULT_WSTOP)
    engine.set_option("ftab", 1)


def test_cascade_matches_cpu_port_all_lds_modes(engine, world):
    ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask)
    # every kernel variant: nothing staged, occ blocks, blocks + text, text only, and the
    # driver's own residency choice
    for mode in (0, 1, 2, 3, -1):
        _, res = run_gpu(engine, world, force_lds_mode=mode, wstop=0, ftab=0)
        assert_same(res, ref)
        staged = [s["lds_bytes"] for s in res.stats]
        if mode == 0:
            assert max(staged) == 0
        elif mode > 0:
            assert staged[0] > 0 and staged[8] > 0  # the miRNA library always fits
    engine.set_option("force_lds_mode", -1)
    # without the 9-mer presence bitmap: same assignments, more jump-table loads (the port without it)
    ref_nf = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask, kmer_filter=False)
    _, res = run_gpu(engine, world, kmer_filter=0, wstop=0, ftab=0)
    assert_same(res, ref_nf)
    engine.set_option("kmer_filter", 1)
    # the LDS budget option alone can also switch staging off
    _, res = run_gpu(engine, world, lds_budget=0, wstop=0, ftab=0)
    assert_same(res, ref)
    assert max(s["lds_bytes"] for s in res.stats) == 0
    engine.set_option("lds_budget", 160 * 1024)


def test_search_shortcuts_same_results_fewer_steps(engine, world):
    """The k-mer jump table and the early hand-over to verification change how many
    LF steps run, never an assignment; step / candidate / lookup counts equal the port's."""
    base = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask)
    for wstop, ftab, wide in ((0, 1, 256), (2, 0, 256), (2, 1, 256), (16, 1, 256), (16, 1, 1), (2, 1, 3)):
        ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask,
                               wstop=wstop, ftab=bool(ftab))
        for k in ("pass_id", "ref_id", "pos", "mm"):
            assert np.array_equal(base[k], ref[k])
        # wide_rows = 1 / 3 pushes nearly every interval through the wave-wide verification
        _, res = run_gpu(engine, world, wstop=wstop, ftab=ftab, wide_rows=wide)
        assert_same(res, ref)
        assert sum(s["steps"] for s in res.stats) < int(base["stats"][:, 2].sum())
    engine.set_option("wstop", DEFAULT_WSTOP)
    engine.set_option("ftab", 1)
    engine.set_option("wide_rows", 64)


def test_cascade_matches_exhaustive_scan(engine, world):
    """Against the independent model of bowtie's rules (subset, it is O(N*text))."""
    _, res = run_gpu(engine, world, lds_budget=160 * 1024)
    pass_id, ref_id, pos, mm = res.to_host()
    sub = np.random.default_rng(5).choice(len(world.reads), 4000, replace=False)
    reads = [world.reads[i] for i in sub]
    libs = {k: model.Library(*world.libs.libs[k]) for k in LIB_ORDER}
    seq_dic = {r: cascade.new_seq_record(r, 1) for r in reads}
    log_dic = {"quantStats": [{}], "annotStats": []}
    align = {}
    cascade.run_annotation_pipeline(seq_dic, libs, log_dic, align_dic=align)
    for i, r in zip(sub, reads):
        got = None if pass_id[i] < 0 else (int(pass_id[i]), int(ref_id[i]), int(pos[i]), int(mm[i]))
        assert align.get(r) == got, r


def test_pair_seeds_match_port_and_exhaustive_scan(engine, world):
    """The 2-mismatch pass through the six anchor pairs (reads whose seed region holds four 4-base
    anchors) and through the pigeonhole pieces (shorter ones) in one stratum_kernel launch: every
    assignment equals the piece-only search's, the five counters equal the port's, which rebuilds
    the pair tables from the suffix array and the text, and a sample equals the exhaustive scan."""
    base = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask,
                            wstop=DEFAULT_WSTOP, ftab=True)
    for fuse in (0, 1):
        _, res = run_gpu(engine, world, pair_seeds=1, fuse=fuse, wstop=DEFAULT_WSTOP, ftab=1)
        st = res.stats
        assert st[8]["pair_anchor"] == 4 and st[8]["lds_mode"] in (5, 6) and st[8]["n_launches"] == 1
        assert all(s["pair_anchor"] == 0 for s in st[:8])
        passes = [dict(p, kbits_log2=st[i]["kbits_log2"], pair_anchor=st[i]["pair_anchor"])
                  for i, p in enumerate(world.passes)]
        ref = model.fm_cascade(world.views, passes, world.words, world.lens, world.nmask, wstop=DEFAULT_WSTOP,
                               ftab=True)
        for k in ("pass_id", "ref_id", "pos", "mm"):
            assert np.array_equal(base[k], ref[k]), k   # the pair search never changes an assignment
        assert_same(res, ref)
        # reads of >= 19 nt (16 seed bases after -5 1 -3 2) through 4-base anchors, the 16..18-nt
        # ones through 3-base anchors: no read of this world is left to the pieces, no LF step
        assert int(ref["stats"][8][2]) == 0 and int(ref["stats"][8][4]) > int(base["stats"][8][4]) // 4
    pass_id, ref_id, pos, mm = res.to_host()
    sub = np.random.default_rng(6).choice(len(world.reads), 3000, replace=False)
    reads = [world.reads[i] for i in sub]
    libs = {k: model.Library(*world.libs.libs[k]) for k in LIB_ORDER}
    seq_dic = {r: cascade.new_seq_record(r, 1) for r in reads}
    log_dic = {"quantStats": [{}], "annotStats": []}
    align = {}
    cascade.run_annotation_pipeline(seq_dic, libs, log_dic, align_dic=align)
    for i, r in zip(sub, reads):
        got = None if pass_id[i] < 0 else (int(pass_id[i]), int(ref_id[i]), int(pos[i]), int(mm[i]))
        assert align.get(r) == got, r
    # reads too short for any anchor set (< 15 nt) take the pigeonhole pieces inside the same launch
    from mirge_amd import pack
    from mirge_amd.engine import ReadSet
    rng = np.random.default_rng(11)
    seqs = world.libs.libs["mirna"][1]
    tiny = []
    for _ in range(600):
        s = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(12, 24))
        o = int(rng.integers(0, len(s) - ln + 1))
        r = list(s[o:o + ln])
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, ln))] = "ACGT"[int(rng.integers(0, 4))]
        tiny.append("".join(r))
    tiny = list(dict.fromkeys(tiny))
    tw, tl, tn = pack.pack_reads(tiny)
    only8 = [dict(world.passes[8], lib=0)]
    eng8_passes = engine.make_passes([dict(world.passes[8], lib="mirna")])
    engine.set_option("pair_seeds", 1)
    res8 = engine.cascade(ReadSet(tw, tl, tn, None, device=engine.device), eng8_passes)
    ref8 = model.fm_cascade([world.views[LIB_ORDER.index("mirna")]], [dict(only8[0], pair_anchor=res8.stats[0]["pair_anchor"])],
                            tw, tl, tn, wstop=DEFAULT_WSTOP, ftab=True)
    plain8 = model.fm_cascade([world.views[LIB_ORDER.index("mirna")]], only8, tw, tl, tn, wstop=DEFAULT_WSTOP, ftab=True)
    assert res8.stats[0]["pair_anchor"] == 4
    assert_same(res8, ref8)
    for k in ("pass_id", "ref_id", "pos", "mm"):
        assert np.array_equal(plain8[k], ref8[k]), k
    assert int(ref8["stats"][0][2]) > 0 and int(ref8["stats"][0][1]) > 100   # pieces ran (LF steps), reads aligned
    # the rows-compacted kernel on the piece search alone (every strata launch / the last stratum)
    for mode in (2, 1):
        _, res = run_gpu(engine, world, stratum_rows=mode, wstop=DEFAULT_WSTOP, ftab=1)
        assert_same(res, base)
        assert res.stats[8]["lds_mode"] in (5, 6) and res.stats[8]["n_launches"] == 2
    engine.set_option("stratum_rows", 0)
    engine.set_option("fuse", 1)


def test_tally_matches_oracle(engine, world):
    from mirge_amd import synth
    from mirge_amd.engine import split_counts
    for S in (1, 3):
        quant = synth.synth_quant(len(world.lens), n_samples=S)
        rs, res = run_gpu(engine, world, quant=quant)
        counts = engine.tally(rs, res, world.n_mirna).cpu().numpy()
        pass_id, ref_id, _, _ = res.to_host()
        want = model.tally(pass_id, ref_id, quant, world.n_mirna, 9, 0, 8)
        assert np.array_equal(counts.astype(np.uint64), want)
        q, c, cat, uniq = split_counts(counts, world.n_mirna, S, 9)
        assert int(cat.sum()) == int(quant.sum())           # every read lands in one category
        assert np.array_equal(uniq, (quant != 0).sum(axis=0))


def test_tally_with_isomir_pass_disabled(engine, world):
    """A single-pass cascade (BASELINE configs[1]) tallies with isomir_pass = -1: unclaimed
    reads (pass -1) must not be mistaken for it."""
    from mirge_amd import synth
    from mirge_amd.engine import ReadSet
    quant = synth.synth_quant(len(world.lens))
    rs = ReadSet(world.words, world.lens, world.nmask, quant, device=engine.device)
    passes = engine.make_passes([dict(lib="mirna", min_len=0, max_len=25, seed_len=28, max_mm_seed=0,
                                      max_mm_total=2)])
    res = engine.cascade(rs, passes)
    counts = engine.tally(rs, res, world.n_mirna, 0, -1).cpu().numpy()
    pass_id, ref_id, _, _ = res.to_host()
    assert pass_id.min() == -1 and pass_id.max() == 0
    want = model.tally(pass_id, ref_id, quant, world.n_mirna, 1, 0, -1)
    assert np.array_equal(counts.astype(np.uint64), want)


def test_host_buffer_entry_point(engine, world):
    from mirge_amd import synth
    quant = synth.synth_quant(len(world.lens))
    out = engine.annotate_host(world.words, world.lens, world.nmask, engine.mirge_passes(),
                               quant=quant, n_mirna=world.n_mirna)
    ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask)
    for k in ("pass_id", "ref_id", "pos", "mm"):
        assert np.array_equal(out[k], ref[k]), k
    want = model.tally(ref["pass_id"], ref["ref_id"], quant, world.n_mirna, 9, 0, 8)
    assert np.array_equal(out["counts"], want)


def test_empty_and_tiny_inputs(engine, world):
    from mirge_amd import pack
    from mirge_amd.engine import ReadSet
    for reads in ([], ["ACGTACGTACGTACGTAC"], ["TTTTTTTTTTTTTTTTTTTT", "NNNNNNNNNNNNNNNNNN", "ACG"]):
        if reads:
            w, l, nm = pack.pack_reads(reads)
        else:
            w, l, nm = np.zeros((1, 0), np.uint64), np.zeros(0, np.uint8), None
        rs = ReadSet(w, l, nm, np.ones((len(reads), 1), np.uint32), device=engine.device)
        res = engine.cascade(rs, engine.mirge_passes())
        ref = model.fm_cascade(world.views, world.passes, w, l, nm)
        got = res.to_host()
        for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
            assert np.array_equal(a, ref[name]), (reads, name)
        counts = engine.tally(rs, res, world.n_mirna).cpu().numpy()
        assert int(counts[2 * world.n_mirna:2 * world.n_mirna + 10].sum()) == len(reads)


def test_spike_in_cascade_and_many_samples(engine, world, native_lib):
    """Ten-pass cascade (-spikeIn, RAP:574-586) and a sample count whose histogram does not
    fit LDS (the tally then uses global atomics)."""
    from mirge_amd import pack, synth
    from mirge_amd.engine import MIRGE_PASS_TABLE, ReadSet, split_counts
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(77)
    names = ["spike-%d" % i for i in range(12)]
    seqs = ["".join("ACGT"[c] for c in rng.integers(0, 4, 40)) for _ in names]
    spike = FmIndex.build(names, seqs)
    engine.add_library("spike-in", spike)
    reads = list(world.reads[:4000]) + [s[5:27] for s in seqs] + [s[:18] for s in seqs]
    w, l, nm = pack.pack_reads(reads)
    passes = engine.mirge_passes(spike_in=True)
    assert len(passes) == 10
    S = 96
    quant = synth.synth_quant(len(reads), n_samples=S)
    rs = ReadSet(w, l, nm, quant, device=engine.device)
    res = engine.cascade(rs, passes)
    views = world.views + [spike.view()]
    order = LIB_ORDER + ["spike-in"]
    pd = [dict(lib=order.index(k), min_len=a, max_len=b, seed_len=s_, max_mm_seed=ms, max_mm_total=mt,
               trim5=t5, trim3=t3, poly_t=pt) for (k, a, b, s_, ms, mt, t5, t3, pt) in MIRGE_PASS_TABLE]
    ref = model.fm_cascade(views, pd, w, l, nm, wstop=DEFAULT_WSTOP, ftab=True)
    got = res.to_host()
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
        assert np.array_equal(a, ref[name]), name
    assert int((got[0] == 9).sum()) >= 20          # the spike-in pass claimed its reads
    counts = engine.tally(rs, res, world.n_mirna).cpu().numpy()
    assert 2 * world.n_mirna * S * 8 > 160 * 1024   # too big for LDS: global-atomic variant
    want = model.tally(ref["pass_id"], ref["ref_id"], quant, world.n_mirna, 10, 0, 8)
    assert np.array_equal(counts.astype(np.uint64), want)
    q, c, cat, uniq = split_counts(counts, world.n_mirna, S, 10)
    assert int(cat[9].sum()) == int(quant[got[0] == 9].sum())


def test_low_complexity_library_and_reads(native_lib, oracle_lib):
    """Repeats: poly-A tails, tandem repeats and duplicated entries make seed intervals
    hundreds of rows wide; results must still equal the exhaustive scan (lowest entry,
    lowest offset among equally good alignments)."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(5)

    def rnd(n):
        return "".join("ACGT"[c] for c in rng.integers(0, 4, n))
    seqs = [rnd(60) + "A" * int(rng.integers(20, 60)) for _ in range(40)]          # poly-A tails
    seqs += [("ACGT" * 30)[:int(rng.integers(40, 120))] for _ in range(10)]          # tandem repeat
    seqs += [rnd(30) + "CACACACACACACACACACACACACACA" + rnd(10) for _ in range(10)]
    seqs += [seqs[3], seqs[3], "A" * 200, "T" * 90]                                   # duplicates, homopolymers
    seqs += [rnd(12) + "A" * 120 + rnd(5) for _ in range(60)]   # > 6000 rows for an A-run seed
    names = ["rep%d" % i for i in range(len(seqs))]
    ix = FmIndex.build(names, seqs)
    reads = ["A" * L for L in (16, 22, 25, 30, 40)] + ["A" * 21 + "C", "C" + "A" * 21, "ACGT" * 6,
             "CGTA" * 5 + "CG", "CA" * 12, "AC" * 11 + "G", "T" * 22, "T" * 19 + "AAA", "G" * 22,
             seqs[3][40:62], seqs[3][50:75], "A" * 10 + "N" + "A" * 11]
    reads += [s[int(o):int(o) + 22] for s in seqs[:30] for o in rng.integers(0, len(s) - 22, 3)]
    reads = list(dict.fromkeys(reads))
    w, l, nm = pack.pack_reads(reads)
    eng = Engine(0)
    eng.add_library("rep", ix)
    lib = model.Library(names, seqs)
    for (seed_len, mm_seed, mm_total, t5, t3, wide) in ((28, 0, 2, 0, 0, 256), (28, 1, 2, 0, 0, 256),
                                                        (1024, 1, 1, 0, 0, 256), (1024, 2, 2, 1, 2, 256),
                                                        (28, 1, 2, 0, 0, 100000), (1024, 2, 2, 1, 2, 8)):
        eng.set_option("wide_rows", wide)
        passes = eng.make_passes([dict(lib="rep", seed_len=seed_len, max_mm_seed=mm_seed,
                                       max_mm_total=mm_total, trim5=t5, trim3=t3)])
        res = eng.cascade(ReadSet(w, l, nm, None, device=eng.device), passes)
        pass_id, ref_id, pos, mm = res.to_host()
        trimmed = [r[t5:len(r) - t3] if t3 else r[t5:] for r in reads]
        want_ref, want_pos, want_mm = model.align_batch(lib, trimmed, seed_len, mm_seed, mm_total)
        for i, r in enumerate(reads):
            got = (int(ref_id[i]), int(pos[i]), int(mm[i])) if pass_id[i] == 0 else (-1, -1, -1)
            assert got == (int(want_ref[i]), int(want_pos[i]), int(want_mm[i])), (r, seed_len, mm_seed)
        assert max(s["candidates"] for s in res.stats) > 20000   # the wave-wide verification ran


def _repeat_world():
    """A library with planted repeats (one alignment reachable from several pigeonhole pieces,
    reads with several best hits), an N, low-complexity runs, and reads cut from it."""
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    from mirge_amd import pack
    rng = np.random.default_rng(77)
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    unit = rnd(40)
    mut = list(unit)
    mut[7] = "A" if mut[7] != "A" else "C"
    mut2 = list(unit)
    mut2[30] = "G" if mut2[30] != "G" else "T"
    seqs = [rnd(3000) + unit + rnd(500) + "".join(mut) + rnd(700) + unit + rnd(100),
            rnd(1200) + "".join(mut2) + "N" + rnd(800) + "A" * 60 + rnd(300) + "ACGT" * 20 + rnd(50),
            unit[:30]]
    names = ["chrA", "chrB", "chrC"]
    olib = model.Library(names, seqs)
    eng = Engine(0)
    eng.add_library("g", FmIndex.build(names, seqs))
    reads = []
    for s in seqs[:2]:
        for _ in range(300):
            L = int(rng.integers(14, 41))
            o = int(rng.integers(0, len(s) - L))
            r = list(s[o:o + L].replace("N", "A"))
            for _ in range(int(rng.integers(0, 4))):
                k = int(rng.integers(0, L))
                r[k] = "ACGT"[int(rng.integers(0, 4))]
            reads.append("".join(r))
    for L in (16, 20, 22, 28, 30, 33, 40):
        for o in (0, 3, 8):
            if o + L <= 40:
                reads += [unit[o:o + L], "".join(mut)[o:o + L], "".join(mut2)[o:o + L]]
    reads += ["A" * 20, "A" * 35, "ACGT" * 6, "ACGTACGTACGTACGTACGTAC", "ACGTNACGTACGTACGTACG", rnd(25), "AC", "A"]
    words, lens, nmask = pack.pack_reads(reads)
    rs = ReadSet(words, lens, nmask, None, device=eng.device)
    return eng, olib, reads, rs


@pytest.mark.gpu
def test_list_best_matches_exhaustive_scan(native_lib, oracle_lib):
    """mrg_list_best_count/fill (`-a --best --strata`, parseAlignment3 RAP:41-52): the same
    (entry, offset) lists as the exhaustive scan."""
    eng, olib, reads, rs = _repeat_world()
    for seed_len, n_seed, n_total in ((64, 1, 1), (64, 0, 0), (28, 1, 2), (64, 2, 2)):
        for opts in ({}, {"wstop": 0, "ftab": 0}):
            for k, v in {"wstop": DEFAULT_WSTOP, "ftab": 1, **opts}.items():
                eng.set_option(k, v)
            mm, off, ref, pos = eng.list_best(rs, "g", seed_len=seed_len, max_mm_seed=n_seed, max_mm_total=n_total)
            assert off[0] == 0 and off[-1] == len(ref) == len(pos)
            multi = 0
            for i, r in enumerate(reads):
                want, want_mm = model.align_all_best(olib, r, seed_len, n_seed, n_total)
                got = list(zip(ref[off[i]:off[i + 1]].tolist(), pos[off[i]:off[i + 1]].tolist()))
                if not want:
                    assert mm[i] == 255 and got == [], r
                else:
                    assert int(mm[i]) == want_mm and got == want, (r, seed_len, n_seed, n_total, opts)
                    multi += len(got) > 1
            assert multi > 20
    eng.set_option("wstop", DEFAULT_WSTOP)
    eng.set_option("ftab", 1)


@pytest.mark.gpu
def test_count_best_matches_exhaustive_scan(native_lib, oracle_lib):
    """mrg_count_best (the -ai genome filters, W2C:1263/:1488): best mismatch count and its
    multiplicity equal the exhaustive scan, on a library with planted repeats so that one
    alignment is reachable from several pigeonhole pieces and one read has several best hits."""
    eng, olib, reads, rs = _repeat_world()
    for seed_len, n_seed, n_total in ((28, 1, 2), (28, 0, 2), (28, 2, 2), (20, 1, 1), (64, 0, 0)):
        for opts in ({}, {"wstop": 0, "ftab": 0}):
            for k, v in {"wstop": DEFAULT_WSTOP, "ftab": 1, **opts}.items():
                eng.set_option(k, v)
            mm, cnt = eng.count_best(rs, "g", seed_len=seed_len, max_mm_seed=n_seed, max_mm_total=n_total)
            for i, r in enumerate(reads):
                em, ec = model.best_stratum(olib, r, seed_len, n_seed, n_total)
                assert (int(mm[i]), int(cnt[i])) == (em, min(ec, 255)), (r, seed_len, n_seed, n_total, opts)
    eng.set_option("wstop", DEFAULT_WSTOP)
    eng.set_option("ftab", 1)


@pytest.mark.gpu
def test_count_and_list_best_on_reads_of_four_and_eight_words(native_lib, oracle_lib):
    """mrg_count_best / mrg_list_best_* (the -ai genome filters, the -trf listings) on reads of 65..255 nt: the
    W = 4 and W = 8 instantiations of count_kernel = the exhaustive scan (seed in the first 28 bases or the whole
    read; repeats, an N, a low-complexity stretch in the library)."""
    from mirge_amd import pack
    from mirge_amd.engine import ReadSet
    eng, olib, _, _ = _repeat_world()
    rng = np.random.default_rng(78)
    seqs = olib.seqs
    for lo, hi, W in ((65, 128, 4), (129, 255, 8)):
        reads = []
        for s in seqs[:2]:
            for _ in range(120):
                L = int(rng.integers(lo, hi + 1))
                o = int(rng.integers(0, len(s) - L))
                r = list(s[o:o + L].replace("N", "A"))
                for _ in range(int(rng.integers(0, 4))):
                    r[int(rng.integers(0, L))] = "ACGTN"[int(rng.integers(0, 5))]
                reads.append("".join(r))
        reads += ["A" * hi, ("ACGT" * 64)[:hi], seqs[0][:hi], seqs[1][-hi:]]
        words, lens, nmask = pack.pack_reads(reads)
        assert words.shape[0] == W
        rs = ReadSet(words, lens, nmask, None, device=eng.device)
        for seed_len, n_seed, n_total in ((28, 1, 2), (28, 0, 2), (1024, 2, 2), (1024, 0, 0)):
            mm, cnt = eng.count_best(rs, "g", seed_len=seed_len, max_mm_seed=n_seed, max_mm_total=n_total)
            bm, off, ref, pos = eng.list_best(rs, "g", seed_len=seed_len, max_mm_seed=n_seed, max_mm_total=n_total)
            hits = 0
            for i, r in enumerate(reads):
                em, ec = model.best_stratum(olib, r, seed_len, n_seed, n_total)
                assert (int(mm[i]), int(cnt[i])) == (em, min(ec, 255)), (r, seed_len, n_seed, n_total)
                want, want_mm = model.align_all_best(olib, r, seed_len, n_seed, n_total)
                got = list(zip(ref[off[i]:off[i + 1]].tolist(), pos[off[i]:off[i + 1]].tolist()))
                assert got == want and (not want or int(bm[i]) == want_mm), (r, seed_len, n_seed, n_total)
                hits += bool(want)
            assert hits > (60 if n_total else 10)


@pytest.mark.gpu
def test_big_library_jump_tables(native_lib, oracle_lib):
    """A 4.8 Mbp library served from HBM with the k = 12 / 11 / 6 / 4 tables: GPU = CPU port,
    for W = 1 and W = 2 reads."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    from tests.util import BIG_PASSES, big_library_case
    names, seqs, reads = big_library_case()
    ix = FmIndex.build(names, seqs)
    eng = Engine(0)
    eng.add_library("big", ix)
    w, l, nm = pack.pack_reads(reads)
    assert w.shape[0] == 2
    ref = model.fm_cascade([ix.view()], BIG_PASSES, w, l, nm, wstop=DEFAULT_WSTOP, ftab=True)
    rs = ReadSet(w, l, nm, None, device=eng.device)
    # the FM kernels on the whole batch (their search counters equal the port's) ...
    eng.set_option("split_mixed", 0)
    res = eng.cascade(rs, eng.make_passes([dict(p, lib="big") for p in BIG_PASSES]))
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), res.to_host()):
        assert np.array_equal(a, ref[name]), name
    for i, st in enumerate(res.stats):
        assert [st[k] for k in ("processed", "aligned", "steps", "candidates", "lookups")] == \
            [int(x) for x in ref["stats"][i]]
    # ... and the default: the batch split, its 20..32-nt reads through the dictionary kernels
    eng.set_option("split_mixed", 1)
    res = eng.cascade(rs, eng.make_passes([dict(p, lib="big") for p in BIG_PASSES]))
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), res.to_host()):
        assert np.array_equal(a, ref[name]), name
    for i, st in enumerate(res.stats):
        assert [st[k] for k in ("processed", "aligned")] == [int(x) for x in ref["stats"][i][:2]]
    assert any(st["ms_rest"] > 0 for st in res.stats)
    short = [r for r in reads if len(r) <= 32]
    w1, l1, nm1 = pack.pack_reads(short)
    assert w1.shape[0] == 1
    ref1 = model.fm_cascade([ix.view()], BIG_PASSES, w1, l1, nm1, wstop=DEFAULT_WSTOP, ftab=True)
    res1 = eng.cascade(ReadSet(w1, l1, nm1, None, device=eng.device),
                       eng.make_passes([dict(p, lib="big") for p in BIG_PASSES]))
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), res1.to_host()):
        assert np.array_equal(a, ref1[name]), name
    # the same library twice more behind a first pass: the two 1-mismatch passes share one fused
    # launch whose candidates come as 16-byte wide rows (row + 32 bases of context) -- reads of
    # 16..40 nt, so the stored context covers all, or only part, of what the seed left open; the
    # boundary reads (entry ends, N) are among them
    fused_passes = [dict(lib=0, seed_len=28, max_mm_seed=0, max_mm_total=2, min_len=0, max_len=20),
                    dict(lib=0, seed_len=28, max_mm_seed=1, max_mm_total=2, min_len=0, max_len=30),
                    dict(lib=0, seed_len=1024, max_mm_seed=1, max_mm_total=1, min_len=0, max_len=255)]
    # ... and, by default, reads whose seed region is 15..19 bases (two pieces of 7..9 bases: 40..600
    # rows each in 4.8 Mbp) go through the three pairs of three 5-base anchors instead (10-base keys,
    # tables built on the device at first use; fused_kernel<W, true>)
    eng.set_option("dict", 0)   # (the fused FM launches; the dictionary kernels of the one-word batch follow below)
    for pair_big in (5, 0, 6):
        eng.set_option("pair_big", pair_big)
        for ww, ll, nn in ((w, l, nm), (w1, l1, nm1)):
            resf = eng.cascade(ReadSet(ww, ll, nn, None, device=eng.device),
                               eng.make_passes([dict(p, lib="big") for p in fused_passes]))
            assert [st["pair_anchor"] for st in resf.stats] == [0, pair_big, pair_big]
            port_passes = [dict(p, pair_anchor=st["pair_anchor"]) for p, st in zip(fused_passes, resf.stats)]
            reff = model.fm_cascade([ix.view()], port_passes, ww, ll, nn, wstop=DEFAULT_WSTOP, ftab=True)
            assert [st["lds_mode"] for st in resf.stats] == [resf.stats[0]["lds_mode"], 4, 4]
            for name, a in zip(("pass_id", "ref_id", "pos", "mm"), resf.to_host()):
                assert np.array_equal(a, reff[name]), (pair_big, name)
            for i, st in enumerate(resf.stats):
                assert [st[k] for k in ("processed", "aligned", "steps", "candidates", "lookups")] == \
                    [int(x) for x in reff["stats"][i]], (pair_big, i)
            assert int((resf.to_host()[0] == 2).sum()) > 50
            if pair_big:   # the same assignments from far fewer candidate rows
                plain = model.fm_cascade([ix.view()], fused_passes, ww, ll, nn, wstop=DEFAULT_WSTOP, ftab=True)
                for name in ("pass_id", "ref_id", "pos", "mm"):
                    assert np.array_equal(plain[name], reff[name]), name
                assert int(reff["stats"][1][3]) < int(plain["stats"][1][3])
    eng.set_option("pair_big", 5)
    # the one-word batch through seed_kernel (dict = 1, the default): seed buckets for the 22..23-nt
    # reads of this 4.8 Mbp library, the jump tables for the other lengths; boundary reads included
    eng.set_option("dict", 1)
    assert nm1 is None
    resd = eng.cascade(ReadSet(w1, l1, None, None, device=eng.device), eng.make_passes([dict(p, lib="big") for p in fused_passes]))
    plain = model.fm_cascade([ix.view()], fused_passes, w1, l1, None, wstop=DEFAULT_WSTOP, ftab=True)
    assert [st["lds_mode"] for st in resd.stats][1:] == [9, 9]
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), resd.to_host()):
        assert np.array_equal(a, plain[name]), name
    for i, st in enumerate(resd.stats):
        assert (st["processed"], st["aligned"]) == (int(plain["stats"][i][0]), int(plain["stats"][i][1]))


@pytest.mark.gpu
def test_count_and_list_best_on_a_20mbp_library(native_lib, oracle_lib):
    """mrg_count_best / mrg_list_best against a library served from HBM with the k = 13 jump
    table (a genome part in miniature): a sample of reads equals the exhaustive scan."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = acgt[rng.integers(0, 4, 30)].tobytes().decode()
    seqs = []
    for c in range(4):
        body = acgt[rng.integers(0, 4, 5_000_000)].tobytes().decode()
        cut = int(rng.integers(1000, 4_000_000))
        seqs.append(body[:cut] + unit + body[cut:])          # one 30-mer present in every chromosome
    names = ["chr%d" % (c + 1) for c in range(4)]
    ix = FmIndex.build(names, seqs)
    assert list(ix.info.ftab_ks) == [13, 11, 6, 4]
    eng = Engine(0)
    eng.add_library("g", ix)
    reads = [unit[:20], unit[5:27], unit[:10] + "A" + unit[11:24]]
    for _ in range(37):
        s = seqs[int(rng.integers(0, 4))]
        L = int(rng.integers(18, 31))
        o = int(rng.integers(0, len(s) - L))
        r = list(s[o:o + L])
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, L))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(r))
    olib = model.Library(names, seqs)
    w, l, nm = pack.pack_reads(reads)
    rs = ReadSet(w, l, nm, None, device=eng.device)
    for n_seed in (0, 1):
        mm, cnt = eng.count_best(rs, "g", seed_len=28, max_mm_seed=n_seed, max_mm_total=2)
        bm, off, ref, pos = eng.list_best(rs, "g", seed_len=28, max_mm_seed=n_seed, max_mm_total=2)
        for i, r in enumerate(reads):
            em, ec = model.best_stratum(olib, r, 28, n_seed, 2)
            assert (int(mm[i]), int(cnt[i])) == (em, min(ec, 255)), (r, n_seed)
            want, _ = model.align_all_best(olib, r, 28, n_seed, 2)
            got = list(zip(ref[off[i]:off[i + 1]].tolist(), pos[off[i]:off[i + 1]].tolist()))
            assert int(bm[i]) == em and got == want, (r, n_seed)
        assert int(cnt[0]) >= 4 and int(cnt[1]) >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("lo,hi,W,per_entry", [(65, 128, 4, 6), (129, 255, 8, 12)])
def test_long_reads_four_and_eight_words(native_lib, oracle_lib, lo, hi, W, per_entry):
    """Reads of 65..128 nt (four packed words, the W = 4 instantiations) and of 129..255 nt (eight words: what
    `-ad none` leaves of a 151- or 250-cycle run, RAP:543-554 offers them to every pass without a length cap):
    cascade = CPU port, and the port = the exhaustive scan on the same reads (seed rule: mismatches beyond the
    first 28 bases count only towards the total)."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from tests.util import World, pass_dicts
    w = World(with_n=True, n_fixed=200, n_var=50)
    rng = np.random.default_rng(31)
    reads = []
    for key in ("mrna", "ncrna_others", "hairpin", "snorna", "mature_trna"):
        for s in w.libs.libs[key][1]:
            if len(s) < lo + 1:
                continue
            for _ in range(per_entry):
                L = int(rng.integers(lo, min(hi, len(s)) + 1))
                o = int(rng.integers(0, len(s) - L + 1))
                r = list(s[o:o + L])
                for _ in range(int(rng.integers(0, 4))):
                    r[int(rng.integers(0, L))] = "ACGTN"[int(rng.integers(0, 5))]
                reads.append("".join(r))
    reads = list(dict.fromkeys(reads))
    reads = [reads[i] for i in rng.permutation(len(reads))[:3000]]   # (every library with entries that long, not the first one's only)
    words, lens, nmask = pack.pack_reads(reads)
    assert words.shape[0] == W and int(lens.max()) > hi - 8 and int(lens.min()) >= lo and len(reads) > 500
    eng = Engine(0)
    for k in LIB_ORDER:
        eng.add_library(k, w.index[k])
    res = eng.cascade(ReadSet(words, lens, nmask, None, device=eng.device), eng.mirge_passes())
    # default options: passes 1..7 run as fused launches with folded bitmaps (stats say which)
    ref = model.fm_cascade(w.views, [dict(p, kbits_log2=st["kbits_log2"], pair_anchor=st["pair_anchor"]) for p, st in zip(pass_dicts(), res.stats)],
                           words, lens, nmask, wstop=DEFAULT_WSTOP, ftab=True)
    assert any(st["lds_mode"] == 4 for st in res.stats)
    assert_same(res, ref)
    assert sum(int(ref["stats"][i][1]) for i in range(9)) > len(reads) // 3
    # the port against the exhaustive-scan cascade on a subset
    olibs = {k: model.Library(*w.libs.libs[k]) for k in LIB_ORDER}
    sub = reads[:400]
    seq_dic = {r: {"quant": [1], "annot": [0] + [""] * 9, "length": len(r)} for r in sub}
    align = {}
    cascade.run_annotation_pipeline(seq_dic, olibs, {"annotStats": []}, align_dic=align)
    for i, r in enumerate(sub):
        got = None
        if ref["pass_id"][i] >= 0:
            got = (int(ref["pass_id"][i]), int(ref["ref_id"][i]), int(ref["pos"][i]), int(ref["mm"][i]))
        assert align.get(r) == got, r


@pytest.mark.gpu
def test_c_abi_allreduce_single_rank(native_lib):
    """mrg_comm_unique_id / mrg_comm_init / mrg_allreduce (RCCL bound at run time): with one rank the
    sum is the identity -- what can be exercised on a one-GPU box; more ranks need more GPUs."""
    import torch
    from mirge_amd.engine import Engine
    eng = Engine(0)
    t = torch.arange(1000, dtype=torch.int64, device=eng.device) * 3
    want = t.clone()
    eng.allreduce(t)                      # no communicator yet: one process, nothing to add
    assert torch.equal(t, want)
    uid = Engine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(uid, 0, 1)
    eng.allreduce(t)
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    eng.comm_destroy()


_TWO_RANK_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
rank, world, uid_path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
import time
import torch
from mirge_amd.engine import Engine
eng = Engine(rank)                                   # one process per GPU
if rank == 0:
    uid = Engine.comm_unique_id()
    with open(uid_path + ".tmp", "wb") as fh:
        fh.write(uid)
    os.replace(uid_path + ".tmp", uid_path)
else:
    for _ in range(600):
        if os.path.exists(uid_path):
            break
        time.sleep(0.1)
    uid = open(uid_path, "rb").read()
eng.comm_init(uid, rank, world)
t = (torch.arange(5000, dtype=torch.int64, device=eng.device) + 1) * (rank + 1)
eng.allreduce(t)
torch.cuda.synchronize()
want = (torch.arange(5000, dtype=torch.int64) + 1) * sum(r + 1 for r in range(world))
assert torch.equal(t.cpu(), want), "rank %d: all-reduce sum is wrong" % rank
eng.comm_destroy()
print("rank %d ok" % rank)
'''


@pytest.mark.gpu
def test_c_abi_allreduce_two_ranks_over_rccl(native_lib, tmp_path):
    """mrg_comm_init / mrg_allreduce with world = 2: two processes, one GPU each, the unique id through a
    file, the fused count vector's reduce over RCCL.  Runs whenever the box shows two
    devices (the round's one-GPU boxes skip it: the multi-GPU path is then covered by the gloo tests only)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    worker = tmp_path / "worker.py"
    worker.write_text(_TWO_RANK_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(worker), root, str(r), "2", str(tmp_path / "uid")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=300)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)
