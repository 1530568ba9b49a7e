"""GPU parity (-m gpu) of the dictionary kernels (csrc/dict.hip): batches of one-word reads without N.
Assignments and the per-pass processed / aligned counters (bowtie's log lines, RAP:9-18) must equal
the oracle's CPU port and the exhaustive scan; the FM kernels (option dict = 0) are a third,
independently implemented witness on the same device."""
import numpy as np
import pytest

from oracle import cascade, model
from tests.util import LIB_ORDER, World

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(native_lib, oracle_lib):
    # 22-mers of the standard mixture + 16..32-nt reads cut from the libraries (0-2 edits, some with a
    # poly-T tail), no N: one word per read
    w = World(scale=0.05, n_fixed=20000, n_var=6000, with_n=False, max_var_len=32)
    assert w.words.shape[0] == 1 and w.nmask is None
    return w


@pytest.fixture(scope="module")
def engine(world):
    from mirge_amd.engine import Engine
    eng = Engine(0)
    for k in LIB_ORDER:
        eng.add_library(k, world.index[k])
    # this module's subject is the dictionary kernels: every read of the batch goes to them (by default
    # reads under 20 nt take the FM kernels: tests/test_gpu_split.py)
    eng.set_option("split_min_len", 0)
    return eng


def run(engine, world, passes=None, **opts):
    from mirge_amd.engine import ReadSet
    for k, v in opts.items():
        engine.set_option(k, v)
    rs = ReadSet(world.words, world.lens, None, None, device=engine.device)
    res = engine.cascade(rs, engine.mirge_passes() if passes is None else engine.make_passes(passes))
    return res


def same_assignments(res, ref):
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), res.to_host()):
        assert np.array_equal(a, ref[name]), name
    for i, st in enumerate(res.stats):
        assert st["processed"] == int(ref["stats"][i][0]), i
        assert st["aligned"] == int(ref["stats"][i][1]), i


def test_dict_cascade_equals_port_and_fm_kernels(engine, world):
    ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, None)
    for fuse, seed_units in ((1, 1), (1, 0), (0, 0)):
        res = run(engine, world, dict=1, fuse=fuse, seed_units=seed_units)
        same_assignments(res, ref)
        st = res.stats
        assert st[0]["lds_mode"] == 7 and st[0]["steps"] == 0   # the first pass ran exact_dict_kernel
        assert st[0]["aligned"] <= st[0]["lookups"] <= st[0]["processed"]   # at most one home-slot load per offered read
        if seed_units:
            # hairpin .. ncRNA-others (small libraries) share one seed_kernel launch: hairpin alone (its own
            # length window), tRNA + snoRNA + rRNA + ncRNA-others as ONE unit over their concatenation,
            # pre-tRNA through its dictionary; mRNA (large; `-n 0`: a dictionary unit, no buckets) has a launch of its own
            assert [s["lds_mode"] for s in st[1:8]] == [8] * 7
            assert [s["group"] for s in st[1:8]] == [1] * 6 + [7]
            assert all(s["steps"] == 0 for s in st[1:8])
        elif fuse == 0:
            assert st[3]["lds_mode"] == 7 and st[7]["lds_mode"] == 7  # pre-tRNA and the 6.8 Mbp mRNA library have dictionaries
        fm = run(engine, world, dict=0, fuse=fuse)
        for a, b in zip(res.to_host(), fm.to_host()):
            assert np.array_equal(a, b)
    engine.set_option("fuse", 1)
    engine.set_option("dict", 1)
    engine.set_option("seed_units", 1)


def test_exact_stratum_of_the_two_mismatch_pass_as_a_dictionary_unit(engine, world):
    """stratum0_unit = 1: the 0-mismatch stratum of pass 8 (`-v 2 --best` on the miRNA library) rides in
    the preceding seed launch as a dictionary unit, the pass's own launch searches what is left and
    does not count `processed` twice.  Same assignments and counters (measured slower on the headline
    workload, hence off by default)."""
    ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, None)
    res = run(engine, world, stratum0_unit=1)
    same_assignments(res, ref)
    engine.set_option("stratum0_unit", 0)


def test_dict_cascade_equals_exhaustive_scan(engine, world):
    res = run(engine, world, dict=1)
    pass_id, ref_id, pos, mm = res.to_host()
    sub = np.random.default_rng(7).choice(len(world.reads), 4000, replace=False)
    reads = [world.reads[i] for i in sub]
    libs = {k: model.Library(*world.libs.libs[k]) for k in LIB_ORDER}
    seq_dic = {r: cascade.new_seq_record(r, 1) for r in reads}
    log_dic = {"quantStats": [{}], "annotStats": []}
    align = {}
    cascade.run_annotation_pipeline(seq_dic, libs, log_dic, align_dic=align)
    for i, r in zip(sub, reads):
        got = None if pass_id[i] < 0 else (int(pass_id[i]), int(ref_id[i]), int(pos[i]), int(mm[i]))
        assert align.get(r) == got, r


def test_exact_policies_on_every_dictionary_library(engine, world):
    """Single-pass cascades (the pass is first AND last: identity list in, 'unannotated' written for
    the rest) with `-n 0`, `-v 0`, trims and the poly-T rule, against the port: reads longer than the
    seed (29..32 nt: mismatches allowed behind base 28), reads shorter than the 16-base key after
    trimming (FM fallback)."""
    for lib in ("mirna", "hairpin", "pre_trna", "snorna"):
        li = LIB_ORDER.index(lib)
        for pol in (dict(seed_len=28, max_mm_seed=0, max_mm_total=2), dict(seed_len=1024, max_mm_seed=0, max_mm_total=0),
                    dict(seed_len=20, max_mm_seed=0, max_mm_total=3, trim5=1, trim3=2),
                    dict(seed_len=1024, max_mm_seed=0, max_mm_total=0, poly_t=1),
                    dict(seed_len=28, max_mm_seed=0, max_mm_total=2, min_len=18, max_len=25, trim3=4)):
            p = dict(lib=li, min_len=0, max_len=255, trim5=0, trim3=0, poly_t=0)
            p.update(pol)
            ref = model.fm_cascade(world.views, [p], world.words, world.lens, None)
            res = run(engine, world, passes=[dict(p, lib=lib)], dict=1)
            same_assignments(res, ref)
            assert res.stats[0]["lds_mode"] == 7
            assert 0 < res.stats[0]["aligned"] < res.stats[0]["processed"]


def test_seed_buckets_and_jump_tables_on_a_large_library(engine, world):
    """One-mismatch passes on the 6.8 Mbp mRNA library of this world (a seed of 11 bases has 1.6 rows
    on average: it gets seed buckets): 22..23-nt reads find their rows in the bucket of the seed,
    other lengths and overflowing buckets through the jump table; `seed_buckets` = 0 sends everything
    through the jump table.  Same assignments as the port either way, and as the FM kernels."""
    mi, li = LIB_ORDER.index("mirna"), LIB_ORDER.index("mrna")
    for pol in (dict(seed_len=28, max_mm_seed=1, max_mm_total=2), dict(seed_len=1024, max_mm_seed=1, max_mm_total=1),
                dict(seed_len=28, max_mm_seed=1, max_mm_total=2, trim5=1, trim3=2)):
        passes = [dict(lib=mi, min_len=0, max_len=25, seed_len=28, max_mm_seed=0, max_mm_total=2, trim5=0, trim3=0, poly_t=0),
                  dict(dict(lib=li, min_len=0, max_len=255, trim5=0, trim3=0, poly_t=0), **pol)]
        ref = model.fm_cascade(world.views, passes, world.words, world.lens, None)
        names = [dict(p, lib=LIB_ORDER[p["lib"]]) for p in passes]
        for sb in (1, 0):
            res = run(engine, world, passes=names, seed_buckets=sb)
            same_assignments(res, ref)
            assert res.stats[1]["lds_mode"] in (8, 9) and res.stats[1]["aligned"] > 300
    engine.set_option("seed_buckets", 1)


def test_short_dictionary_keys(world):
    """dict_key = 12: nothing takes the fallback; dict_key = 16 is the default tested above."""
    from mirge_amd.engine import Engine
    eng = Engine(0)
    eng.set_option("dict_key", 12)
    for k in LIB_ORDER:
        eng.add_library(k, world.index[k])
    ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, None)
    res = run(eng, world)
    same_assignments(res, ref)
    assert res.stats[0]["lds_mode"] == 7
    eng.close()


def test_packed_assignments_round_trip(engine, world):
    """mrg_pack_assignments: one word per read carries pass, mismatches, entry and offset (the last
    three saturating); unpacked on the host it equals the four arrays."""
    from mirge_amd.engine import unpack_assignments, PACKED_POS_SAT, PACKED_REF_SAT
    res = run(engine, world)
    packed = engine.pack_assignments(res).cpu().numpy()
    pass_id, ref_id, pos, mm = res.to_host()
    u_pass, u_ref, u_pos, u_mm = unpack_assignments(packed)
    assert np.array_equal(u_pass, pass_id)
    assert np.array_equal(u_ref, np.where(ref_id >= 0, np.minimum(ref_id, PACKED_REF_SAT), -1))
    assert np.array_equal(u_pos, np.where(pos >= 0, np.minimum(pos, PACKED_POS_SAT), -1))
    assert np.array_equal(u_mm, np.minimum(mm, 3))
    assert int((pos > PACKED_POS_SAT).sum()) > 0 and int((pass_id < 0).sum()) > 0   # both edges are exercised
    # an odd, unaligned slice takes the scalar path
    import torch
    from mirge_amd.engine import CascadeResult
    n = len(pass_id) - 3
    sub = CascadeResult(res.pass_id[1:1 + n], res.ref_id[1:1 + n], res.pos[1:1 + n], res.mm[1:1 + n], None, engine, 9)
    got = engine.pack_assignments(sub, out=torch.empty(n, dtype=torch.int32, device=engine.device)).cpu().numpy()
    assert np.array_equal(got, packed[1:1 + n])


def test_packed_cascade_and_tallies_equal_the_four_array_form(engine, world, native_lib, oracle_lib):
    """mrg_cascade_run_packed / mrg_tally_run_packed / mrg_edit_tally_run_packed: every kernel writes the
    packed word instead of the four arrays, the tallies read it.  Equal to packing the four-array run, for
    the dictionary kernels (this world), the FM kernels (`dict` = 0) and a batch with N and two-word reads."""
    from mirge_amd import synth
    from mirge_amd.engine import Engine, ReadSet, unpack_assignments, PACKED_POS_SAT, PACKED_REF_SAT

    def check(eng, w):
        quant = synth.synth_quant(len(w.lens), n_samples=1)
        rs = ReadSet(w.words, w.lens, w.nmask, quant, device=eng.device)
        passes = eng.mirge_passes()
        full = eng.cascade(rs, passes)
        c_full = eng.tally(rs, full, w.n_mirna).cpu().numpy()
        e_full = eng.edit_tally(rs, full).cpu().numpy()
        pk = eng.cascade_packed(rs, passes)
        c_pk = eng.tally(rs, pk, w.n_mirna).cpu().numpy()
        e_pk = eng.edit_tally(rs, pk).cpu().numpy()
        assert np.array_equal(pk.packed.cpu().numpy(), eng.pack_assignments(full).cpu().numpy())
        assert np.array_equal(c_pk, c_full) and np.array_equal(e_pk, e_full)
        assert np.array_equal(pk.pass_counts.cpu().numpy(), full.pass_counts.cpu().numpy())
        e_pass, e_ref, e_pos, e_mm = full.to_host()
        u_pass, u_ref, u_pos, u_mm = pk.to_host()
        assert np.array_equal(u_pass, e_pass) and np.array_equal(u_mm, np.minimum(e_mm, 3))
        assert np.array_equal(u_ref, np.where(e_ref >= 0, np.minimum(e_ref, PACKED_REF_SAT), -1))
        assert np.array_equal(u_pos, np.where(e_pos >= 0, np.minimum(e_pos, PACKED_POS_SAT), -1))

    for dict_on in (1, 0):
        engine.set_option("dict", dict_on)
        check(engine, world)
    engine.set_option("dict", 1)
    w2 = World(scale=0.05, n_fixed=6000, n_var=2000)   # reads with N, up to 44 nt: the FM kernels, two words per read
    assert w2.words.shape[0] == 2 and w2.nmask is not None
    eng2 = Engine(0)
    for k in LIB_ORDER:
        eng2.add_library(k, w2.index[k])
    check(eng2, w2)
    eng2.close()


def test_unaligned_outputs_take_the_non_streaming_first_pass(engine, world):
    """The first pass's streaming instantiation writes EVERY output of the batch, and the launches behind it then
    write claims only (round 4).  An output array it cannot address 16 bytes at a time gets the other instantiation,
    which leaves the unclaimed reads to the cascade's last launch: same words either way, no stale value."""
    import torch
    from mirge_amd.engine import ReadSet
    rs = ReadSet(world.words, world.lens, None, None, device=engine.device)
    passes = engine.mirge_passes()
    n = rs.n
    want = engine.cascade_packed(rs, passes).packed.cpu().numpy()
    big = torch.full((n + 8,), -1, dtype=torch.int32, device=engine.device)      # poisoned: a word nobody writes would show
    counts = torch.zeros(2 * len(passes), dtype=torch.int64, device=engine.device)
    got = engine.cascade_packed(rs, passes, out=(big[1:1 + n], counts)).packed.cpu().numpy()
    assert np.array_equal(got, want)
    assert int(big[0].item()) == -1 and int(big[n + 1].item()) == -1


def test_the_streaming_pass_by_chunks_and_by_stretches(engine, world):
    """exact_dict_kernel's streaming instantiation hands the batch out as every gridDim-th chunk of 4096 reads or -- when
    more than 2 % of the (workgroup, trip) slots of that would stay empty -- as one contiguous stretch per workgroup
    (mrg_pass_stats.variant bit 4).  With five workgroups (grid_pct 1 of 512): 20 480 reads are one round of chunks, one
    read more is stretches; both equal the CPU port, as does the whole batch on the full grid."""
    from mirge_amd.engine import ReadSet
    seen = {}
    n_all = world.words.shape[1]
    assert 2 * n_all > 20481
    w2, l2 = np.concatenate([world.words, world.words], axis=1), np.concatenate([world.lens, world.lens])   # (records, not distinct reads)
    for n_take, pct in ((20480, 1), (20481, 1), (n_all, 100)):
        words, lens = np.ascontiguousarray(w2[:, :n_take]), np.ascontiguousarray(l2[:n_take])
        ref = model.fm_cascade(world.views, world.passes, words, lens, None)
        engine.set_option("grid_pct", pct)
        res = engine.cascade(ReadSet(words, lens, None, None, device=engine.device), engine.mirge_passes())
        same_assignments(res, ref)
        assert res.stats[0]["lds_mode"] == 7
        seen[(n_take, pct)] = res.stats[0]["variant"] & 16
    engine.set_option("grid_pct", 100)
    assert seen[(20480, 1)] == 0 and seen[(20481, 1)] == 16 and seen[(n_all, 100)] == 16, seen
