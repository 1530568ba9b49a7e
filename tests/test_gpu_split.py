"""GPU parity (-m gpu) of split batches: a batch with reads longer than 32 nt or reads with N (two words
per read and / or an N mask -- what a real trimmed FASTQ gives) runs as two cascades over disjoint
lists, its one-word N-free reads through the dictionary kernels and the rest through the FM kernels
(capi.hip: cascade_run_impl).  Assignments, the per-pass processed / aligned counters, the packed form
and the tally must equal the CPU port's and the unsplit FM run's."""
import numpy as np
import pytest

from oracle import model
from mirge_amd.engine import DEFAULT_WSTOP
from tests.util import LIB_ORDER, World

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(native_lib, oracle_lib):
    w = World(scale=0.05, n_fixed=12000, n_var=5000, with_n=True, max_var_len=60)
    assert w.words.shape[0] == 2 and w.nmask is not None
    assert int((w.lens <= 32).sum()) > 5000 and int((w.lens > 32).sum()) > 300 and int((w.lens < 20).sum()) > 100
    return w


@pytest.fixture(scope="module")
def engine(world):
    from mirge_amd.engine import Engine
    eng = Engine(0)
    for k in LIB_ORDER:
        eng.add_library(k, world.index[k])
    yield eng
    eng.close()


def same(res, ref):
    for name, a in zip(("pass_id", "ref_id", "pos", "mm"), res.to_host()):
        assert np.array_equal(a, ref[name]), name
    for i, st in enumerate(res.stats):
        assert (st["processed"], st["aligned"]) == (int(ref["stats"][i][0]), int(ref["stats"][i][1])), i


def test_split_batch_equals_port_and_unsplit_run(engine, world):
    from mirge_amd import synth
    from mirge_amd.engine import ReadSet
    ref = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask, wstop=DEFAULT_WSTOP, ftab=True)
    quant = synth.synth_quant(world.words.shape[1], n_samples=3)
    rs = ReadSet(world.words, world.lens, world.nmask, quant, device=engine.device)
    passes = engine.mirge_passes()
    res = engine.cascade(rs, passes)
    same(res, ref)
    st = res.stats
    # the one-word reads' cascade ran last: its kernels are the ones reported, both cascades launched pass 0
    assert st[0]["lds_mode"] == 7 and st[0]["n_launches"] == 2
    assert st[2]["lds_mode"] in (8, 9)
    assert st[0]["ms_rest"] > 0 and st[0]["ms"] > 0     # each cascade has its own per-pass times
    engine.set_option("long_lane", 1)    # (round 6: the reads of 33..63 nt in the one-word lane too)
    lane = engine.cascade(rs, passes)
    engine.set_option("long_lane", 0)
    same(lane, ref)
    assert lane.stats[2]["variant"] & 8 and lane.stats[0]["lds_mode"] == 7
    engine.set_option("split_mixed", 0)
    whole = engine.cascade(rs, passes)
    engine.set_option("split_mixed", 1)
    same(whole, ref)
    assert whole.stats[0]["lds_mode"] != 7 and whole.stats[0]["n_launches"] == 1 and whole.stats[0]["ms_rest"] == 0
    # packed outputs and the tally on top of them
    pk = engine.cascade_packed(rs, passes)
    assert np.array_equal(pk.packed.cpu().numpy(), engine.pack_assignments(res).cpu().numpy())   # (res equals the port's arrays)
    counts = engine.tally(rs, pk, world.n_mirna).cpu().numpy()
    assert np.array_equal(counts.astype(np.uint64), model.tally(ref["pass_id"], ref["ref_id"], quant, world.n_mirna, 9, 0, 8))


@pytest.mark.parametrize("keep", ["short", "long", "n_only"])
def test_split_batch_with_an_empty_side(engine, world, keep):
    """Two-word batches whose reads are all short (an N mask of zeros / a W = 2 array the caller padded),
    all long, or short but every one with an N: one of the two lists is empty."""
    from mirge_amd.engine import ReadSet
    if keep == "short":
        sel = np.flatnonzero((world.lens <= 32) & (world.lens >= 20) & (world.nmask[0] == 0))[:4000]
    elif keep == "long":
        sel = np.flatnonzero(world.lens > 32)
    else:
        sel = np.flatnonzero((world.lens <= 32) & (world.nmask[0] != 0))
    assert len(sel) > 50
    words = np.ascontiguousarray(world.words[:, sel])
    lens = np.ascontiguousarray(world.lens[sel])
    nmask = np.ascontiguousarray(world.nmask[:, sel])
    ref = model.fm_cascade(world.views, world.passes, words, lens, nmask, wstop=DEFAULT_WSTOP, ftab=True)
    res = engine.cascade(ReadSet(words, lens, nmask, None, device=engine.device), engine.mirge_passes())
    same(res, ref)
    # one word per read but an N mask: still a split batch
    if keep == "n_only":
        res1 = engine.cascade(ReadSet(words[:1].copy(), lens, nmask[:1].copy(), None, device=engine.device), engine.mirge_passes())
        same(res1, ref)


def test_which_kernels_serve_which_length_class(engine, world):
    """Round 4: reads of 16..32 nt without N -- the reference's own length floor is 16 (trim_file.py:33) -- run the
    cascade through the dictionary kernels (exact_dict_kernel, seed_kernel / wave_seed_kernel with the pair tables
    of three anchors for the 16..19-nt reads' one-mismatch passes on a large library, pair_wave_kernel); round 6: with
    `long_lane` = 1 so do the N-free reads of 33..63 nt (the LONG instantiations: built, parity-green, measured no faster than
    the FM kernels, hence off by default); reads with N, reads under 16 nt and beyond 63 take the FM kernels.  Told from the per-pass statistics of
    single-class batches: a batch of one class launches one cascade, and `lds_mode` / `variant` name its kernels."""
    from mirge_amd.engine import ReadSet
    clean = world.nmask[0] == 0
    passes = engine.mirge_passes()
    classes = {
        "16..19": (world.lens >= 16) & (world.lens <= 19) & clean,
        "20..32": (world.lens >= 20) & (world.lens <= 32) & clean,
        "33..60": world.lens > 32,
        "with N": (world.lens <= 32) & ~clean,
    }
    for name, mask in classes.items():
        sel = np.flatnonzero(mask)
        assert len(sel) > 50, name
        words = np.ascontiguousarray(world.words[:, sel])
        lens = np.ascontiguousarray(world.lens[sel])
        nmask = np.ascontiguousarray(world.nmask[:, sel])
        ref = model.fm_cascade(world.views, world.passes, words, lens, nmask, wstop=DEFAULT_WSTOP, ftab=True)
        if name in ("16..19", "20..32"):   # what a trimmed small-RNA batch is: one word per read, no N mask
            words, nmask = np.ascontiguousarray(words[:1]), None
        engine.set_option("long_lane", 1 if name == "33..60" else 0)
        res = engine.cascade(ReadSet(words, lens, nmask, None, device=engine.device), passes)
        engine.set_option("long_lane", 0)
        same(res, ref)
        st = res.stats
        launched = [s for s in st if s["n_launches"]]
        if name in ("16..19", "20..32"):
            # (the hairpin pass, len > 25, is skipped by the hint for the short class)
            assert st[0]["lds_mode"] == 7, name                                  # exact_dict_kernel
            assert st[2]["lds_mode"] in (8, 9) and st[7]["lds_mode"] in (8, 9), name   # seed launches
            assert st[2]["variant"] & 3 == 0, name                               # the small libraries: seed_kernel (tiles)
            assert st[7]["variant"] & 3 == 2, name                               # the large one (mRNA, 6.8 Mbp here): wave_seed_kernel, 96 registers
            assert st[8]["lds_mode"] == 11, name                                 # pair_wave_kernel
            assert all(s["ms_rest"] == 0 and s["n_launches"] <= 1 for s in st), name   # one cascade: nothing went to the FM kernels
            if name == "16..19":
                assert st[8]["pair_anchor"] == 4
                # (the pair tables of three anchors for these reads' one-mismatch passes on a LARGE library are exercised
                # by tests/test_gpu_dict.py::test_seed_buckets_and_jump_tables_on_a_large_library)
        elif name == "33..60":
            # round 6, option long_lane = 1: the N-free reads of 33..63 nt ride the dictionary kernels' LONG instantiations (variant + 8): the
            # hairpin pass (len > 25) and the other one-mismatch passes in the seed launches, the 2-mismatch pass in
            # pair_wave_kernel (it sees at most 32 bases of a 33..35-nt read behind `-5 1 -3 2`; a 36-nt read could still
            # align to a 33-nt miRNA entry with all of its 33 bases: that length stays with the FM kernels, as do reads with N)
            assert st[1]["lds_mode"] in (8, 9) and st[1]["variant"] & 8, (name, st[1])
            assert st[6]["lds_mode"] in (8, 9) and st[6]["variant"] & 8 and st[7]["variant"] & 3 == 2, name
            assert st[8]["lds_mode"] == 11 and st[8]["variant"] & 8, name
            assert st[1]["aligned"] > 50 and st[1]["ms_rest"] > 0, name          # two cascades: some reads (N, 36 nt) stayed with the FM kernels
            old = engine.cascade(ReadSet(words, lens, nmask, None, device=engine.device), passes)   # the default: long_lane = 0
            same(old, ref)
            assert all(s["lds_mode"] not in (7, 8, 9, 11) for s in old.stats if s["n_launches"]), name   # match / fused / stratum kernels only
        else:
            # reads with N of at most 32 nt: the mask says "some read has an N", not which -- the batch is split on the
            # device, every read lands on the FM side (its cascade does the work), the one-word list stays empty
            assert st[0]["n_launches"] == 2 and st[0]["ms_rest"] > 0, name
