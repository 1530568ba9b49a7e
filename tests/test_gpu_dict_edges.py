"""GPU parity (-m gpu), edge cases of the dictionary kernels (csrc/dict.hip): the inputs the reference's
own runs meet at their margins -- repeats and homopolymers (seed intervals thousands of rows wide),
reads shorter than every key and seed table, the ten-pass spike-in cascade, tiny and empty batches,
many samples -- against the exhaustive scan and the CPU port."""
import numpy as np
import pytest

from oracle import model
from mirge_amd.engine import DEFAULT_WSTOP
from tests.util import LIB_ORDER, World

pytestmark = pytest.mark.gpu


def rnd(rng, n):
    return "".join("ACGT"[c] for c in rng.integers(0, 4, n))


def test_low_complexity_library_through_the_dictionary_kernels(native_lib, oracle_lib):
    """Poly-A tails, tandem repeats, duplicated entries, homopolymers: a seed of such a read names
    hundreds to thousands of rows (the workgroup-wide verification and the row queue's overflow paths
    run), dictionary chains overflow into the FM fallback; every policy must still equal the
    exhaustive scan (lowest entry, lowest offset among equally good alignments).  One-word reads, no N."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(5)
    seqs = [rnd(rng, 60) + "A" * int(rng.integers(20, 60)) for _ in range(40)]
    seqs += [("ACGT" * 30)[:int(rng.integers(40, 120))] for _ in range(10)]
    seqs += [rnd(rng, 30) + "CACACACACACACACACACACACACACA" + rnd(rng, 10) for _ in range(10)]
    seqs += [seqs[3], seqs[3], "A" * 200, "T" * 90]
    seqs += [rnd(rng, 12) + "A" * 120 + rnd(rng, 5) for _ in range(60)]
    names = ["rep%d" % i for i in range(len(seqs))]
    ix = FmIndex.build(names, seqs)
    assert ix.exact_dict(16)["n_overflow"] > 0     # the A-run keys overflow their chains
    reads = ["A" * L for L in (16, 22, 25, 30, 32)] + ["A" * 21 + "C", "C" + "A" * 21, "ACGT" * 6, "CGTA" * 5 + "CG", "CA" * 12,
                                                        "AC" * 11 + "G", "T" * 22, "T" * 19 + "AAA", "G" * 22, seqs[3][40:62],
                                                        seqs[3][50:75], "A" * 12, "ACGTACGTAC", "C" * 9]
    reads += [s[int(o):int(o) + L] for s in seqs[:30] for o, L in zip(rng.integers(0, len(s) - 32, 3), (18, 22, 31))]
    reads = list(dict.fromkeys(reads))
    w, l, nm = pack.pack_reads(reads)
    assert w.shape[0] == 1 and nm is None
    eng = Engine(0)
    eng.set_option("split_min_len", 0)     # the short reads too go to the dictionary kernels
    eng.add_library("rep", ix)
    eng.add_library("decoy", FmIndex.build(["decoy"], ["GATTACAGATTACAGGCCTTAAGGCCTTAACGCGCGTATATA" * 3]))
    lib = model.Library(names, seqs)
    first = dict(lib="decoy", seed_len=28, max_mm_seed=0, max_mm_total=2)   # launched first, claims nothing: the
    for (seed_len, mm_seed, mm_total, t5, t3) in ((28, 0, 2, 0, 0), (1024, 0, 0, 0, 0), (28, 1, 2, 0, 0), (1024, 1, 1, 0, 0),
                                                  (20, 1, 3, 1, 2), (28, 0, 2, 2, 1)):                    # pass under test runs second
        pol = dict(lib="rep", seed_len=seed_len, max_mm_seed=mm_seed, max_mm_total=mm_total, trim5=t5, trim3=t3)
        for plan in ([pol], [first, pol]):     # alone (exact_dict_kernel / match_kernel first) and as a seed_kernel unit
            res = eng.cascade(ReadSet(w, l, None, None, device=eng.device), eng.make_passes(plan))
            pass_id, ref_id, pos, mm = res.to_host()
            k = len(plan) - 1
            trimmed = [r[t5:len(r) - t3] if t3 else r[t5:] for r in reads]
            want_ref, want_pos, want_mm = model.align_batch(lib, trimmed, seed_len, mm_seed, mm_total)
            for i, r in enumerate(reads):
                got = (int(ref_id[i]), int(pos[i]), int(mm[i])) if pass_id[i] == k else (-1, -1, -1)
                assert got == (int(want_ref[i]), int(want_pos[i]), int(want_mm[i])), (r, seed_len, mm_seed, len(plan))
            if len(plan) == 2:
                assert res.stats[0]["aligned"] == 0 and res.stats[1]["lds_mode"] in (8, 9)
                assert mm_seed == 0 or res.stats[1]["candidates"] > 5000
    eng.close()


def test_reads_shorter_than_keys_and_seed_tables(native_lib, oracle_lib):
    """8..15-nt reads (a direct C-ABI caller may send them; the poly-T pass strips reads down to 11):
    below the dictionary's 16-base key and with seeds of 4..7 bases, i.e. the FM fallback and the
    shortest jump tables; plus full-length reads for contrast.  GPU = CPU port."""
    from mirge_amd import pack
    from mirge_amd.engine import Engine, ReadSet
    w0 = World(scale=0.03, n_fixed=1500, n_var=0, with_n=False)
    rng = np.random.default_rng(8)
    reads = list(w0.reads)
    for key in ("mirna", "pre_trna", "snorna", "mature_trna"):
        for s in w0.libs.libs[key][1][:150]:
            L = int(rng.integers(8, 16))
            if len(s) > L + 2:
                o = int(rng.integers(0, len(s) - L))
                r = s[o:o + L]
                if rng.random() < 0.3:
                    i = int(rng.integers(0, L))
                    r = r[:i] + "ACGT"[int(rng.integers(0, 4))] + r[i + 1:]
                reads.append(r + ("TTTT" if rng.random() < 0.3 else ""))
    reads = list(dict.fromkeys(reads))
    w, l, nm = pack.pack_reads(reads)
    assert w.shape[0] == 1 and nm is None and int(l.min()) <= 9
    eng = Engine(0)
    for k in LIB_ORDER:
        eng.add_library(k, w0.index[k])
    ref = model.fm_cascade(w0.views, w0.passes, w, l, None, wstop=DEFAULT_WSTOP, ftab=True)
    for split_min_len in (0, 20):          # all reads through the dictionary kernels / the short ones through the FM kernels
        eng.set_option("split_min_len", split_min_len)
        res = eng.cascade(ReadSet(w, l, None, None, device=eng.device), eng.mirge_passes())
        for name, a in zip(("pass_id", "ref_id", "pos", "mm"), res.to_host()):
            assert np.array_equal(a, ref[name]), (split_min_len, name)
        for i, st in enumerate(res.stats):
            assert (st["processed"], st["aligned"]) == (int(ref["stats"][i][0]), int(ref["stats"][i][1])), (split_min_len, i)
        assert res.stats[0]["lds_mode"] == 7 and res.stats[2]["lds_mode"] in (8, 9)
        assert res.stats[0]["n_launches"] == (2 if split_min_len else 1)
    eng.close()


def test_spike_in_tiny_batches_and_many_samples(native_lib, oracle_lib):
    """The ten-pass cascade (-spikeIn, RAP:574-586: a dictionary pass AFTER the 2-mismatch pass ends the
    cascade), batches of 0, 1, 3, 5 and 4 097 reads (partial quartets, partial tiles), and a packed-output
    tally over 96 samples (the histogram does not fit LDS)."""
    from mirge_amd import pack, synth
    from mirge_amd.engine import Engine, MIRGE_PASS_TABLE, ReadSet
    from mirge_amd.index import FmIndex
    w0 = World(scale=0.03, n_fixed=5000, n_var=500, with_n=False, max_var_len=32)
    rng = np.random.default_rng(77)
    names = ["spike-%d" % i for i in range(12)]
    seqs = [rnd(rng, 40) for _ in names]
    spike = FmIndex.build(names, seqs)
    eng = Engine(0)
    for k in LIB_ORDER:
        eng.add_library(k, w0.index[k])
    eng.add_library("spike-in", spike)
    reads = list(w0.reads) + [s[5:27] for s in seqs] + [s[:18] for s in seqs]
    views = w0.views + [spike.view()]
    order = LIB_ORDER + ["spike-in"]
    pd = [dict(lib=order.index(k), min_len=a, max_len=b, seed_len=s_, max_mm_seed=ms, max_mm_total=mt, trim5=t5, trim3=t3, poly_t=pt)
          for (k, a, b, s_, ms, mt, t5, t3, pt) in MIRGE_PASS_TABLE]
    passes = eng.mirge_passes(spike_in=True)
    for n in (len(reads), 4097, 5, 3, 1, 0):
        sub = reads[:n]
        if n:
            w, l, nm = pack.pack_reads(sub)
        else:
            w, l, nm = np.zeros((1, 0), dtype=np.uint64), np.zeros(0, dtype=np.uint8), None
        assert nm is None and w.shape[0] == 1
        res = eng.cascade(ReadSet(w, l, None, None, device=eng.device), passes)
        got = res.to_host()
        if n == 0:
            assert all(len(a) == 0 for a in got) and all(s["processed"] == 0 for s in res.stats)
            continue
        ref = model.fm_cascade(views, pd, w, l, None, wstop=DEFAULT_WSTOP, ftab=True)
        for name, a in zip(("pass_id", "ref_id", "pos", "mm"), got):
            assert np.array_equal(a, ref[name]), (n, name)
        for i, st in enumerate(res.stats):
            assert (st["processed"], st["aligned"]) == (int(ref["stats"][i][0]), int(ref["stats"][i][1])), (n, i)
        if n == len(reads):
            assert int((got[0] == 9).sum()) >= 20 and res.stats[9]["lds_mode"] in (7, 8, 9)
            S = 96
            quant = synth.synth_quant(n, n_samples=S)
            rs = ReadSet(w, l, None, quant, device=eng.device)
            pk = eng.cascade_packed(rs, passes)
            counts = eng.tally(rs, pk, w0.n_mirna).cpu().numpy()
            want = model.tally(ref["pass_id"], ref["ref_id"], quant, w0.n_mirna, 10, 0, 8)
            assert np.array_equal(counts.astype(np.uint64), want)
    # sixteen passes do not fit the packed word's four pass bits
    from mirge_amd._native import MirgeAmdError
    w, l, _ = pack.pack_reads(reads[:10])
    many = eng.make_passes([dict(lib="mirna", seed_len=28, max_mm_seed=0, max_mm_total=2)] * 16)
    with pytest.raises(MirgeAmdError):
        eng.cascade_packed(ReadSet(w, l, None, None, device=eng.device), many)
    eng.close()
