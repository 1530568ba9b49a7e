"""The A-to-I position tally (writeDataToCSV.py:145-229).

CPU: oracle/edit_tally.c (dash-padded strings on the cascade's diagonal) against the Python host
path mirge_amd/a2i.py (judge_align / a2i_editing, pinned to the reference's own a2IEditing.* files
by tests/golden/a2i.json) on randomized miRNA groups.
GPU (-m gpu): mrg_edit_tally_run against oracle/edit_tally.c on a seeded world with A->G edits,
several samples, a keep mask and merged-name bins."""
import io

import numpy as np
import pytest

from mirge_amd import a2i, pack, synth
from mirge_amd.index import FmIndex
from oracle import model


def _random_group(rng, n_mir=40):
    """miRNA entries (2 + mature + 6) and reads cut around their mature sequences with shifts,
    3' additions and substitutions; returns (names, seqs, reads, pass, ref, pos)."""
    names, seqs, matures = [], [], []
    for k in range(n_mir):
        mlen = int(rng.integers(18, 26))
        body = "".join("ACGT"[c] for c in rng.integers(0, 4, mlen + 8))
        names.append("m%d" % k)
        seqs.append(body)
        matures.append(body[2:-6])
    reads, pas, ref, pos = [], [], [], []
    for _ in range(3000):
        e = int(rng.integers(0, n_mir))
        ent = seqs[e]
        d = int(rng.integers(-2, 3))              # read start relative to the mature start
        L = int(rng.integers(16, 27))
        o0 = 2 + d
        if o0 < 0 or o0 + L > len(ent) + 3:
            continue
        r = list((ent + "ACGTAC")[o0:o0 + L])     # may run past the entry end (3' addition)
        for _ in range(int(rng.integers(0, 3))):
            j = int(rng.integers(0, L))
            r[j] = "ACGT"[int(rng.integers(0, 4))]
        if rng.random() < 0.4:                    # an A -> G edit inside the scored part
            cand = [j for j in range(L) if 0 <= d + j < len(matures[e]) - 5 and matures[e][d + j] == "A"]
            if cand:
                r[cand[int(rng.integers(0, len(cand)))]] = "G"
        is_iso = rng.random() < 0.5
        reads.append("".join(r))
        pas.append(8 if is_iso else 0)
        ref.append(e)
        pos.append(o0 + (1 if is_iso else 0))     # the isomiR pass reports the base after its -5 1 trim
    return names, seqs, matures, reads, np.array(pas, np.int8), np.array(ref, np.int32), np.array(pos, np.int32)


def test_c_restatement_equals_the_python_host_path(native_lib, oracle_lib):
    rng = np.random.default_rng(3)
    names, seqs, matures, reads, pas, ref, pos = _random_group(rng)
    ix = FmIndex.build(names, seqs)
    words, lens, nmask = pack.pack_reads(reads)
    quant = rng.integers(1, 9, size=(len(reads), 1)).astype(np.uint32)
    got = model.edit_tally(ix, pas, ref, pos, words, lens, quant, nmask=nmask)
    M = len(names)
    tot = got[:M * 3].reshape(M, 1, 3)
    posc = got[M * 3:].reshape(M, 32, 1)
    seen_hits = 0
    for e in range(M):
        idx = [i for i in range(len(reads)) if ref[i] == e]
        if not idx:
            continue
        # a2i_editing pads with its own local alignment (best ungapped diagonal); keep the cases
        # where that IS the cascade's diagonal (always, up to ties on these random sequences)
        sel = []
        for i in idx:
            tpad, spad = a2i.local_pair(matures[e], reads[i])
            d_py = a2i.dash_count(spad)[0] - a2i.dash_count(tpad)[0]
            d_gpu = int(pos[i]) - (1 if pas[i] == 8 else 0) - 2
            if d_py == d_gpu:
                sel.append(i)
        assert len(sel) >= 0.95 * len(idx)
        keep = np.zeros(len(reads), np.uint8)
        keep[sel] = 1
        one = model.edit_tally(ix, pas, ref, pos, words, lens, quant, nmask=nmask, keep=keep)
        rs = [reads[i] for i in sel]
        cs = [int(quant[i, 0]) for i in sel]
        kept, positions, pos_count, ratio, pval, count_true, seq_true, canonical = a2i.a2i_editing(
            matures[e], rs, cs, names[e], io.StringIO(), set(rs))
        # duplicated reads in `rs` are separate records here and there alike
        assert (int(one[e * 3]), int(one[e * 3 + 1]), int(one[e * 3 + 2])) == (count_true, seq_true, canonical), e
        want = np.zeros(32, np.int64)
        for p in positions:
            want[p - 1] = pos_count[p]
        assert np.array_equal(one[M * 3:].reshape(M, 32)[e].astype(np.int64), want), e
        seen_hits += int(want.sum() > 0)
    assert seen_hits >= 10 and int(tot[:, 0, 0].sum()) > 0 and int(posc.sum()) > 0


@pytest.mark.gpu
def test_gpu_edit_tally_matches_oracle(native_lib, oracle_lib):
    import torch
    from mirge_amd.engine import Engine, ReadSet
    libs = synth.SynthLibraries(seed=synth.MOUSE_SEED, scale=0.05, shapes=synth.MOUSE_SHAPES)
    keys = list(synth.LIB_KEYS)
    index = {k: FmIndex.build(*libs.libs[k]) for k in keys}
    n = 300_000
    w = synth.synth_reads_packed(libs, n, seed=4355, mix=synth.A2I_MIX)[None, :]
    lens = np.full(n, 22, np.uint8)
    quant = synth.synth_quant(n, 3, seed=11)
    eng = Engine(0)
    for k in keys:
        eng.add_library(k, index[k])
    rs = ReadSet(w, lens, None, quant, device=eng.device)
    res = eng.cascade(rs, eng.mirge_passes())
    pass_id, ref_id, pos, mm = res.to_host()
    M = index["mirna"].n_ref
    assert int(((pass_id == 0) | (pass_id == 8)).sum()) > n // 3
    got = eng.edit_tally(rs, res, "mirna").cpu().numpy().astype(np.uint64)
    want = model.edit_tally(index["mirna"], pass_id, ref_id, pos, w, lens, quant)
    assert np.array_equal(got, want)
    tot, per_pos = got[:M * 9].reshape(M, 3, 3), got[M * 9:].reshape(M, 32, 3)
    assert int(per_pos.sum()) > 1000 and int(tot[:, :, 2].sum()) > 0      # edits were found, canonical reads too
    # keep mask + merged-name bins + another substitution type
    rng = np.random.default_rng(8)
    keep = (rng.random(n) < 0.7).astype(np.uint8)
    remap = (np.arange(M) // 2).astype(np.uint32)
    nb = int(remap.max()) + 1
    got2 = eng.edit_tally(rs, res, "mirna", keep=torch.from_numpy(keep).to(eng.device),
                          remap=torch.from_numpy(remap.astype(np.int32)).to(eng.device), n_bins=nb,
                          from_base=1, to_base=3).cpu().numpy().astype(np.uint64)
    want2 = model.edit_tally(index["mirna"], pass_id, ref_id, pos, w, lens, quant, keep=keep, remap=remap,
                             n_bins=nb, from_base=1, to_base=3)
    assert np.array_equal(got2, want2)
    # two-word reads (16..40 nt) with N bases go through the same kernel
    wv, lv = synth.synth_reads_varlen(libs, 60_000, seed=5)
    seqs = pack.unpack_reads(wv, lv, None)
    seqs = [s[:7] + "N" + s[8:] if i % 50 == 0 else s for i, s in enumerate(seqs)]
    wv, lv, nmv = pack.pack_reads(seqs)
    qv = synth.synth_quant(len(seqs), 1, seed=2)
    rsv = ReadSet(wv, lv, nmv, qv, device=eng.device)
    resv = eng.cascade(rsv, eng.mirge_passes())
    pv, rv, ov, _ = resv.to_host()
    gotv = eng.edit_tally(rsv, resv, "mirna").cpu().numpy().astype(np.uint64)
    wantv = model.edit_tally(index["mirna"], pv, rv, ov, wv, lv, qv, nmask=nmv)
    assert np.array_equal(gotv, wantv) and int(gotv.sum()) > 0
