"""GPU parity (-m gpu) on FULL-SIZE libraries (round-5 verdict: the 2^29-slot mRNA dictionary, the k = 14 jump table,
the device-built 8.6 GB table were only ever checked inside bench.py's gate): the 137 Mbp mRNA set and the 11 Mbp
ncRNA-others set of SURVEY.md 8d, indexed here, their derived tables filled on the device, compared word for word with
the host functions (`mrg_ctx_library_check_tables`), and 4 000 reads through `-n 1` (ncRNA-others: seed buckets,
position lists) then `-n 0` (mRNA: the dictionary) against the exhaustive scan of the library texts
(`oracle/bowtie_model.c`: no index at all).  ~90 s, most of it the suffix array of 137 Mbp on one core."""
import numpy as np
import pytest

from oracle import model

pytestmark = pytest.mark.gpu


def test_full_size_mrna_and_ncrna_libraries(native_lib, oracle_lib):
    from mirge_amd import pack, synth
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    libs = synth.SynthLibraries(seed=20181, scale=1.0)
    assert libs.total_bases("mrna") > 130_000_000 and libs.total_bases("ncrna_others") > 10_000_000
    rng = np.random.default_rng(86)
    reads = []
    for key, n in (("mrna", 1800), ("ncrna_others", 1400)):
        names, seqs = libs.libs[key]
        for _ in range(n):
            s = seqs[int(rng.integers(0, len(seqs)))]
            L = int(rng.integers(16, 33)) if rng.random() < 0.4 else 22
            o = int(rng.integers(0, len(s) - L + 1))
            r = list(s[o:o + L]) if rng.random() < 0.9 else list(s[:L] if rng.random() < 0.5 else s[-L:])
            for p in rng.integers(0, L, int(rng.integers(0, 3)) if rng.random() < 0.5 else 0):
                r[p] = "ACGT"[("ACGT".index(r[p]) + 1 + int(rng.integers(0, 3))) % 4]
            reads.append("".join(r))
    reads += ["".join("ACGT"[c] for c in rng.integers(0, 4, 22)) for _ in range(800)]
    reads = list(dict.fromkeys(reads))
    w, l, nm = pack.pack_reads(reads)
    assert w.shape[0] == 1 and nm is None
    index = {k: FmIndex.build(*libs.libs[k]) for k in ("ncrna_others", "mrna")}
    eng = Engine(0)
    eng.add_library("ncrna_others", index["ncrna_others"])
    eng.add_library("mrna", index["mrna"], exact_dict=True)
    assert eng.library_dict_stats("mrna")[0] > 100_000_000          # the 2^29-slot dictionary, filled on the device
    for k in ("ncrna_others", "mrna"):
        chk = eng.check_tables(k)
        assert chk["jump_tables"] == 0 and chk["row_context"] == 0 and chk["wide_rows"] == 0 and chk["seed_buckets"] in (0, None), (k, chk)
    assert eng.check_tables("ncrna_others")["seed_buckets"] == 0
    plan = [dict(lib="ncrna_others", seed_len=28, max_mm_seed=1, max_mm_total=2),
            dict(lib="mrna", seed_len=28, max_mm_seed=0, max_mm_total=2)]
    res = eng.cascade(ReadSet(w, l, None, None, device=eng.device), eng.make_passes(plan))
    pass_id, ref_id, pos, mm = res.to_host()
    assert res.stats[1]["lds_mode"] in (7, 8, 9), res.stats[1]      # a dictionary kernel served the mRNA pass
    eng.close()
    want = {}
    left = list(range(len(reads)))
    for pi, (key, ms) in enumerate((("ncrna_others", 1), ("mrna", 0))):
        lib = model.Library(*libs.libs[key])
        r_, p_, m_ = model.align_batch(lib, [reads[i] for i in left], 28, ms, 2)
        nxt = []
        for j, i in enumerate(left):
            if int(r_[j]) >= 0:
                want[i] = (pi, int(r_[j]), int(p_[j]), int(m_[j]))
            else:
                nxt.append(i)
        left = nxt
    for i in range(len(reads)):
        got = (int(pass_id[i]), int(ref_id[i]), int(pos[i]), int(mm[i])) if pass_id[i] >= 0 else None
        assert got == want.get(i), (reads[i], got, want.get(i))
    assert sum(1 for v in want.values() if v[0] == 0) > 800 and sum(1 for v in want.values() if v[0] == 1) > 800


def test_properties_at_the_headline_size(native_lib):
    """BASELINE configs[2] / configs[3] at their full size -- 100 M reads x 22 nt (bench.py's own read set), the nine-pass
    cascade over the eight full-size libraries -- through properties that need no oracle: the same input gives the same
    answer; a read's outcome does not depend on its neighbours (a permutation); every pass is offered what the passes in
    front left; the fused count vector of the whole job = the sum of the vectors of the 8 contiguous shards configs[3]
    gives its ranks (the all-reduce's arithmetic, on one GPU) = the vector of the collapsed set (18 M distinct reads
    with their counts: what the reference itself annotates, quantReads.py:3-24); packed outputs = the four arrays.
    (Equality with the CPU port on all 100 M reads is bench.py's gate; ~60 s here, most of it generating the reads.)"""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from mirge_amd import dist, synth
    from mirge_amd.engine import Engine, ReadSet, unpack_assignments, split_counts
    from mirge_amd.index import FmIndex
    N = 100_000_000
    libs = synth.SynthLibraries(seed=20181, scale=1.0)
    keys = list(synth.LIB_KEYS)
    with ThreadPoolExecutor(max_workers=len(keys)) as pool:
        fut = {k: pool.submit(FmIndex.build, *libs.libs[k]) for k in keys}
        words, lens, quant = synth.global_read_slice(libs, N, 0, N, workload="cascade", seed0=355, mix=None, n_samples=1)
        index = {k: f.result() for k, f in fut.items()}
    eng = Engine(0)
    for k in keys:
        eng.add_library(k, index[k])
    passes = eng.mirge_passes()
    n_pass, M = len(passes), index["mirna"].n_ref
    rs = ReadSet(words, lens, None, quant, device=eng.device)
    res = eng.cascade(rs, passes)
    whole = eng.tally(rs, res, M).cpu().numpy()
    st = res.stats
    a = [t.clone() for t in (res.pass_id, res.ref_id, res.pos, res.mm)]
    # conservation: all reads are 22 nt, so every pass whose window holds 22 nt is offered exactly what is left
    assert st[0]["processed"] == N
    left = N
    for i, s in enumerate(st):
        if i not in (1, 3):       # (the hairpin pass takes reads of more than 25 nt, the pre-tRNA pass the poly-T rule's)
            assert s["processed"] == left, (i, st)
        left -= s["aligned"]
    assert sum(s["aligned"] for s in st) == int((a[0] >= 0).sum().item()) and 0 < left < N // 4
    q, iscan, cat, uniq = split_counts(whole, M, 1, n_pass)
    assert int(cat.sum()) == int(quant.astype(np.int64).sum()) and int(uniq[0]) == N
    # idempotence, and the packed form of the same run
    res_p = eng.cascade_packed(rs, passes)
    whole_p = eng.tally(rs, res_p, M).cpu().numpy()
    assert np.array_equal(whole, whole_p)
    p_pass, p_ref, p_pos, p_mm = unpack_assignments(res_p.packed.cpu().numpy())
    assert np.array_equal(p_pass, a[0].cpu().numpy())
    from mirge_amd.engine import PACKED_POS_SAT, PACKED_REF_SAT
    ref_h, pos_h = a[1].cpu().numpy(), a[2].cpu().numpy()
    assert np.array_equal(p_ref, np.where(ref_h >= 0, np.minimum(ref_h, PACKED_REF_SAT), -1))
    assert np.array_equal(p_pos, np.where(pos_h >= 0, np.minimum(pos_h, PACKED_POS_SAT), -1))
    del p_pass, p_ref, p_pos, p_mm, ref_h, pos_h
    # the 8 shards of configs[3] (dist.shard_bounds: contiguous), summed as the all-reduce would
    total = np.zeros_like(whole)
    for r in range(8):
        lo, hi = dist.shard_bounds(N, r, 8)
        rs_r = ReadSet.from_device(rs.words[:, lo:hi].contiguous(), rs.lens[lo:hi], None, rs.quant[lo:hi], 22, 22)
        res_r = eng.cascade(rs_r, passes)
        total += eng.tally(rs_r, res_r, M).cpu().numpy()
        assert torch.equal(res_r.pass_id, a[0][lo:hi]) and torch.equal(res_r.ref_id, a[1][lo:hi]) and torch.equal(res_r.pos, a[2][lo:hi])
    # (the distinct-read counter adds up over shards of RECORDS too: every record is one read here)
    assert np.array_equal(total, whole)
    # a permutation of the reads
    perm = torch.randperm(N, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(7))
    rs_q = ReadSet.from_device(rs.words[:, perm].contiguous(), rs.lens[perm], None, rs.quant[perm], 22, 22)
    res_q = eng.cascade(rs_q, passes)
    assert torch.equal(res_q.pass_id, a[0][perm]) and torch.equal(res_q.ref_id, a[1][perm]) and torch.equal(res_q.pos, a[2][perm]) and \
        torch.equal(res_q.mm, a[3][perm])
    assert np.array_equal(eng.tally(rs_q, res_q, M).cpu().numpy(), whole)
    del rs_q, res_q, perm
    # the collapsed set (the records taken as RAW reads, every record counting once): distinct reads with their
    # multiplicities give the same miRNA / isomiR / category counts; only trimmedUniq counts uniques instead of records
    from mirge_amd.engine import CascadeResult
    ones = torch.ones((N, 1), dtype=torch.int32, device=eng.device)
    rec = eng.tally(ReadSet.from_device(rs.words, rs.lens, None, ones, 22, 22), CascadeResult(a[0], a[1], a[2], a[3], None, eng, n_pass), M).cpu().numpy()
    rs_u, hist = eng.collapse(rs.words, rs.lens, None, None, 1, 22)
    rs_u.min_len = 22
    assert 10_000_000 < rs_u.n < 30_000_000 and int(rs_u.quant.sum().item()) == N and int(hist[22, 0].item()) == N
    res_u = eng.cascade(rs_u, passes)
    coll = eng.tally(rs_u, res_u, M).cpu().numpy()
    k = 2 * M + n_pass + 1
    assert np.array_equal(coll[:k], rec[:k]) and int(coll[k]) == rs_u.n and int(rec[k]) == N
    # ... and a distinct read got what its records got
    pick = torch.randint(0, N, (500_000,), device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(11))
    where = torch.searchsorted(rs_u.words[0], rs.words[0, pick])
    assert torch.equal(rs_u.words[0][where], rs.words[0, pick])
    for x_u, x_r in zip((res_u.pass_id, res_u.ref_id, res_u.pos, res_u.mm), a):
        assert torch.equal(x_u[where], x_r[pick])
    eng.close()


def test_properties_at_the_a2i_size(native_lib):
    """BASELINE configs[4] at its full size -- 50 M reads with seeded A->G edits, mouse-seeded libraries, cascade + count
    tally + A-to-I position tally -- sharded 8 ways as its 8-GPU form does: the shards' count vectors and edit-tally
    vectors add up to the whole job's (the one all-reduce's arithmetic), a shard's reads get what they got in the whole
    job, and a permutation of the reads changes nothing.  (Equality with oracle/edit_tally.c and the CPU port on all
    50 M reads is bench.py's gate.)"""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from mirge_amd import dist, synth
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    N = 50_000_000
    libs = synth.SynthLibraries(seed=synth.MOUSE_SEED, scale=1.0, shapes=synth.MOUSE_SHAPES)
    keys = list(synth.LIB_KEYS)
    with ThreadPoolExecutor(max_workers=len(keys)) as pool:
        fut = {k: pool.submit(FmIndex.build, *libs.libs[k]) for k in keys}
        words, lens, quant = synth.global_read_slice(libs, N, 0, N, workload="cascade", seed0=4355, mix=synth.A2I_MIX, n_samples=1)
        index = {k: f.result() for k, f in fut.items()}
    eng = Engine(0)
    for k in keys:
        eng.add_library(k, index[k])
    passes = eng.mirge_passes()
    M = index["mirna"].n_ref
    rs = ReadSet(words, lens, None, quant, device=eng.device)
    res = eng.cascade(rs, passes)
    whole = eng.tally(rs, res, M).cpu().numpy()
    edits = eng.edit_tally(rs, res, "mirna").cpu().numpy()
    assert int(edits.sum()) > 1_000_000 and int((res.pass_id == 8).sum().item()) > 100_000   # (the edited reads: found by the 2-mismatch pass)
    a = [t.clone() for t in (res.pass_id, res.ref_id, res.pos, res.mm)]
    total, total_e = np.zeros_like(whole), np.zeros_like(edits)
    for r in range(8):
        lo, hi = dist.shard_bounds(N, r, 8)
        rs_r = ReadSet.from_device(rs.words[:, lo:hi].contiguous(), rs.lens[lo:hi], None, rs.quant[lo:hi], 22, 22)
        res_r = eng.cascade(rs_r, passes)
        total += eng.tally(rs_r, res_r, M).cpu().numpy()
        total_e += eng.edit_tally(rs_r, res_r, "mirna").cpu().numpy()
        assert torch.equal(res_r.pass_id, a[0][lo:hi]) and torch.equal(res_r.ref_id, a[1][lo:hi]) and torch.equal(res_r.pos, a[2][lo:hi]) and \
            torch.equal(res_r.mm, a[3][lo:hi])
    assert np.array_equal(total, whole) and np.array_equal(total_e, edits)
    perm = torch.randperm(N, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(9))
    rs_q = ReadSet.from_device(rs.words[:, perm].contiguous(), rs.lens[perm], None, rs.quant[perm], 22, 22)
    res_q = eng.cascade(rs_q, passes)
    assert torch.equal(res_q.pass_id, a[0][perm]) and torch.equal(res_q.pos, a[2][perm])
    assert np.array_equal(eng.tally(rs_q, res_q, M).cpu().numpy(), whole)
    assert np.array_equal(eng.edit_tally(rs_q, res_q, "mirna").cpu().numpy(), edits)
    eng.close()


def test_properties_on_the_repeats_set(native_lib):
    """bench.py --workload repeats at its full size (20 M reads; full-size libraries with poly-A/T tails, tandem repeats,
    paralog families and the 10^5-copy element): the reads whose seeds meet repeats are answered behind the stream, by
    whichever wave happened to take them -- so a permutation of the reads, a split into 8 shards and a second run all
    have to give every read the same answer, and the lane has to have run (walk_diag counts its records)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from mirge_amd import dist, synth
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    N = 20_000_000
    libs = synth.SynthLibraries(seed=20181, scale=1.0, repeats=True)
    keys = list(synth.LIB_KEYS)
    with ThreadPoolExecutor(max_workers=len(keys)) as pool:
        fut = {k: pool.submit(FmIndex.build, *libs.libs[k]) for k in keys}
        words, lens, quant = synth.global_read_slice(libs, N, 0, N, workload="cascade", seed0=355, mix=None, n_samples=1)
        index = {k: f.result() for k, f in fut.items()}
    eng = Engine(0)
    for k in keys:
        eng.add_library(k, index[k])
    eng.set_option("walk_diag", 1)
    passes = eng.mirge_passes()
    M = index["mirna"].n_ref
    rs = ReadSet(words, lens, None, quant, device=eng.device)
    res = eng.cascade(rs, passes)
    whole = eng.tally(rs, res, M).cpu().numpy()
    st = res.stats
    assert st[6]["steps"] > 50_000, st[6]        # records left behind the stream in the ncRNA-others / mRNA launch
    a = [t.clone() for t in (res.pass_id, res.ref_id, res.pos, res.mm)]
    res2 = eng.cascade(rs, passes)
    for x, y in zip(a, (res2.pass_id, res2.ref_id, res2.pos, res2.mm)):
        assert torch.equal(x, y)
    total = np.zeros_like(whole)
    for r in range(8):
        lo, hi = dist.shard_bounds(N, r, 8)
        rs_r = ReadSet.from_device(rs.words[:, lo:hi].contiguous(), rs.lens[lo:hi], None, rs.quant[lo:hi], 22, 22)
        res_r = eng.cascade(rs_r, passes)
        total += eng.tally(rs_r, res_r, M).cpu().numpy()
        assert torch.equal(res_r.pass_id, a[0][lo:hi]) and torch.equal(res_r.ref_id, a[1][lo:hi]) and torch.equal(res_r.pos, a[2][lo:hi]) and \
            torch.equal(res_r.mm, a[3][lo:hi])
    assert np.array_equal(total, whole)
    perm = torch.randperm(N, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(3))
    rs_q = ReadSet.from_device(rs.words[:, perm].contiguous(), rs.lens[perm], None, rs.quant[perm], 22, 22)
    res_q = eng.cascade(rs_q, passes)
    assert torch.equal(res_q.pass_id, a[0][perm]) and torch.equal(res_q.ref_id, a[1][perm]) and torch.equal(res_q.pos, a[2][perm]) and \
        torch.equal(res_q.mm, a[3][perm])
    # the collapse order (equal reads side by side: the lane's records bunch up in few waves)
    order = torch.argsort(rs.words[0], stable=True)
    rs_s = ReadSet.from_device(rs.words[:, order].contiguous(), rs.lens[order], None, rs.quant[order], 22, 22)
    res_s = eng.cascade(rs_s, passes)
    assert torch.equal(res_s.pass_id, a[0][order]) and torch.equal(res_s.ref_id, a[1][order]) and torch.equal(res_s.pos, a[2][order]) and \
        torch.equal(res_s.mm, a[3][order])
    eng.close()


def test_properties_on_the_varlen_set(native_lib):
    """bench.py --workload varlen at its full size (20 M reads of 16..40 nt, two words per read: the batch is split on the
    device into the one-word lane -- dictionary kernels -- and the rest -- FM kernels): the lanes' lists are built by
    whichever workgroup took a read, so shards and a permutation must not change any read's answer; and the reads of
    33..40 nt get the same answers on the dictionary kernels' LONG instantiations (long_lane = 1) as on the FM kernels."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from mirge_amd import dist, synth
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    N = 20_000_000
    libs = synth.SynthLibraries(seed=20181, scale=1.0)
    keys = list(synth.LIB_KEYS)
    with ThreadPoolExecutor(max_workers=len(keys)) as pool:
        fut = {k: pool.submit(FmIndex.build, *libs.libs[k]) for k in keys}
        words, lens, quant = synth.global_read_slice(libs, N, 0, N, workload="varlen", seed0=355, mix=None, n_samples=1)
        index = {k: f.result() for k, f in fut.items()}
    assert words.shape[0] == 2 and int(lens.min()) == 16 and int(lens.max()) == 40
    eng = Engine(0)
    for k in keys:
        eng.add_library(k, index[k])
    passes = eng.mirge_passes()
    M = index["mirna"].n_ref
    rs = ReadSet(words, lens, None, quant, device=eng.device)
    res = eng.cascade(rs, passes)
    whole = eng.tally(rs, res, M).cpu().numpy()
    st = res.stats
    assert st[1]["aligned"] > 100_000, st[1]          # (the hairpin pass, len > 25, has work here)
    a = [t.clone() for t in (res.pass_id, res.ref_id, res.pos, res.mm)]
    long_hit = int(((rs.lens > 32) & (a[0] >= 0)).sum().item())
    assert long_hit > 1_000_000
    eng.set_option("long_lane", 1)
    res_l = eng.cascade(rs, passes)
    for x, y in zip(a, (res_l.pass_id, res_l.ref_id, res_l.pos, res_l.mm)):
        assert torch.equal(x, y)
    assert np.array_equal(eng.tally(rs, res_l, M).cpu().numpy(), whole)
    for lane in (1, 0):
        eng.set_option("long_lane", lane)
        total = np.zeros_like(whole)
        for r in range(8):
            lo, hi = dist.shard_bounds(N, r, 8)
            rs_r = ReadSet.from_device(rs.words[:, lo:hi].contiguous(), rs.lens[lo:hi], None, rs.quant[lo:hi], 16, 40)
            res_r = eng.cascade(rs_r, passes)
            total += eng.tally(rs_r, res_r, M).cpu().numpy()
            assert torch.equal(res_r.pass_id, a[0][lo:hi]) and torch.equal(res_r.ref_id, a[1][lo:hi]) and torch.equal(res_r.pos, a[2][lo:hi]) and \
                torch.equal(res_r.mm, a[3][lo:hi])
        assert np.array_equal(total, whole)
    perm = torch.randperm(N, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(5))
    rs_q = ReadSet.from_device(rs.words[:, perm].contiguous(), rs.lens[perm], None, rs.quant[perm], 16, 40)
    res_q = eng.cascade(rs_q, passes)
    assert torch.equal(res_q.pass_id, a[0][perm]) and torch.equal(res_q.ref_id, a[1][perm]) and torch.equal(res_q.pos, a[2][perm]) and \
        torch.equal(res_q.mm, a[3][perm])
    eng.close()
