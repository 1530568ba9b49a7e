"""CPU suite: the exhaustive-scan model, the FM port and the index builder agree.

(The aligner boundary is PARITY UNPINNED against bowtie itself -- see
oracle/__init__.py; these tests pin our two independent restatements to each
other and the index layout to a naive construction.)"""
import numpy as np
import pytest

from oracle import cascade, model
from tests.util import LIB_ORDER, World


@pytest.fixture(scope="module")
def world(native_lib, oracle_lib):
    return World()


def test_suffix_array_and_bwt_match_naive(native_lib):
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(3)
    seqs = ["".join("ACGT"[c] for c in rng.integers(0, 4, int(L))) for L in rng.integers(1, 90, 40)]
    seqs[3] = seqs[3][:5] + "NNN" + seqs[3][5:]
    seqs[7] = "A" * 70          # low complexity
    seqs[8] = seqs[7]           # exact duplicate entry
    ix = FmIndex.build(["e%d" % i for i in range(len(seqs))], seqs)
    v = ix.view()
    text = "".join(s.replace("N", "") for s in seqs)
    assert v["n"] == len(text)
    t = text + "$"
    naive = sorted(range(len(t)), key=lambda i: t[i:])
    sa = v["sa"]
    assert [int(x) & 0xFFFFFFFF for x in sa] == naive
    # occ blocks (16 B / 32 rows, uint16 counts relative to the superblock) + superblocks
    # reproduce C[c] + rank over the BWT
    bwt = [t[i - 1] if i else "$" for i in naive]
    blocks = v["blocks"].reshape(-1, 4)
    sup = v["super"].reshape(-1, 4)
    for c, ch in enumerate("ACGT"):
        first_row = 1 + sum(x < ch for x in text)
        assert v["C"][c] == first_row
        run = 0
        for i, b in enumerate(bwt):
            if i % 32 == 0:
                pair = int(blocks[i // 32][c >> 1])
                cnt = (pair >> 16) if (c & 1) else (pair & 0xFFFF)
                assert int(sup[i >> 16][c]) + cnt == first_row + run
            run += b == ch
    assert bwt[v["primary"]] == "$"
    # suffix-array rows carry the distance to both ends of their N-free segment
    seg_start = [int(x) for x in v["seg_start"]]
    for row in sa:
        row = int(row)
        p = row & 0xFFFFFFFF
        if p == len(text):
            continue
        sg = max(k for k in range(len(seg_start) - 1) if seg_start[k] <= p)
        assert (row >> 32) & 255 == min(255, p - seg_start[sg])
        assert (row >> 40) & 255 == min(255, seg_start[sg + 1] - p)
        assert (row >> 48) == sg
    # bowtie-inspect equivalents
    assert [ix.sequence(i) for i in range(len(seqs))] == seqs
    assert ix.names == ["e%d" % i for i in range(len(seqs))]


def test_index_roundtrip_file(native_lib, tmp_path):
    from mirge_amd.index import FmIndex
    ix = FmIndex.build(["a", "b"], ["ACGTTGCANNACGT", "GGGTTTAAACCC"])
    p = str(tmp_path / "x.mrgfm")
    ix.save(p)
    iy = FmIndex.load(p)
    va, vb = ix.view(), iy.view()
    for k in ("blocks", "super", "text", "sa", "seg_start", "seg_ref", "seg_off", "chunk_seg"):
        assert np.array_equal(va[k], vb[k])
    assert iy.names == ["a", "b"] and iy.sequence(0) == "ACGTTGCANNACGT"


def test_fm_port_equals_exhaustive_scan(world):
    """Every read: same claiming pass, entry, offset and mismatch count; same
    per-pass processed/aligned counters (runAnnotationPipeline.py:648-650)."""
    res = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask)
    libs = {k: model.Library(*world.libs.libs[k]) for k in LIB_ORDER}
    seq_dic = {r: cascade.new_seq_record(r, 1) for r in world.reads}
    log_dic = {"quantStats": [{}], "annotStats": []}
    align = {}
    cascade.run_annotation_pipeline(seq_dic, libs, log_dic, align_dic=align)
    for i, r in enumerate(world.reads):
        got = None
        if res["pass_id"][i] >= 0:
            got = (int(res["pass_id"][i]), int(res["ref_id"][i]), int(res["pos"][i]), int(res["mm"][i]))
        assert align.get(r) == got, r
    for i, st in enumerate(log_dic["annotStats"]):
        assert st["readsProcessed"] == int(res["stats"][i][0])
        assert st["readsAligned"] == int(res["stats"][i][1])
    # every pass of the cascade claimed something, so all nine policies were exercised
    assert all(int(res["stats"][i][1]) > 0 for i in range(9))


def test_fm_port_equals_exhaustive_scan_on_reads_of_eight_words(native_lib, oracle_lib):
    """Reads of 129..255 nt (what `-ad none` leaves of a 151- or 250-cycle run; RAP:543-554 caps no pass but the
    first at a length): the FM port on eight packed words = the exhaustive scan on the strings -- seed in the
    first 28 bases, mismatches behind it count towards the total only, Ns mismatch."""
    from mirge_amd import pack
    from tests.util import pass_dicts
    w = World(with_n=True, n_fixed=50, n_var=20)
    rng = np.random.default_rng(77)
    reads = []
    for key in ("mrna", "ncrna_others", "snorna", "rrna"):
        for s in w.libs.libs[key][1]:
            if len(s) < 130:
                continue
            for _ in range(2):
                L = int(rng.integers(129, min(255, len(s)) + 1))
                o = int(rng.integers(0, len(s) - L + 1))
                r = list(s[o:o + L])
                for _ in range(int(rng.integers(0, 4))):
                    r[int(rng.integers(0, L))] = "ACGTN"[int(rng.integers(0, 5))]
                reads.append("".join(r))
    reads = list(dict.fromkeys(reads))
    reads = [reads[i] for i in rng.permutation(len(reads))[:500]] + ["A" * 255, w.libs.libs["mrna"][1][0][:255]]
    words, lens, nmask = pack.pack_reads(reads)
    assert words.shape[0] == 8 and int(lens.max()) == 255 and int(lens.min()) >= 129
    res = model.fm_cascade(w.views, pass_dicts(), words, lens, nmask)
    libs = {k: model.Library(*w.libs.libs[k]) for k in LIB_ORDER}
    seq_dic = {r: cascade.new_seq_record(r, 1) for r in reads}
    align = {}
    cascade.run_annotation_pipeline(seq_dic, libs, {"quantStats": [{}], "annotStats": []}, align_dic=align)
    for i, r in enumerate(reads):
        got = None
        if res["pass_id"][i] >= 0:
            got = (int(res["pass_id"][i]), int(res["ref_id"][i]), int(res["pos"][i]), int(res["mm"][i]))
        assert align.get(r) == got, r
    claimed = [int(res["stats"][i][1]) for i in range(9)]
    assert sum(claimed) > len(reads) // 3 and claimed[4] > 0 and claimed[6] > 0 and claimed[7] > 0   # snoRNA, ncRNA, mRNA
    assert res["pass_id"][len(reads) - 1] == 7 and res["pos"][len(reads) - 1] == 0 and res["pass_id"][len(reads) - 2] < 0


def test_index_arrays_are_the_fm_index_of_the_strings(world):
    """oracle/index_check.c: every array the GPU uploads and the CPU port reads, checked by definition against
    the libraries' strings (entries with N runs: several segments) -- and a single wrong value in any of them
    is found (the port and the kernels share these arrays: a construction bug would not show in their
    comparison)."""
    for k, v in zip(LIB_ORDER, world.views):
        rep = model.check_index(v, world.libs.libs[k][1])
        assert rep["rows"] == v["n"] + 1 and rep["jump_entries"] > 0
    from mirge_amd.index import FmIndex
    names, seqs = world.libs.libs["ncrna_others"]
    seqs = list(seqs)
    for r in range(0, len(seqs), 7):                # N runs: several segments per entry, one entry starting with N
        s = seqs[r]
        seqs[r] = ("N" if r % 14 == 0 else "") + s[:40] + "NNN" + s[43:90] + "n" + s[91:]
    v = FmIndex.build(names, seqs).view()
    assert len(v["seg_ref"]) > len(seqs) + 100
    model.check_index(v, seqs)

    def broken(key, edit):
        w = dict(v)
        w[key] = np.array(v[key], copy=True)
        edit(w[key])
        with pytest.raises(AssertionError, match="index check failed"):
            model.check_index(w, seqs)

    def swap_rows(a):
        a[1000], a[1001] = int(a[1001]), int(a[1000])

    def flip(i, bit):
        def f(a):
            a[i] ^= type(a[i])(bit)
        return f
    broken("sa", swap_rows)                                      # two suffixes out of order
    broken("sa", flip(5000, 1 << 32))                            # a row's distance to its segment start
    broken("sa", flip(5000, 1 << 48))                            # a row's segment id
    broken("sa", lambda a: a.__setitem__(7, a[8]))               # a position twice
    broken("text", flip(300, 1 << 7))                            # one base
    broken("blocks", flip(4 * 100 + 2, 1 << 5))                  # one BWT bit
    broken("blocks", flip(4 * 100 + 1, 1))                       # one 16-bit count
    broken("super", flip(5, 1))
    broken("ftab", flip(12345, 1))                               # one jump-table boundary
    broken("ftab", flip(len(v["ftab"]) - 1, 1))
    broken("chunk_seg", flip(50, 1))
    broken("seg_off", flip(3, 1))
    v0 = world.views[LIB_ORDER.index("mirna")]
    w = dict(v0, kbits=np.array(v0["kbits"], copy=True))
    set_word = int(np.flatnonzero(w["kbits"])[0])
    w["kbits"][set_word] &= w["kbits"][set_word] - np.uint32(1)  # a 9-mer of the text missing from the bitmap
    with pytest.raises(AssertionError, match="9-mer bitmap"):
        model.check_index(w, world.libs.libs["mirna"][1])
    w = dict(v, primary=int(v["primary"]) + 1)
    with pytest.raises(AssertionError, match="BWT"):
        model.check_index(w, seqs)


def test_pair_seed_port_equals_piece_search(world):
    """The 2-mismatch pass searched through anchor pairs (mrg_pass_stats.pair_anchor = 4) claims the
    same reads at the same place as the stratum-first pigeonhole search, without an LF step for the
    reads long enough to hold the four anchors."""
    base = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask, wstop=8, ftab=True)
    passes = [dict(p, pair_anchor=4 if p["max_mm_seed"] == 2 else 0) for p in world.passes]
    alt = model.fm_cascade(world.views, passes, world.words, world.lens, world.nmask, wstop=8, ftab=True)
    for k in ("pass_id", "ref_id", "pos", "mm"):
        assert np.array_equal(base[k], alt[k]), k
    assert np.array_equal(base["stats"][:, :2], alt["stats"][:, :2])
    assert np.array_equal(base["stats"][:8], alt["stats"][:8])
    # (at this library size -- 2.5 K bases -- an 8-base key is no sharper than a 6-base piece; at
    # 84 K bases the pairs leave ~8 candidate rows per read against ~50, bench.py's passes[8])
    assert int(alt["stats"][8][2]) < int(base["stats"][8][2])


def test_search_shortcuts_do_not_change_results(world):
    base = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask)
    for wstop, ftab in ((1, False), (4, False), (64, False), (0, True), (2, True)):
        alt = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask,
                               wstop=wstop, ftab=ftab)
        for k in ("pass_id", "ref_id", "pos", "mm"):
            assert np.array_equal(base[k], alt[k])
        assert int(alt["stats"][:, 2].sum()) < int(base["stats"][:, 2].sum())
        assert (int(alt["stats"][:, 4].sum()) > 0) == ftab


def test_jump_table_equals_backward_search(native_lib):
    """[T[c], T[c+1]) is the BWT interval a step-by-step backward search of k-mer c ends in, plus
    at most the few suffixes shorter than k that the k-mer continues (end of the text)."""
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(11)
    seqs = ["".join("ACGT"[c] for c in rng.integers(0, 4, int(L))) for L in rng.integers(20, 400, 60)]
    ix = FmIndex.build(["e%d" % i for i in range(len(seqs))], seqs)
    v = ix.view()
    assert v["ftab_ks"] == [9, 8, 6, 4] and len(v["ftab"]) == (4 ** 9 + 1) + (4 ** 8 + 1) + (4 ** 6 + 1) + (4 ** 4 + 1)
    text = "".join(seqs)
    sa = [int(x) & 0xFFFFFFFF for x in v["sa"]]
    tail = text[-9:]
    for trial in range(450):
        base = 4 ** 9 + 1
        k, off = ((8, base), (6, base + 4 ** 8 + 1), (4, base + 4 ** 8 + 1 + 4 ** 6 + 1), (9, 0))[trial % 4]
        if trial < 12:
            kmer = (tail[-(trial % 7 + 1):] + "A" * 9)[:k]      # continues a too-short suffix
        elif rng.random() < 0.7:
            p = int(rng.integers(0, len(text) - k))
            kmer = text[p:p + k]
        else:
            kmer = "".join("ACGT"[c] for c in rng.integers(0, 4, k))
        code = sum("ACGT".index(ch) << (2 * (k - 1 - t)) for t, ch in enumerate(kmer))
        lo, hi = int(v["ftab"][off + code]), int(v["ftab"][off + code + 1])
        rows = [i for i, s in enumerate(sa) if text[s:s + k] == kmer]
        extra = [i for i in range(lo, hi) if i not in rows]
        assert set(rows) <= set(range(lo, hi))
        assert all(len(text) - sa[i] < k for i in extra) and len(extra) < k + 1     # only short suffixes
    assert int(v["ftab"][0]) == 0 and int(v["ftab"][4 ** 9]) == len(sa)


def test_kmer_bitmap_is_exact_and_changes_no_result(world):
    """The 9-mer presence bitmap of a small library: bit c set iff the 9-mer occurs in the stored
    text; with it the port skips jump-table loads but assigns the same."""
    v = world.views[0]
    assert v["kbits"] is not None
    assert [w["kbits"] is not None for w in world.views] == [w["n"] <= 190000 for w in world.views]
    text = "".join(s.replace("N", "") for s in world.libs.libs["mirna"][1])
    have = {sum("ACGT".index(ch) << (2 * t) for t, ch in enumerate(text[p:p + 9])) for p in range(len(text) - 8)}
    bits = np.unpackbits(v["kbits"].view(np.uint8), bitorder="little")
    assert set(np.nonzero(bits)[0].tolist()) == have
    on = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask, wstop=8, ftab=True)
    off = model.fm_cascade(world.views, world.passes, world.words, world.lens, world.nmask, wstop=8, ftab=True,
                           kmer_filter=False)
    for k in ("pass_id", "ref_id", "pos", "mm"):
        assert np.array_equal(on[k], off[k])
    assert int(on["stats"][:, 4].sum()) < int(off["stats"][:, 4].sum()) * 0.7     # far fewer table loads


def test_big_jump_table_same_results(native_lib, oracle_lib):
    """A library beyond 4^11 bases gets the k = 12 table (whole-read seeds) next to the k = 11 one
    (pieces of a 22-nt read); results equal the plain backward search."""
    from mirge_amd import pack
    from mirge_amd.index import FmIndex
    from tests.util import BIG_PASSES, big_library_case
    names, seqs, reads = big_library_case()
    ix = FmIndex.build(names, seqs)
    v = ix.view()
    assert v["ftab_ks"] == [12, 11, 6, 4] and len(v["ftab"]) == 4 ** 12 + 4 ** 11 + 4 ** 6 + 4 ** 4 + 4
    w, l, nm = pack.pack_reads(reads)
    base = model.fm_cascade([v], BIG_PASSES, w, l, nm)
    alt = model.fm_cascade([v], BIG_PASSES, w, l, nm, wstop=2, ftab=True)
    for k in ("pass_id", "ref_id", "pos", "mm"):
        assert np.array_equal(base[k], alt[k]), k
    assert all(int(alt["stats"][i][1]) > 100 for i in range(3))
    assert int(alt["stats"][:, 2].sum()) < int(base["stats"][:, 2].sum()) // 4
    # context rows (>= 2^20 bases): the port against the exhaustive scan on the reads that touch
    # entry boundaries and the planted N
    from tests.util import special_reads_of_big_case
    # the row-context array of a library of >= 2^20 bases: 8 bases left of each row's position and
    # the bases 8..15 after it, as stored text (entries back to back, N-free segments only)
    text_codes = np.concatenate([np.frombuffer(s.replace("N", "").encode(), dtype=np.uint8) for s in seqs])
    lut = np.zeros(256, dtype=np.uint32)
    lut[ord("C")], lut[ord("G")], lut[ord("T")] = 1, 2, 3
    tc = np.concatenate([np.zeros(8, np.uint32), lut[text_codes], np.zeros(24, np.uint32)])
    pos = (v["sa"] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    sample = np.random.default_rng(1).integers(0, len(pos), 5000)
    for i in sample:
        p0 = int(pos[i]) + 8                       # index into the padded array
        left = sum(int(tc[p0 - d]) << (16 - 2 * d) for d in range(1, 9))
        right = sum(int(tc[p0 + 8 + d]) << (2 * d) for d in range(8))
        assert int(v["ctx"][i]) == (left | right << 16), i
    olib = model.Library(names, seqs)
    special = special_reads_of_big_case(reads)
    sw, sl, sn = pack.pack_reads(special, 2)
    for pol in BIG_PASSES:
        got = model.fm_cascade([v], [dict(pol, min_len=0, max_len=255)], sw, sl, sn, wstop=2, ftab=True)
        ref, pos, mm = model.align_batch(olib, special, pol["seed_len"], pol["max_mm_seed"], pol["max_mm_total"])
        want_pass = np.where(ref >= 0, 0, -1)
        assert np.array_equal(got["pass_id"], want_pass)
        hit = ref >= 0
        assert np.array_equal(got["ref_id"][hit], ref[hit]) and np.array_equal(got["pos"][hit], pos[hit])
        assert np.array_equal(got["mm"][hit], mm[hit])
    assert hit.sum() > 20 and (~hit).sum() > 3
