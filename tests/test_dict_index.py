"""The exact-match dictionary (csrc/dict_index.cpp, mrg_index_get_dict) on the CPU: a lookup written
here from the layout documented in include/mirge_amd.h must return, for every read, what a plain
string search over the entries returns: the first entry that contains the read, at its lowest offset
(`-n 0` / `-v 0` with the tie rule of mrg_cascade_run; runAnnotationPipeline.py:577, :598, :688)."""
import numpy as np
import pytest

from mirge_amd import synth
from mirge_amd.index import FmIndex

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
MUL = 0x9E3779B1


def pack(read):
    v = 0
    for i, ch in enumerate(read):
        v |= CODE[ch] << (2 * i)
    return v


def lookup(d, read):
    """-> (ref, off) | None | 'fallback' (the FM index must answer)."""
    L, key = len(read), d["key_bases"]
    if L < key or L > 32:
        return "fallback"
    v = pack(read)
    log2 = d["log2_slots"]
    home = (((v & ((1 << (2 * key)) - 1)) * MUL) & 0xFFFFFFFF) >> (32 - log2)
    slots = d["slots"]
    chain = (int(slots[home, 1]) >> 32 >> 6) & 15
    if chain == 15:
        return "fallback"
    lmask = (1 << (2 * L)) - 1
    for j in range(chain + 1):
        win, hi = int(slots[(home + j) & ((1 << log2) - 1), 0]), int(slots[(home + j) & ((1 << log2) - 1), 1])
        meta = hi >> 32
        if (meta >> 10) & 1 and ((win ^ v) & lmask) == 0 and L <= (meta & 63):
            return (hi & 0xFFFFFFFF, meta >> 11)
    return None


def brute(seqs, read):
    for e, s in enumerate(seqs):
        # an alignment never spans an N
        o = s.find(read)
        if o >= 0:
            return (e, o)
    return None


def sample_reads(seqs, rng, n):
    reads = []
    for _ in range(n):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(16, 33))
        if len(s) < L:
            continue
        kind = rng.random()
        if kind < 0.2:
            o = 0
        elif kind < 0.4:
            o = len(s) - L
        else:
            o = int(rng.integers(0, len(s) - L + 1))
        r = s[o:o + L]
        if rng.random() < 0.3:  # one substitution: usually no exact hit any more
            i = int(rng.integers(0, L))
            r = r[:i] + "ACGT"[(CODE.get(r[i], 0) + 1) % 4] + r[i + 1:]
        if "N" not in r:
            reads.append(r)
    # reads that only exist across two neighbouring entries must not match
    for e in range(min(len(seqs) - 1, 200)):
        a, b = seqs[e], seqs[e + 1]
        if len(a) >= 11 and len(b) >= 11 and "N" not in a[-11:] + b[:11]:
            reads.append(a[-11:] + b[:11])
    reads += ["".join("ACGT"[c] for c in rng.integers(0, 4, 22)) for _ in range(n // 4)]
    return reads


@pytest.mark.parametrize("key", ["mirna", "pre_trna", "snorna"])
def test_lookup_equals_string_search(native_lib, key):
    libs = synth.SynthLibraries(scale=0.05)
    names, seqs = libs.libs[key]
    ix = FmIndex.build(names, seqs)
    d = ix.exact_dict(16)
    assert d["key_bases"] == 16 and d["n_overflow"] == 0
    rng = np.random.default_rng(11)
    n_hit = 0
    for r in sample_reads(seqs, rng, 3000):
        got, want = lookup(d, r), brute(seqs, r)
        assert got == want, (r, got, want)
        n_hit += want is not None
    assert n_hit > 1000


def test_paralogs_ns_and_short_keys(native_lib):
    """Identical entries (paralogous miRNAs) resolve to the first one; an N splits an entry into
    segments a read cannot span; the last bases of a segment are found for every length that still
    fits; key lengths other than 16 work the same."""
    core = "TGAGGTAGTAGGTTGTATAGTTACGTACGTAC"
    seqs = ["AC" + core, "GG" + core, "AC" + core, "TTTTGCGCGCGATATATCGCGNTGAGGTAGTAGGTTGTATAGTTGCA",
            "CATCATCATCATCATCATCATCATCATCATCATCAT", core[:20]]
    names = ["e%d" % i for i in range(len(seqs))]
    ix = FmIndex.build(names, seqs)
    for kb in (16, 12, 8):
        d = ix.exact_dict(kb)
        for e, s in enumerate(seqs):
            for L in range(kb, 33):
                for o in range(0, len(s) - L + 1):
                    r = s[o:o + L]
                    if "N" in r:
                        continue
                    got = lookup(d, r)
                    assert got == brute(seqs, r), (kb, r, got)
        assert lookup(d, "GCGATATATCGCGATGAGGTAG") is None          # would have to span the N
        assert lookup(d, core[:15]) == ("fallback" if kb == 16 else brute(seqs, core[:15]))
    d = ix.exact_dict(16)
    assert lookup(d, core[:22]) == (0, 2)


def test_repeats_overflow_to_the_fm_index(native_lib):
    """More distinct positions behind one key than a chain holds: the home slot says so and no
    lookup of that key claims a wrong 'absent'."""
    rng = np.random.default_rng(3)
    stem = "ACGTTGCAAGCTTGCA"  # one 16-base key ...
    seqs = [stem + "".join("ACGT"[c] for c in rng.integers(0, 4, 16)) for _ in range(40)]  # ... 40 different tails
    ix = FmIndex.build(["r%d" % i for i in range(40)], seqs)
    d = ix.exact_dict(16)
    assert d["n_overflow"] > 0
    assert lookup(d, seqs[7][:24]) == "fallback"
    # other keys of the same library are unaffected
    assert lookup(d, seqs[7][3:27]) == brute(seqs, seqs[7][3:27])


def test_too_large_or_bad_key_is_an_error(native_lib):
    from mirge_amd._native import MirgeAmdError
    ix = FmIndex.build(["a"], ["ACGTACGTACGTACGTACGTACGT"])
    with pytest.raises(MirgeAmdError):
        ix.exact_dict(17)
    with pytest.raises(MirgeAmdError):
        ix.exact_dict(7)


def test_large_library_is_filled_in_parallel(native_lib):
    """A library beyond 4 Mbp (the size class of the mRNA library, runAnnotationPipeline.py:584/598) is
    filled by several workers, each owning a range of home slots; a lookup still finds the first entry
    that holds the read at its lowest offset -- including repeats that sit in different workers' ranges
    or next to a range boundary -- and never claims a wrong 'absent'."""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [acgt[rng.integers(0, 4, 150_000)].tobytes().decode() for _ in range(30)]   # 4.5 Mbp
    # repeats: the same 40-mer in several entries (first occurrence wins), at various offsets
    rep = [acgt[rng.integers(0, 4, 40)].tobytes().decode() for _ in range(50)]
    for i, r in enumerate(rep):
        for e in (3 + i % 5, 11 + i % 7, 25):
            o = 1000 * (i + 1) + e
            seqs[e] = seqs[e][:o] + r + seqs[e][o + 40:]
    ix = FmIndex.build(["big%d" % i for i in range(len(seqs))], seqs)
    d = ix.exact_dict(16)
    assert d["log2_slots"] >= 24 and d["n_keys"] > 4_000_000
    joined = "\x00".join(seqs)
    starts = np.cumsum([0] + [len(s) + 1 for s in seqs])

    def first_hit(read):
        o = joined.find(read)
        if o < 0:
            return None
        e = int(np.searchsorted(starts, o, side="right")) - 1
        return (e, o - int(starts[e]))

    n_hit = 0
    reads = sample_reads(seqs, rng, 1500) + [r[:L] for r in rep for L in (16, 22, 32)] + [r[5:27] for r in rep]
    for r in reads:
        got = lookup(d, r)
        if got == "fallback":
            continue
        want = first_hit(r)
        assert got == want, (r, got, want)
        n_hit += want is not None
    assert n_hit > 700
