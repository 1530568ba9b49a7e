"""f2: the reference's library layout ships only bowtie 1 `.1.ebwt` files
(MAIN:262-281).  mirge_amd reads entry names and sequences back out of them (bowtie-inspect's job)
so that layout works unchanged.  No bowtie-built index exists in the image: the reader is
validated by ROUND TRIP only, against tests/helpers/ebwt_writer.cpp -- a stand-alone restatement of
bowtie-build's layout that shares no code with the reader and fills the fields a real index has
(side occurrence counts, fchr, ftab, eftab).  The reader re-derives all of them from the BWT it
decoded and must refuse a file in which any of them is off."""
import os
import struct

import numpy as np
import pytest

from mirge_amd import synth
from mirge_amd.index import FmIndex


def random_entries(rng, n=40):
    names, seqs = [], []
    for k in range(n):
        L = int(rng.integers(1, 400))
        s = "".join("ACGT"[c] for c in rng.integers(0, 4, L))
        names.append("entry-%d some description" % k if k % 5 == 0 else "entry-%d" % k)
        seqs.append(s)
    seqs[3] = "NNN" + seqs[3] + "NN"                # leading / trailing gaps
    seqs[7] = seqs[7][:5] + "NNNNNN" + seqs[7][5:]  # an inner gap: two fragments
    seqs[9] = "NNNN"                                # nothing but gaps
    seqs[11] = "A"                                  # a single base
    return names, seqs


@pytest.mark.parametrize("line_rate,lines_per_side,ftab_chars", [(6, 1, 4), (6, 2, 3), (5, 1, 5), (7, 1, 2)])
def test_round_trip_names_sequences_and_n_runs(native_lib, ebwt_writer, tmp_path, line_rate, lines_per_side, ftab_chars):
    names, seqs = random_entries(np.random.default_rng(12))
    prefix = str(tmp_path / "lib")
    ebwt_writer(prefix, names, seqs, ftab_chars=ftab_chars, line_rate=line_rate, lines_per_side=lines_per_side)
    assert os.path.isfile(prefix + ".1.ebwt")
    back = FmIndex.from_ebwt(prefix)
    assert back.names == [n.split()[0] for n in names]   # bowtie reports a name up to its first blank
    assert [back.sequence(i) for i in range(back.n_ref)] == [s.upper() for s in seqs]


def test_library_sized_round_trip_and_prefix_resolution(native_lib, ebwt_writer, tmp_path):
    """A synthetic miRNA library (2 980 entries, 88 kbp) through `.1.ebwt` with bowtie-build's default
    ftabChars, opened the way the command line opens a library prefix.  An index recovered from a
    `.1.ebwt` is NOT cached next to the library (the reader is unpinned)."""
    libs = synth.SynthLibraries(scale=1.0)
    names, seqs = libs.libs["mirna"]
    prefix = str(tmp_path / "hsa_mirna_miRBase")
    ebwt_writer(prefix, names, seqs)
    ix = FmIndex.open_prefix(prefix, cache=True)          # only <prefix>.1.ebwt exists
    assert ix.names == names and ix.name_seq_dict() == dict(zip(names, seqs))
    assert not os.path.isfile(prefix + ".mrgfm")
    again = FmIndex.open_prefix(prefix)
    assert again.names == names and again.sequence(17) == seqs[17]


def test_rejects_foreign_truncated_and_inconsistent_files(native_lib, ebwt_writer, tmp_path):
    """Every redundant field is checked against the decoded BWT: a flipped BWT character, a wrong
    occurrence count inside a side, a wrong fchr entry, a descending ftab, a fragment table that does
    not tile the text -- each one is an error, not a silently different library."""
    from mirge_amd._native import MirgeAmdError
    p = str(tmp_path / "x")
    with open(p + ".1.ebwt", "wb") as fh:
        fh.write(b"\x02\x00\x00\x00 not an index")
    with pytest.raises(MirgeAmdError):
        FmIndex.from_ebwt(p)
    good = str(tmp_path / "good")
    names, seqs = random_entries(np.random.default_rng(5), n=30)
    ebwt_writer(good, names, seqs, ftab_chars=3)
    assert FmIndex.from_ebwt(good).n_ref == 30
    blob = bytearray(open(good + ".1.ebwt", "rb").read())

    def broken(mutate, what):
        b = bytearray(blob)
        mutate(b)
        with open(p + ".1.ebwt", "wb") as fh:
            fh.write(b)
        with pytest.raises(MirgeAmdError) as e:
            FmIndex.from_ebwt(p)
        assert what in str(e.value), (what, str(e.value))

    broken(lambda b: b.__delitem__(slice(len(b) // 2, None)), "truncated")
    n_pat, = struct.unpack_from("<I", blob, 28)
    n_frag, = struct.unpack_from("<I", blob, 32 + 4 * n_pat)
    ebwt_at = 36 + 4 * n_pat + 12 * n_frag
    length, = struct.unpack_from("<I", blob, 4)
    n_pairs = (length + 1 + 2 * 224 - 1) // (2 * 224)
    after_bwt = ebwt_at + n_pairs * 128

    def flip_base(b):
        b[ebwt_at + 3] ^= 0x0C          # one BWT character of the first (backward) side
    broken(flip_base, "disagree with the BWT")

    def bump_count(b):
        struct.pack_into("<I", b, ebwt_at + 64 + 56, struct.unpack_from("<I", b, ebwt_at + 64 + 56)[0] + 1)   # G count, first pair
    broken(bump_count, "occurrence counts")

    def bump_fchr(b):
        struct.pack_into("<I", b, after_bwt + 4 + 8, struct.unpack_from("<I", b, after_bwt + 4 + 8)[0] + 1)
    broken(bump_fchr, "fchr")

    def ftab_descends(b):
        struct.pack_into("<I", b, after_bwt + 4 + 20 + 4 * 10, length + 5)
    broken(ftab_descends, "ftab")

    def shift_fragment(b):
        struct.pack_into("<I", b, 36 + 4 * n_pat, 1)   # the first fragment no longer starts at base 0
    broken(shift_fragment, "fragment")
