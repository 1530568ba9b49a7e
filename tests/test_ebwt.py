"""f2: the reference's library layout ships only bowtie 1 `.1.ebwt` files
(MAIN:262-281).  mirge_amd reads entry names and sequences back out of them (bowtie-inspect's job)
so that layout works unchanged.  No bowtie-built index exists in the image: the reader is
validated by ROUND TRIP with the in-tree test writer only (labelled so in ebwt.cpp)."""
import os

import numpy as np
import pytest

from mirge_amd import synth
from mirge_amd.index import FmIndex


def test_round_trip_names_sequences_and_n_runs(native_lib, tmp_path):
    rng = np.random.default_rng(12)
    names, seqs = [], []
    for k in range(40):
        L = int(rng.integers(1, 400))
        s = "".join("ACGT"[c] for c in rng.integers(0, 4, L))
        names.append("entry-%d some description" % k if k % 5 == 0 else "entry-%d" % k)
        seqs.append(s)
    seqs[3] = "NNN" + seqs[3] + "NN"                # leading / trailing gaps
    seqs[7] = seqs[7][:5] + "NNNNNN" + seqs[7][5:]  # an inner gap: two fragments
    seqs[9] = "NNNN"                                # nothing but gaps
    seqs[11] = "A"                                  # a single base
    src = FmIndex.build([n.split()[0] for n in names], seqs)
    prefix = str(tmp_path / "lib")
    src.write_ebwt_for_tests(prefix, ftab_chars=4)
    assert os.path.isfile(prefix + ".1.ebwt")
    back = FmIndex.from_ebwt(prefix)
    assert back.names == [n.split()[0] for n in names]
    assert [back.sequence(i) for i in range(back.n_ref)] == [s.upper() for s in seqs]


def test_library_sized_round_trip_and_prefix_resolution(native_lib, tmp_path):
    """A synthetic miRNA library (2 980 entries, 88 kbp) through `.1.ebwt` with bowtie-build's default
    ftabChars, opened the way the command line opens a library prefix."""
    libs = synth.SynthLibraries(scale=1.0)
    names, seqs = libs.libs["mirna"]
    prefix = str(tmp_path / "hsa_mirna_miRBase")
    FmIndex.build(names, seqs).write_ebwt_for_tests(prefix)
    ix = FmIndex.open_prefix(prefix, cache=True)          # only <prefix>.1.ebwt exists
    assert ix.names == names and ix.name_seq_dict() == dict(zip(names, seqs))
    assert os.path.isfile(prefix + ".mrgfm")               # cached for the next run
    again = FmIndex.open_prefix(prefix)
    assert again.names == names and again.sequence(17) == seqs[17]


def test_rejects_foreign_and_truncated_files(native_lib, tmp_path):
    from mirge_amd._native import MirgeAmdError
    p = str(tmp_path / "x")
    with open(p + ".1.ebwt", "wb") as fh:
        fh.write(b"\x02\x00\x00\x00 not an index")
    with pytest.raises(MirgeAmdError):
        FmIndex.from_ebwt(p)
    good = str(tmp_path / "good")
    FmIndex.build(["a", "b"], ["ACGTACGTTTGA", "GGGATTTACA"]).write_ebwt_for_tests(good, ftab_chars=3)
    blob = open(good + ".1.ebwt", "rb").read()
    with open(p + ".1.ebwt", "wb") as fh:
        fh.write(blob[:len(blob) // 2])
    with pytest.raises(MirgeAmdError):
        FmIndex.from_ebwt(p)
