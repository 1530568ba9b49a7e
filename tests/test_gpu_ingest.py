"""GPU parity (-m gpu) of the device-side FASTQ ingest (csrc/ingest.hip, mrg_fastq_parse_device): record
splitting, the 3' quality rule, the `-ad +N` cutter, adapter sequences (cutadapt's 3' search), the 16-nt minimum and the 2-bit packing of
trim_file.py:24-66,89-134 / quantReads.py:4-24 on raw text blocks.  Checked against the oracle's Python
restatement (oracle/ingest.py) and, array for array, against the host parser (csrc/fastq.cpp)."""
import gzip
import os

import numpy as np
import pytest

from mirge_amd import ingest, pack
from oracle import ingest as oingest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(native_lib):
    from mirge_amd.engine import Engine
    return Engine(0)


ILLUMINA = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"   # `-ad illumina` (__main__.py:123-127)


def make_fastq(path, rng, n=6000, max_len=50, phred=33, crlf=False, final_newline=True, long_reads=0, adapters=()):
    eol = "\r\n" if crlf else "\n"
    recs = []
    for i in range(n):
        L = int(rng.integers(14, max_len + 1))
        seq = "".join("ACGTN"[c] for c in rng.choice(5, L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        if adapters and rng.random() < 0.8:
            # an insert followed by (a prefix of) an adapter, sometimes with an error in it, sometimes run off the 3' end
            ad = list(adapters[int(rng.integers(0, len(adapters)))])
            kind = rng.random()
            if kind < 0.3:
                k = int(rng.integers(0, len(ad)))
                ad[k] = "ACGT"[("ACGT".index(ad[k]) + 1) % 4] if kind < 0.2 else ""   # substitution / deletion
            ins = int(rng.integers(0, 36))
            seq = (seq[:ins] + "".join(ad))[:max(L, int(rng.integers(14, max_len + 1)))]
            L = len(seq)
        if rng.random() < 0.1:
            seq = seq.lower()
        q = rng.integers(20, 41, L)
        if rng.random() < 0.5:   # a low-quality 3' tail (sometimes the whole read)
            t = int(rng.integers(0, L + 1))
            q[t:] = rng.integers(0, 12, L - t)
        qual = "".join(chr(int(v) + phred) for v in q)
        if i == 0 and phred == 64:
            qual = "h" * L   # (the first record decides the base: make it unmistakably phred 64)
        recs.append("@r%d some text%s%s%s+%s%s" % (i, eol, seq, eol, eol, qual))
    for i in range(long_reads):
        L = 302
        recs.append("@long%d%s%s%s+%s%s" % (i, eol, "ACGT" * 75 + "AC", eol, eol, "I" * L))
    text = eol.join(recs) + (eol if final_newline else "")
    if path.endswith(".gz"):
        with gzip.open(path, "wt", newline="") as fh:
            fh.write(text)
    else:
        with open(path, "w", newline="") as fh:
            fh.write(text)


def same_as_host_and_oracle(engine, path, adapter, **kw):
    dev = ingest.load_fastq_device(engine, path, adapter=adapter, **kw)
    host = ingest.load_fastq(path, adapter=adapter)
    want, total, phred = oingest.load_fastq(path, adapter=ingest.resolve_adapter(adapter))
    assert (dev["total"], dev["kept"], dev["phred"]) == (total, len(want), phred) == (host["total"], host["kept"], host["phred"])
    w = dev["words"].cpu().numpy().view(np.uint64)
    l = dev["lens"].cpu().numpy()
    nm = None if dev["nmask"] is None else dev["nmask"].cpu().numpy().view(np.uint64)
    assert pack.unpack_reads(w, l, nm) == [s.upper() for s in want]          # same reads, same order
    W = max(w.shape[0], host["words"].shape[0])

    def pad(a):
        return a if a.shape[0] == W else np.concatenate([a, np.zeros((W - a.shape[0], a.shape[1]), dtype=np.uint64)])
    assert np.array_equal(pad(w), pad(host["words"])) and np.array_equal(l, host["lens"])
    assert (nm is None) == (host["nmask"] is None) and (nm is None or np.array_equal(pad(nm), pad(host["nmask"])))
    assert dev["max_len"] == host["max_len"]
    return dev


def test_device_ingest_equals_host_and_oracle(engine, tmp_path):
    rng = np.random.default_rng(31)
    for name, kw, adapter in (("plain.fastq", {}, "none"), ("crlf.fastq", dict(crlf=True), "none"),
                              ("nofinal.fastq", dict(final_newline=False), "+3"), ("p64.fastq", dict(phred=64), "none"),
                              ("short.fastq", dict(max_len=30), "+2"), ("z.fastq.gz", {}, "none"),
                              # untrimmed 151- and 250-cycle runs: four and eight packed words per read
                              ("c151.fastq", dict(max_len=151, n=3000), "none"), ("c250.fastq", dict(max_len=250, n=3000), "+1")):
        p = str(tmp_path / name)
        make_fastq(p, rng, **kw)
        dev = same_as_host_and_oracle(engine, p, adapter)
        assert 500 < dev["kept"] < dev["total"]
        if "max_len" in kw:
            assert dev["words"].shape[0] == pack.words_for(kw["max_len"] - (1 if adapter == "+1" else 0)) and dev["max_len"] > kw["max_len"] - 12
    # adapter sequences (`-ad illumina`, a list of two): cutadapt's 3' search, one thread per read
    for name, adapter, ads in (("ill.fastq", "illumina", (ILLUMINA,)), ("two.fastq", "ACGTTGCAAGGCTTAC,TGGAATTCTCGG", ("ACGTTGCAAGGCTTAC", "TGGAATTCTCGG")),
                               ("ill.fastq.gz", "illumina", (ILLUMINA,))):
        p = str(tmp_path / name)
        make_fastq(p, rng, adapters=ads)
        dev = same_as_host_and_oracle(engine, p, adapter)
        assert 500 < dev["kept"] < dev["total"]
        untrimmed = ingest.load_fastq(p, adapter="none")
        assert untrimmed["kept"] > dev["kept"] + 500          # the adapters did cut reads below the minimum length
    # many small blocks: the tail behind the last record boundary is carried into the next block
    p = str(tmp_path / "plain.fastq")
    a = same_as_host_and_oracle(engine, p, "none", block_bytes=64 << 10, read_threads=3)
    b = ingest.load_fastq_device(engine, p, adapter="none")
    assert bool((a["words"] == b["words"]).all()) and bool((a["lens"] == b["lens"]).all())


def test_what_the_device_parser_refuses(engine, tmp_path):
    """An adapter of more than 64 bases, blank lines between records, a missing '+' line, reads beyond 255 nt:
    the device parser says so (the caller then takes the host parser)."""
    rng = np.random.default_rng(32)
    p = str(tmp_path / "ok.fastq")
    make_fastq(p, rng, n=200)
    with pytest.raises(ingest.DeviceIngestUnsupported):
        ingest.load_fastq_device(engine, p, adapter="ACGT" * 17)
    text = open(p).read()
    blank = str(tmp_path / "blank.fastq")
    recs = text.split("\n@r")
    open(blank, "w").write(recs[0] + "\n\n@r" + "\n@r".join(recs[1:]))
    with pytest.raises(ingest.DeviceIngestUnsupported):
        ingest.load_fastq_device(engine, blank)
    assert ingest.load_fastq(blank)["total"] == 200            # (the host parser accepts blank lines)
    noplus = str(tmp_path / "noplus.fastq")
    lines = text.split("\n")
    lines[4 * 7 + 2] = "x"
    open(noplus, "w").write("\n".join(lines))
    with pytest.raises(ingest.DeviceIngestUnsupported):
        ingest.load_fastq_device(engine, noplus)
    longp = str(tmp_path / "long.fastq")
    make_fastq(longp, rng, n=100, long_reads=2)
    with pytest.raises(ingest.DeviceIngestUnsupported):
        ingest.load_fastq_device(engine, longp)
    assert len(ingest.load_fastq(longp)["long_reads"]) == 2


def test_block_cut_on_the_host(native_lib):
    """mrg_fastq_block_cut: the offset of the last record header that can be told from a quality line
    starting with '@' (the line after next starts with '+')."""
    import ctypes as C
    rec = lambda i, q: "@r%d\nACGTACGTACGTACGTAC\n+\n%s\n" % (i, q)
    text = (rec(0, "I" * 18) + rec(1, "@" + "I" * 17) + rec(2, "I" * 18) + "@r3\nACGT").encode()
    cut = C.c_uint64(0)
    assert native_lib.mrg_fastq_block_cut(text, len(text), 0, C.byref(cut)) == 0
    assert text[cut.value:cut.value + 3] == b"@r2" or text[cut.value:cut.value + 3] == b"@r3"
    assert text[:cut.value].count(b"\n") % 4 == 0
    assert native_lib.mrg_fastq_block_cut(text, len(text), 1, C.byref(cut)) == 0 and cut.value == len(text)


@pytest.mark.parametrize("n,S", [(0, 1), (1, 1), (3, 2), (4, 1), (4099, 1), (20001, 3)])
def test_compact_wire_form_expands_to_the_arrays(engine, n, S):
    """mrg_expand_compact: reads grouped by length as a bit stream of 2 L bits each + one-byte counts with
    an escape list, widened on the device, must equal the arrays they were made from (every tail size,
    several samples, counts around the escape value, a single length and all of 1..32)."""
    import torch
    rng = np.random.default_rng(n + S)
    lens = rng.integers(1, 33, n).astype(np.uint8)
    if n == 4:
        lens[:] = 22
    words = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    words = np.where(lens < 32, words & ((np.uint64(1) << (2 * np.minimum(lens, 31).astype(np.uint64))) - np.uint64(1)), words)
    quant = rng.choice([0, 1, 2, 7, 254, 255, 256, 70000, 2**32 - 1], (n, S)).astype(np.uint32)
    c = pack.compact_read_set(words[None, :], lens, quant)
    order = np.arange(n) if c["order"] is None else c["order"]
    dev = engine.device
    bits = torch.from_numpy(c["bits"].view(np.int64)).to(dev)
    q8 = torch.from_numpy(c["quant8"]).to(dev)
    esc = torch.from_numpy(c["esc"].view(np.int32)).to(dev)
    rs = engine.expand_compact(bits, c["runs"], q8, esc, n_samples=S)
    torch.cuda.synchronize()
    assert rs.n == n and rs.W == 1
    assert np.array_equal(rs.words.cpu().numpy().view(np.uint64)[0], words[order])
    assert np.array_equal(rs.lens.cpu().numpy(), lens[order])
    assert np.array_equal(rs.quant.cpu().numpy().view(np.uint32), quant[order])
    if n:
        assert (rs.min_len, rs.max_len) == (int(lens.min()), int(lens.max()))
    # reads only
    rs2 = engine.expand_compact(bits, c["runs"])
    torch.cuda.synchronize()
    assert rs2.quant is None and np.array_equal(rs2.words.cpu().numpy().view(np.uint64)[0], words[order])


def test_compact_wire_form_refuses_what_it_cannot_hold(engine):
    import torch
    from mirge_amd._native import MirgeAmdError
    with pytest.raises(ValueError):
        pack.compact_read_set(np.zeros((1, 2), dtype=np.uint64), np.array([22, 33], dtype=np.uint8))
    bits = torch.zeros(7, dtype=torch.int64, device=engine.device)      # 8 x 22 nt = 352 bits = 6 words + padding
    engine.expand_compact(bits, np.array([[22, 8]], dtype=np.uint32))
    with pytest.raises(MirgeAmdError):
        engine.expand_compact(bits[:6], np.array([[22, 8]], dtype=np.uint32))      # no padding word
    with pytest.raises(MirgeAmdError):
        engine.expand_compact(bits, np.array([[33, 1]], dtype=np.uint32))          # one-word reads only


def test_cascade_over_a_compact_upload_equals_the_arrays(engine, native_lib, oracle_lib):
    """The whole host-resident path: a collapsed set of 16..32-nt reads -> pack.compact_read_set -> upload ->
    mrg_expand_compact -> packed cascade + tally, against the same reads sent as arrays (the compact form
    regroups the reads by length: `order` maps back)."""
    import torch
    from mirge_amd import synth
    from mirge_amd.engine import Engine, ReadSet
    from tests.util import LIB_ORDER, World
    w0 = World(scale=0.03, n_fixed=6000, n_var=3000, with_n=False, max_var_len=32)
    eng = Engine(0)
    for k in LIB_ORDER:
        eng.add_library(k, w0.index[k])
    n = w0.words.shape[1]
    quant = synth.synth_quant(n, n_samples=2)
    quant[::97, 0] = 1000 + np.arange(len(quant[::97]))        # some counts beyond one byte
    rs = ReadSet(w0.words, w0.lens, None, quant, device=eng.device)
    passes = eng.mirge_passes()
    want = eng.cascade_packed(rs, passes)
    want_stats = want.stats                      # (read before the next cascade: the counters live in the context)
    want_counts = eng.tally(rs, want, w0.n_mirna).cpu().numpy()
    c = pack.compact_read_set(w0.words, w0.lens, quant)
    assert c["order"] is not None and len(c["runs"]) > 10 and len(c["esc"]) > 0
    dev = eng.device
    rs2 = eng.expand_compact(torch.from_numpy(c["bits"].view(np.int64)).to(dev), c["runs"], torch.from_numpy(c["quant8"]).to(dev),
                             torch.from_numpy(c["esc"].view(np.int32)).to(dev), n_samples=2)
    got = eng.cascade_packed(rs2, passes)
    assert np.array_equal(got.packed.cpu().numpy(), want.packed.cpu().numpy()[c["order"]])
    assert np.array_equal(eng.tally(rs2, got, w0.n_mirna).cpu().numpy(), want_counts)
    for a, b in zip(got.stats, want_stats):
        assert (a["processed"], a["aligned"]) == (b["processed"], b["aligned"])
    eng.close()


@pytest.mark.parametrize("seed", range(10))
def test_device_ingest_on_random_adapter_sets(engine, tmp_path, seed):
    """Random adapter sets (one to four adapters of 6..48 bases, some of them prefixes of each other or low in
    complexity), read lengths, phred bases and line ends: the device parser's adapter search (cutadapt's semi-global
    alignment, one thread per read) = the host parser = oracle/ingest.py, array for array."""
    rng = np.random.default_rng(500 + seed)
    ads = []
    for _ in range(int(rng.integers(1, 5))):
        L = int(rng.integers(6, 49))
        kind = rng.random()
        if kind < 0.2 and ads:
            a = ads[0][:max(6, L // 2)]                                   # a prefix of another adapter
        elif kind < 0.35:
            a = ("ACGT"[int(rng.integers(0, 4))] * L)[:L]                 # a homopolymer
        elif kind < 0.5:
            a = ("AC" * L)[:L]                                            # a dinucleotide repeat
        else:
            a = "".join("ACGT"[c] for c in rng.integers(0, 4, L))
        ads.append(a)
    ads = list(dict.fromkeys(ads))
    p = str(tmp_path / ("r%d.fastq" % seed))
    make_fastq(p, rng, n=4000, max_len=int(rng.choice([36, 50, 76, 101])), phred=int(rng.choice([33, 33, 64])), crlf=bool(rng.random() < 0.3),
               final_newline=bool(rng.random() < 0.7), adapters=tuple(ads))
    dev = same_as_host_and_oracle(engine, p, ",".join(ads))
    assert 0 < dev["kept"] < dev["total"]
