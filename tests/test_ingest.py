"""FASTQ ingest (host, CPU test) and GPU collapse against the oracle and the vectors
captured from the reference's quantReads."""
import gzip
import json
import os

import numpy as np
import pytest

from mirge_amd import ingest, pack
from oracle import cascade
from oracle import ingest as oingest
from tests.conftest import ROOT


def write_fastq(path, rng, n, phred_base=33, gz=False, first_hi=False):
    opener = gzip.open if gz else open
    with opener(path, "wt") as fh:
        for i in range(n):
            L = int(rng.integers(10, 60))
            seq = "".join("ACGTN"[int(c)] for c in rng.choice(5, L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
            q = rng.integers(20, 41, L)
            cut = int(rng.integers(0, L + 1))
            q[cut:] = rng.integers(0, 14, L - cut)       # a low-quality tail
            if rng.random() < 0.2:
                q[int(rng.integers(0, L))] = 2             # an isolated bad base
            if first_hi and i == 0:
                q[:] = 45                                    # chr > 74 at base 33 -> "phred64" sniffed
            fh.write("@r%d extra\n%s\n+\n%s\n" % (i, seq, "".join(chr(int(x) + phred_base) for x in q)))


@pytest.mark.parametrize("gz,first_hi", [(False, False), (True, False), (False, True)])
def test_fastq_loader_matches_oracle(native_lib, tmp_path, gz, first_hi):
    rng = np.random.default_rng(3)
    p = str(tmp_path / ("x.fastq.gz" if gz else "x.fastq"))
    write_fastq(p, rng, 3000, gz=gz, first_hi=first_hi)
    want, total, phred = oingest.load_fastq(p)
    got = ingest.load_fastq(p)
    assert (got["total"], got["kept"], got["phred"]) == (total, len(want), phred)
    assert pack.unpack_reads(got["words"], got["lens"], got["nmask"]) == want
    assert 0 < len(want) < total


def test_fastq_loader_reads_of_every_packed_width(native_lib, tmp_path):
    """Untrimmed long-cycle runs (`-ad none`): reads of up to 255 nt are packed (1, 2, 4 or 8 words, the smallest
    that holds the file's longest read), longer ones are kept aside in file order and counted among the kept."""
    rng = np.random.default_rng(5)
    for cycles, W in ((32, 1), (33, 2), (128, 4), (129, 8), (255, 8), (300, 8)):
        p = str(tmp_path / ("c%d.fastq" % cycles))
        with open(p, "w") as fh:
            for i in range(400):
                L = cycles if i % 50 == 0 else int(rng.integers(16, cycles + 1))
                seq = "".join("ACGTN"[int(c)] for c in rng.choice(5, L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
                fh.write("@r%d\n%s\n+\n%s\n" % (i, seq, "I" * L))
        want, total, phred = oingest.load_fastq(p)
        got = ingest.load_fastq(p)
        assert got["words"].shape[0] == W and (got["total"], got["kept"]) == (400, len(want)) == (400, 400)
        assert pack.unpack_reads(got["words"], got["lens"], got["nmask"]) == [s for s in want if len(s) <= 255]
        assert got["long_reads"] == [s for s in want if len(s) > 255] and (len(got["long_reads"]) > 0) == (cycles > 255)
        assert got["max_len"] == min(cycles, 255) or cycles > 255
        w2, l2, n2 = pack.pack_reads([s for s in want if len(s) <= 255])
        assert np.array_equal(w2, got["words"]) and np.array_equal(l2, got["lens"])


@pytest.mark.parametrize("crlf", [False, True])
def test_fastq_loader_block_boundaries(native_lib, tmp_path, crlf):
    """A file of several reader blocks (the reader cuts the text at the last record header of each
    ~4 MB block; the workers split the blocks into records): quality lines that start with '@' or
    '+', blank lines between records, a last record without a newline, CRLF line ends, a phred-64
    looking record among the first 1000 but not the first -- every record arrives, in file order."""
    rng = np.random.default_rng(17)
    n = 120_000
    p = str(tmp_path / "blocks.fastq")
    nl = "\r\n" if crlf else "\n"
    with open(p, "w", newline="") as fh:
        for i in range(n):
            L = int(rng.integers(16, 45))
            seq = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, L))
            q = rng.integers(25, 41, L)
            qs = "".join(chr(int(x) + 33) for x in q)
            if i % 7 == 0:
                qs = "@" + qs[1:]          # '@' = quality 31
            if i % 11 == 0:
                qs = "+" + qs[1:]          # '+' = quality 10
            if i == 500:
                qs = "h" * L               # > 'J': the file is sniffed as phred 64, the workers trim with 33
            rec = "@read%d some text%s%s%s+%s%s" % (i, nl, seq, nl, nl, qs)
            fh.write(rec + (nl if i + 1 < n else ""))
    assert os.path.getsize(p) > 9_000_000
    want, total, phred = oingest.load_fastq(p)
    for threads in (1, 5):
        got = ingest.load_fastq(p, threads=threads)
        assert (got["total"], got["kept"], got["phred"]) == (total, len(want), phred) == (n, len(want), 64)
        assert pack.unpack_reads(got["words"], got["lens"], got["nmask"]) == want
    # blank lines between records (the oracle reads four lines at a time; the product skips them)
    blank = str(tmp_path / "blank.fastq")
    with open(p, newline="") as src, open(blank, "w", newline="") as dst:
        recs = src.read().split(nl + "@read")
        dst.write((nl + nl + "@read").join(recs[:2000]) + nl + ("@read" + (nl + "@read").join(recs[2000:])))
    gotb = ingest.load_fastq(blank, threads=3)
    assert (gotb["total"], gotb["kept"]) == (n, len(want))
    assert pack.unpack_reads(gotb["words"], gotb["lens"], gotb["nmask"]) == want
    # a broken record deep in the file is reported with its number in the file
    from mirge_amd._native import MirgeAmdError
    bad = str(tmp_path / "bad_deep.fastq")
    with open(p) as src, open(bad, "w", newline="") as dst:
        text = src.read()
        k = text.index("@read100000 ")
        dst.write(text[:k] + "read100000 broken" + text[k + len("@read100000 some text"):])
    with pytest.raises(MirgeAmdError, match="record 100001"):
        ingest.load_fastq(bad)


def test_quality_trim_rule_known_answers():
    q = lambda s: "".join(chr(x + 33) for x in s)
    t = oingest.quality_trim_3p
    assert t(q([30] * 20)) == 20                    # nothing to trim
    assert t(q([30] * 15 + [2] * 5)) == 15          # low tail goes
    assert t(q([30] * 10 + [5, 30, 5, 5])) == 12    # the walk stops at the first good base
    assert t(q([30] * 10 + [5, 12, 5, 5])) == 10    # a mediocre base does not rescue the tail
    assert t(q([30] * 10 + [5, 30, 30, 30, 5])) == 14
    assert t(q([2] * 8)) == 0
    assert t("") == 0


ILLUMINA = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"       # MAIN:125


def test_adapter_rule_known_answers(native_lib):
    """cutadapt's documented 3' adapter behaviour (min overlap 3, 12 % errors, indels allowed),
    same answers from the product (C++) and the oracle restatement."""
    ins = "ACGTTGCAAGCTTGACCTGA"                      # 20 nt without adapter-like content
    cases = [
        (ins + ILLUMINA + "ACGT", 20),                 # whole adapter inside the read
        (ins + ILLUMINA[:8], 20),                      # an 8-nt adapter prefix at the 3' end
        (ins + ILLUMINA[:3], 20),                      # 3 nt: the minimum overlap
        (ins + ILLUMINA[:2], None),                    # 2 nt: left alone
        (ins + ILLUMINA[:10] + "A" + ILLUMINA[11:], 20),          # one substitution in 29 nt
        (ins + ILLUMINA[:12] + ILLUMINA[13:] + "CC", 20),         # one base deleted from the adapter
        (ins + ILLUMINA[:12] + "A" + ILLUMINA[12:], 20),          # one base inserted
        (ins + ILLUMINA[:4] + "C" + ILLUMINA[5:10], 20),          # 1 error in a 10-nt overlap: 1 <= 1.2
        (ins + ILLUMINA[:4] + "C" + ILLUMINA[5:8], None),         # 1 error in 8 nt: 1 > 0.96
        (ins, None),
        ("", None),
        (ILLUMINA, 0),
        (ins + ILLUMINA + ins + ILLUMINA, 20),         # leftmost exact occurrence
    ]
    for read, cut in cases:
        got = ingest.adapter_locate(ILLUMINA, read)
        want = oingest.locate_adapter_3p(ILLUMINA, read)
        assert got == want, read
        assert (None if got is None else got[0]) == cut, read
    assert ingest.adapter_locate("11", ins + "ACGT") is None        # `-ad ion` (MAIN:127) never matches
    assert oingest.trim_read("ACGTACGTACGT", "+3") == "TACGTACGT"


def test_adapter_search_matches_oracle_on_random_reads(native_lib):
    rng = np.random.default_rng(12)
    adapters = [ILLUMINA, "AGATCGGAAGAGC", "ACGT", "TTTTTTTTTT"]
    hits = 0
    for it in range(4000):
        ad = adapters[it % len(adapters)]
        L = int(rng.integers(0, 60))
        read = "".join("ACGTN"[int(c)] for c in rng.choice(5, L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        if rng.random() < 0.7:                          # plant a (possibly damaged, possibly cut) adapter
            a = list(ad[:int(rng.integers(1, len(ad) + 1))] if rng.random() < 0.5 else ad)
            for _ in range(int(rng.integers(0, 4))):
                op, k = int(rng.integers(0, 3)), int(rng.integers(0, len(a))) if a else 0
                if not a:
                    break
                if op == 0:
                    a[k] = "ACGT"[int(rng.integers(0, 4))]
                elif op == 1:
                    del a[k]
                else:
                    a.insert(k, "ACGT"[int(rng.integers(0, 4))])
            tail = "" if rng.random() < 0.5 else "".join("ACGT"[int(c)] for c in rng.integers(0, 4, int(rng.integers(0, 12))))
            read = read + "".join(a) + tail
        got = ingest.adapter_locate(ad, read)
        assert got == oingest.locate_adapter_3p(ad, read), (ad, read)
        hits += got is not None
    assert 1000 < hits < 3900


@pytest.mark.parametrize("adapter", ["illumina", "ion", "+4", "AGATCGGAAGAGC,TGGAATTCTCGGGTGCC"])
def test_fastq_loader_with_adapter_matches_oracle(native_lib, tmp_path, adapter):
    rng = np.random.default_rng(5)
    p = str(tmp_path / "a.fastq")
    with open(p, "w") as fh:
        for i in range(2000):
            L = int(rng.integers(14, 30))
            seq = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, L))
            r = rng.random()
            if r < 0.6:
                seq += ILLUMINA[:int(rng.integers(1, 30))]
            elif r < 0.8:
                seq += "AGATCGGAAGAGC" + "ACGT"[int(rng.integers(0, 4))] * int(rng.integers(0, 5))
            q = rng.integers(25, 41, len(seq))
            if rng.random() < 0.3:
                q[-int(rng.integers(1, 6)):] = 3
            fh.write("@r%d\n%s\n+\n%s\n" % (i, seq, "".join(chr(int(x) + 33) for x in q)))
    resolved = ingest.resolve_adapter(adapter)
    want, total, phred = oingest.load_fastq(p, adapter=resolved)
    got = ingest.load_fastq(p, adapter=adapter)
    assert (got["total"], got["kept"], got["phred"]) == (total, len(want), phred)
    assert pack.unpack_reads(got["words"], got["lens"], got["nmask"]) == want
    plain = ingest.load_fastq(p)
    if adapter == "ion":
        assert got["kept"] == plain["kept"] and (got["lens"] == plain["lens"]).all()
    else:
        assert got["lens"].sum() < plain["lens"].sum()


def test_fastq_errors(native_lib, tmp_path):
    from mirge_amd._native import MirgeAmdError
    bad = tmp_path / "bad.fastq"
    bad.write_text("@r0\nACGT\n+\nII\n")
    with pytest.raises(MirgeAmdError):
        ingest.load_fastq(str(bad))
    with pytest.raises(MirgeAmdError):
        ingest.load_fastq(str(tmp_path / "missing.fastq"))


@pytest.mark.parametrize("gz", [False, True])
def test_one_file_read_in_parts(native_lib, tmp_path, gz, monkeypatch):
    """mrg_fastq_load_part (`--gpus N` on one sample): a plain file cut into byte ranges at record starts -- quality
    lines that start with '@', CRLF, blocks of several MB --, a gzip file shared out block by block: whatever the number
    of parts, their reads together are the file's reads (in file order for a plain file), their counts add up, the
    quality base is the file's (a phred-64 file whose first record alone says so)."""
    monkeypatch.setenv("MIRGE_AMD_GZ_CHUNK", "65536")
    rng = np.random.default_rng(9)
    p = str(tmp_path / ("one.fastq" + (".gz" if gz else "")))
    for first_hi, n_parts_list in ((False, (2, 3, 7)), (True, (3,))):
      write_fastq(p, rng, 60000, gz=gz, first_hi=first_hi)
      whole = ingest.load_fastq(p, threads=3)
      want = pack.unpack_reads(whole["words"], whole["lens"], whole["nmask"])
      assert whole["total"] == 60000 and whole["phred"] == (64 if first_hi else 33) and (len(want) > 30000 or first_hi)
      for n_parts in n_parts_list:
        got, total, kept = [], 0, 0
        for part in range(n_parts):
            fq = ingest.load_fastq(p, threads=2, part=part, n_parts=n_parts)
            got.append(pack.unpack_reads(fq["words"], fq["lens"], fq["nmask"]) if fq["packed"] else [])
            total += fq["total"]
            kept += fq["kept"]
            assert fq["phred"] == (whole["phred"] if part == 0 else 0)
            assert fq["total"] > 0 or gz   # (a gzip file is shared out in blocks of ~4 MB of text: this one has two)
        assert total == whole["total"] and kept == whole["kept"]
        flat = [r for g in got for r in g]
        if gz:
            assert sorted(flat) == sorted(want)
        else:
            assert flat == want


@pytest.mark.gpu
def test_gpu_collapse_matches_quantReads_golden(native_lib):
    from mirge_amd.engine import Engine
    with open(os.path.join(ROOT, "tests", "golden", "cascade_small.json")) as fh:
        g = json.load(fh)
    reads, sample = [], []
    for si, rs in enumerate(g["samples"]):
        reads += rs
        sample += [si] * len(rs)
    order = np.random.default_rng(1).permutation(len(reads))
    reads = [reads[i] for i in order]
    sample = np.array(sample, dtype=np.uint16)[order]
    words, lens, nmask = pack.pack_reads(reads)
    eng = Engine(0)
    out = ingest.collapse(eng, words, lens, nmask, sample, n_samples=2)
    uniq = pack.unpack_reads(out["words"], out["lens"], out["nmask"])
    exp = g["expected"]["seqDic"]
    assert len(uniq) == len(set(uniq)) == len(exp)
    for s, q in zip(uniq, out["quant"]):
        assert [int(x) for x in q] == exp[s]["quant"], s
    assert {str(k): v for k, v in out["length_hist"].items()} == g["expected"]["readLengthDic"]
    # deterministic order: by length, then packed bases
    keys = [(int(l), [int(w) for w in out["words"][::-1, i]]) for i, l in enumerate(out["lens"])]
    assert all(keys[i][0] <= keys[i + 1][0] for i in range(len(keys) - 1))


@pytest.mark.gpu
def test_gpu_collapse_fused_and_general_paths(native_lib):
    from mirge_amd import synth
    from mirge_amd.engine import Engine
    eng = Engine(0)
    libs = synth.SynthLibraries(scale=0.02)
    w = synth.synth_reads_packed(libs, 300000, zipf_s=1.3)
    lens = np.full(w.shape[0], 22, dtype=np.uint8)
    for n_samples in (1, 3):
        sample = (np.arange(w.shape[0]) % n_samples).astype(np.uint16)
        # a dict keyed by the read, one increment per record, as quantReads.py:9-16 does
        want = {}
        for key, s in zip(w.tolist(), sample.tolist()):
            want.setdefault(key, [0] * n_samples)[s] += 1
        for fast in (1, 0):   # the duplication-aware path; the general column-by-column sort
            eng.set_option("collapse_fast", fast)
            out = ingest.collapse(eng, w[None, :], lens, None, sample, n_samples=n_samples, max_len=22)
            assert out["words"].shape[1] == len(want)
            for key, q in zip(out["words"][0].tolist(), out["quant"].tolist()):
                assert q == want[key]
            assert out["length_hist"] == {22: [int((sample == s).sum()) for s in range(n_samples)]}
    eng.set_option("collapse_fast", 1)
    # empty input
    out = ingest.collapse(eng, np.zeros((1, 0), np.uint64), np.zeros(0, np.uint8))
    assert out["words"].shape == (1, 0) and out["quant"].shape[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("max_read,W", [(32, 1), (60, 2), (128, 4), (255, 8)])
def test_gpu_collapse_of_reads_of_every_packed_width(native_lib, max_read, W):
    """quantReads.py:3-24 on the device for one to eight packed words per read, reads with N, three samples: the
    unique reads, their per-sample counts and the length histogram of a dict keyed by the read string."""
    from mirge_amd.engine import Engine
    rng = np.random.default_rng(max_read)
    pool = []
    for _ in range(4000):
        L = int(rng.integers(16, max_read + 1))
        pool.append("".join("ACGTN"[c] for c in rng.choice(5, L, p=[.2475, .2475, .2475, .2475, .01])))
    pool += [pool[0][:-1], pool[0] + "A" if len(pool[0]) < max_read else pool[1][:-2], "A" * max_read, "A" * (max_read - 1), "N" * 16]
    idx = rng.zipf(1.3, 60000) % len(pool)
    reads = [pool[i] for i in idx]
    sample = rng.integers(0, 3, len(reads)).astype(np.uint16)
    words, lens, nmask = pack.pack_reads(reads)
    assert words.shape[0] == W
    want, hist = {}, {}
    for r, s in zip(reads, sample.tolist()):
        want.setdefault(r, [0, 0, 0])[s] += 1
        hist.setdefault(len(r), [0, 0, 0])[s] += 1
    eng = Engine(0)
    out = ingest.collapse(eng, words, lens, nmask, sample, n_samples=3, max_len=int(lens.max()))
    got = pack.unpack_reads(np.ascontiguousarray(out["words"]), out["lens"], out["nmask"])
    assert len(got) == len(want) == len(set(got))
    for r, q in zip(got, out["quant"].tolist()):
        assert q == want[r], r
    assert out["length_hist"] == hist


def test_compact_read_set_groups_by_length_and_escapes_large_counts():
    """pack.compact_read_set (the host side of mrg_expand_compact): a bit stream of 2 L bits per read in
    length groups, one-byte counts, escapes for counts >= 255 -- and back, bit by bit."""
    from mirge_amd import pack
    rng = np.random.default_rng(3)
    n = 1000
    lens = rng.integers(1, 33, n).astype(np.uint8)
    words = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    words = np.where(lens < 32, words & ((np.uint64(1) << (2 * np.minimum(lens, 31).astype(np.uint64))) - np.uint64(1)), words)
    quant = rng.choice([0, 1, 5, 254, 255, 300, 10**6], (n, 2)).astype(np.uint32)
    c = pack.compact_read_set(words[None, :], lens, quant)
    order = c["order"]
    assert sorted(order.tolist()) == list(range(n)) and np.all(np.diff(lens[order].astype(int)) >= 0)
    assert c["runs"][:, 1].sum() == n and list(c["runs"][:, 0]) == sorted(set(lens.tolist()))
    # decode the stream with Python integers
    stream = [int(x) for x in c["bits"]]
    base, i = 0, 0
    for L, count in c["runs"].tolist():
        big = sum(x << (64 * k) for k, x in enumerate(stream[base:base + (count * 2 * L + 63) // 64 + 1]))
        for j in range(count):
            assert (big >> (2 * L * j)) & ((1 << (2 * L)) - 1) == int(words[order][i]), (L, j)
            i += 1
        base += (count * 2 * L + 63) // 64
    assert base + 1 == len(stream)
    q = c["quant8"].astype(np.uint32).reshape(-1)
    assert np.all((q == 255) == (quant[order].reshape(-1) >= 255))
    q[c["esc"][:, 0]] = c["esc"][:, 1]
    assert np.array_equal(q.reshape(n, 2), quant[order])
    same = pack.compact_read_set(words[None, :] & np.uint64((1 << 44) - 1), np.full(n, 22, dtype=np.uint8))
    assert same["order"] is None and same["runs"].tolist() == [[22, n]] and same["quant8"] is None
    assert same["bits"].shape[0] == (n * 44 + 63) // 64 + 1
    with pytest.raises(ValueError):
        pack.compact_read_set(words[None, :2], np.array([22, 33], dtype=np.uint8))
