"""The parallel inflate (csrc/pgzip.cpp, mrg_gz_open / mrg_gz_read) on the CPU: the bytes it returns are
the bytes zlib returns for the same `.fastq.gz` -- the reference reads gzip samples through one inflate
stream (parseArgument.py:32, __main__.py:289-314, trim_file.py:89-134) -- for every compression level,
for concatenated members, for stored blocks, with chunk starts that fall anywhere; a corrupt or truncated
file is an error, never other bytes.  MIRGE_AMD_GZ_CHUNK cuts these small files into many chunks."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest


def fastq_text(rng, n, qual_hi=74):
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(16, 52))
        s = acgt[rng.choice(5, L, p=[.25, .25, .25, .24, .01])].tobytes()
        q = bytes(rng.integers(33, qual_hi, L).astype(np.uint8))
        out.append(b"@M0:%d:%d 1:N:0:ACGT\n%s\n+\n%s\n" % (i, int(rng.integers(0, 1 << 20)), s, q))
    return b"".join(out)


def inflate(lib, path, threads, want_parallel=None, info=None):
    from mirge_amd._native import check
    h = C.c_void_p()
    check(lib.mrg_gz_open(path.encode(), threads, C.byref(h)))
    try:
        par, merged = C.c_int32(-1), C.c_uint64(0)
        check(lib.mrg_gz_info(h, C.byref(par), C.byref(merged)))
        if want_parallel is not None:
            assert par.value == want_parallel
        buf = C.create_string_buffer(1 << 20)
        out = []
        sizes = [1 << 20, 12345, 1, 700000]   # (odd request sizes: a chunk's bytes go out over several calls)
        i = 0
        while True:
            got = C.c_uint64(0)
            check(lib.mrg_gz_read(h, buf, sizes[i % len(sizes)], C.byref(got)))
            i += 1
            if not got.value:
                break
            out.append(buf.raw[:got.value])
        if info is not None:   # chunk starts that turned out not to be any (dropped, their chunk merged into the one in front)
            check(lib.mrg_gz_info(h, C.byref(par), C.byref(merged)))
            info["merged"] = int(merged.value)
        return b"".join(out)
    finally:
        lib.mrg_gz_close(h)


@pytest.fixture(scope="module")
def text():
    return fastq_text(np.random.default_rng(77), 60000)   # ~ 7 MB of text


@pytest.fixture()
def small_chunks(monkeypatch):
    monkeypatch.setenv("MIRGE_AMD_GZ_CHUNK", "65536")


@pytest.mark.parametrize("level", [1, 6, 9])
def test_parallel_inflate_equals_zlib(native_lib, tmp_path, text, small_chunks, level):
    p = str(tmp_path / "x.fastq.gz")
    with open(p, "wb") as fh:
        fh.write(gzip.compress(text, level))
    for threads in (2, 5):
        assert inflate(native_lib, p, threads, want_parallel=1) == text
    assert inflate(native_lib, p, 1, want_parallel=0) == text   # one thread: zlib's own reader


def test_concatenated_members_and_stored_blocks(native_lib, tmp_path, text, small_chunks):
    """Members of different levels back to back (what `cat a.gz b.gz` and bgzip produce), one of them
    stored (level 0: no Huffman block to find in it) and an empty one; chunk borders fall inside members,
    on member borders and inside the stored data."""
    a, b, c = text[:2_000_000], text[2_000_000:2_700_000], text[2_700_000:]
    p = str(tmp_path / "m.fastq.gz")
    with open(p, "wb") as fh:
        fh.write(gzip.compress(a, 6) + gzip.compress(b"", 6) + gzip.compress(b, 0) + gzip.compress(c, 2))
    assert inflate(native_lib, p, 4, want_parallel=1) == text
    # bgzf-like: many small members
    p2 = str(tmp_path / "b.fastq.gz")
    with open(p2, "wb") as fh:
        for o in range(0, len(text), 60000):
            fh.write(gzip.compress(text[o:o + 60000], 5))
    info = {}
    assert inflate(native_lib, p2, 4, want_parallel=1, info=info) == text
    # every member is ONE final block: there is no non-final dynamic block to find in such a file, the members themselves
    # are the chunk starts (round 4 dropped every chunk and inflated the whole file behind chunk 0: 5 x slower than
    # zlib, the sample held in memory three times over)
    n_chunks = os.path.getsize(p2) // 65536
    assert n_chunks > 15 and info["merged"] <= 2, info
    # real BGZF members: the extra field "BC" + block size, an empty EOF member at the end
    p3 = str(tmp_path / "bgzf.fastq.gz")
    with open(p3, "wb") as fh:
        for o in list(range(0, len(text), 0xff00)) + [len(text)]:
            raw = text[o:o + 0xff00]
            z = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = z.compress(raw) + z.flush()
            bsize = 18 + len(body) + 8 - 1
            fh.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + b"\x06\x00" + b"BC\x02\x00" + bsize.to_bytes(2, "little") + body +
                     (zlib.crc32(raw) & 0xffffffff).to_bytes(4, "little") + len(raw).to_bytes(4, "little"))
    assert gzip.decompress(open(p3, "rb").read()) == text
    info = {}
    assert inflate(native_lib, p3, 6, want_parallel=1, info=info) == text and info["merged"] <= 2, info


def test_plain_and_tiny_files_take_zlib(native_lib, tmp_path, text, small_chunks):
    p = str(tmp_path / "plain.fastq")
    with open(p, "wb") as fh:
        fh.write(text[:300000])
    assert inflate(native_lib, p, 4, want_parallel=0) == text[:300000]
    p2 = str(tmp_path / "tiny.fastq.gz")
    with open(p2, "wb") as fh:
        fh.write(gzip.compress(text[:5000]))
    assert inflate(native_lib, p2, 4, want_parallel=0) == text[:5000]


def test_corrupt_and_truncated_files_are_errors(native_lib, tmp_path, text, small_chunks):
    from mirge_amd._native import MirgeAmdError
    comp = bytearray(gzip.compress(text, 6))
    good = bytes(comp)
    # a flipped bit in the middle of the deflate data: a bad code, a broken chain of chunks or the CRC says so
    for at in (len(comp) // 2, len(comp) // 3 + 7, len(comp) - 5000):
        bad = bytearray(good)
        bad[at] ^= 0x10
        p = str(tmp_path / "bad.fastq.gz")
        with open(p, "wb") as fh:
            fh.write(bytes(bad))
        try:
            got = inflate(native_lib, p, 4)
        except MirgeAmdError:
            continue
        assert got == text   # (never other bytes without an error; zlib.decompress agrees the flip changed something)
        with pytest.raises(zlib.error):
            zlib.decompress(bytes(bad), 31)
    p = str(tmp_path / "trunc.fastq.gz")
    with open(p, "wb") as fh:
        fh.write(good[:len(good) * 2 // 3])
    with pytest.raises(MirgeAmdError):
        inflate(native_lib, p, 4)
    # a wrong CRC in the trailer
    bad = bytearray(good)
    bad[-6] ^= 0xFF
    with open(p, "wb") as fh:
        fh.write(bytes(bad))
    with pytest.raises(MirgeAmdError):
        inflate(native_lib, p, 4)


def test_fastq_loader_reads_gzip_through_the_parallel_reader(native_lib, tmp_path, small_chunks):
    """mrg_fastq_load on a `.fastq.gz` cut into many chunks = the same text as a plain file."""
    from mirge_amd import ingest
    rng = np.random.default_rng(5)
    text = fastq_text(rng, 30000, qual_hi=70)
    plain, gz = str(tmp_path / "s.fastq"), str(tmp_path / "s.fastq.gz")
    with open(plain, "wb") as fh:
        fh.write(text)
    with open(gz, "wb") as fh:
        fh.write(gzip.compress(text, 6))
    a = ingest.load_fastq(plain, adapter="none", threads=4)
    b = ingest.load_fastq(gz, adapter="none", threads=4)
    for k in ("total", "kept", "phred", "max_len"):
        assert a[k] == b[k], k
    assert np.array_equal(a["words"], b["words"]) and np.array_equal(a["lens"], b["lens"])
    assert (a["nmask"] is None) == (b["nmask"] is None) and (a["nmask"] is None or np.array_equal(a["nmask"], b["nmask"]))


@pytest.mark.parametrize("seed", range(8))
def test_parallel_inflate_on_random_streams(native_lib, tmp_path, monkeypatch, seed):
    """Deflate streams as other writers make them: every zlib strategy (fixed-Huffman blocks only, Huffman without
    matches, run-length, filtered), small memLevels (many short blocks), full flushes in odd places (empty stored
    blocks, byte-aligned restarts), several members, text of three kinds (FASTQ, long repeats, random bytes), chunk
    sizes that put chunk starts anywhere: the parallel reader returns zlib's bytes."""
    rng = np.random.default_rng(900 + seed)
    chunk = int(rng.choice([4096, 20000, 65536]))
    monkeypatch.setenv("MIRGE_AMD_GZ_CHUNK", str(chunk))
    parts = []
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            parts.append(fastq_text(rng, int(rng.integers(3000, 20000))))
        elif kind == 1:
            unit = bytes(rng.integers(65, 91, int(rng.integers(1, 300))).astype(np.uint8))
            parts.append(unit * int(rng.integers(2000, 20000)))                     # matches at every distance up to the window
        else:
            parts.append(bytes(rng.integers(0, 256, int(rng.integers(100000, 900000))).astype(np.uint8)))   # stored blocks mostly
    blob, text = [], b"".join(parts)
    for part in parts:     # one member per part
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, 31, int(rng.integers(1, 10)),
                              int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])))
        at = 0
        while at < len(part):
            step = int(rng.integers(1, 400000))
            blob.append(co.compress(part[at:at + step]))
            at += step
            if rng.random() < 0.3:
                blob.append(co.flush(zlib.Z_FULL_FLUSH if rng.random() < 0.5 else zlib.Z_SYNC_FLUSH))
        blob.append(co.flush())
    p = str(tmp_path / "r.gz")
    with open(p, "wb") as fh:
        fh.write(b"".join(blob))
    assert gzip.open(p).read() == text
    big = os.path.getsize(p) >= 4 * chunk     # (a file of less than two chunks is zlib's)
    for threads in (3, 8):
        assert inflate(native_lib, p, threads, want_parallel=1 if big else None) == text
