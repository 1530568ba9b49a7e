"""FASTQ ingest and collapse.

  load_fastq   trim_file (utils/trim_file.py:89-134): 3' quality trimming at Q10 (cutadapt's /
               BWA's rule), `-ad` adapter removal (cutadapt's 3' adapter search at error rate
               0.12, or `+N`), 16-nt minimum, phred sniffing (:104-106)
  collapse     quantReads (utils/quantReads.py:3-24): unique reads, per-sample counts and the
               read-length histogram, on the GPU (hipCUB radix sort + scans)

cutadapt is absent from the image: both rules are restated from its published algorithms
(parity unpinned, see csrc/fastq.cpp).
"""
import ctypes as C
import os

import numpy as np

from . import _native
from ._native import check

QUAL_CUTOFF = 10  # trim_file.py:30
MIN_LENGTH = 16   # trim_file.py:33


def resolve_adapter(adapter):
    """MAIN:123-127: the `-ad` keywords."""
    if adapter == "illumina":
        return "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
    if adapter == "ion":
        return "11"     # the reference passes this on as an adapter *sequence*; it never matches
    return adapter


def adapter_locate(adapter, read, max_error_rate=0.12, min_overlap=3):
    """cutadapt's 3' adapter search on one read: None or (read_start, read_stop, adapter_stop,
    matches, errors); the read is cut at read_start."""
    out = (C.c_int32 * 6)()
    check(_native.load().mrg_adapter_locate(adapter.upper().encode(), read.upper().encode(), max_error_rate,
                                            min(min_overlap, len(adapter)), out))
    return tuple(out[1:6]) if out[0] else None


def load_fastq(path, words_per_read=None, qual_cutoff=QUAL_CUTOFF, min_len=MIN_LENGTH, adapter="none",
               threads=0, part=0, n_parts=1):
    """Returns dict(words [W, n], lens, nmask|None, total, kept, packed, phred, max_len, long_reads):
    `kept` = reads that survive trimming (the reference's trimmedReads), `packed` = n of them in the
    arrays, `long_reads` = the kept reads beyond 255 nt (ASCII; too long for eight packed words and a length byte).
    part / n_parts: this reader's share of a file several processes ingest (mrg_fastq_load_part: a byte range of a
    plain file, every n_parts-th block of a gzip file); the counts are then those of the share."""
    lib = _native.load()
    h = C.c_void_p()
    if n_parts > 1:
        check(lib.mrg_fastq_load_part(os.fsencode(path), qual_cutoff, min_len, resolve_adapter(adapter).encode(),
                                      int(threads), int(part), int(n_parts), C.byref(h)))
    else:
        check(lib.mrg_fastq_load(os.fsencode(path), qual_cutoff, min_len, resolve_adapter(adapter).encode(),
                                 int(threads), C.byref(h)))
    try:
        info = _native.FastqInfo()
        check(lib.mrg_fastq_get_info(h, C.byref(info)))
        W = max(int(info.words_per_read), int(words_per_read or 1))
        n = int(info.n_kept)
        words = np.zeros((W, n), dtype=np.uint64)
        lens = np.zeros(n, dtype=np.uint8)
        nmask = np.zeros((W, n), dtype=np.uint64) if info.has_n else None
        check(lib.mrg_fastq_copy(h, W, words.ctypes.data, lens.ctypes.data,
                                 None if nmask is None else nmask.ctypes.data))
        long_reads = []
        ptr = C.c_char_p()
        for i in range(int(info.n_long)):
            check(lib.mrg_fastq_long_read(h, i, C.byref(ptr)))
            long_reads.append(ptr.value.decode("ascii"))
        return dict(words=words, lens=lens, nmask=nmask, total=int(info.n_total), kept=n + len(long_reads),
                    packed=n, phred=int(info.phred), max_len=int(info.max_len), long_reads=long_reads)
    finally:
        lib.mrg_fastq_free(h)


class DeviceIngestUnsupported(Exception):
    """The device parser does not take this input (adapter sequence, ill-formed or blank-line-separated
    records, reads beyond 255 nt): the caller uses load_fastq, which does and which words the errors."""


def load_fastq_device(engine, path, adapter="none", qual_cutoff=QUAL_CUTOFF, min_len=MIN_LENGTH, block_bytes=256 << 20,
                      read_threads=8, inflate_threads=None):
    """load_fastq with the record splitting, trimming and packing ON THE DEVICE (mrg_fastq_parse_device):
    the host reads the file into a pinned buffer (plain text: `read_threads` parallel preads; gzip: the
    parallel inflate of mrg_gz_open with `inflate_threads` workers, default half the cores, at most 32: its reader saturates there), cuts it at record boundaries (mrg_fastq_block_cut) and uploads text blocks.  `-ad none`,
    `-ad +N` and adapter sequences (`-ad illumina`: cutadapt's 3' search, one thread per read).  Returns dict(words int64 [W, n], lens uint8 [n], nmask int64 [W, n] | None -- DEVICE
    tensors, reads in file order --, total, kept, packed, phred, max_len, long_reads=[])."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    ad = resolve_adapter(adapter)
    cut, ad_seqs = 0, None
    if ad in (None, "", "none"):
        pass
    elif ad.startswith("+"):
        cut = int(ad)
    else:
        # adapter sequences: cutadapt's 3' search, one thread per read (mrg_fastq_parse_device_ad)
        seqs = [a for a in ad.upper().split(",") if a]
        if len(seqs) > 4 or any(len(a) > 64 for a in seqs):
            raise DeviceIngestUnsupported("more than 4 adapters, or an adapter of more than 64 bases")
        ad_seqs = ",".join(seqs).encode()
    lib = engine._lib
    dev = engine.device
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    is_gz = path.endswith(".gz")
    # two pinned buffers, used in turn: the tail behind a block's last record boundary is copied to the head
    # of the other one.  (Reading block i + 1 in a background thread while the device parses block i was
    # measured SLOWER: 130-230 M reads/s against 230-266 -- the parallel preads and the upload compete.)
    pins = [torch.empty(block_bytes + (1 << 20), dtype=torch.uint8).pin_memory() for _ in range(2)]
    pin_nps = [t.numpy() for t in pins]
    pin_mvs = [memoryview(a) for a in pin_nps]
    d_text = torch.empty(block_bytes + (1 << 20), dtype=torch.uint8, device=dev)
    outs, total, kept, max_len, any64, base = [], 0, 0, 0, False, None
    # gzip samples: the parallel inflate of the C-ABI (mrg_gz_open, csrc/pgzip.cpp) writes straight into the pinned buffer
    gz = C.c_void_p()
    if is_gz:
        check(lib.mrg_gz_open(path.encode(), max(1, int(inflate_threads or min(32, max(2, (os.cpu_count() or 2) // 2)))), C.byref(gz)))
    fh = None if is_gz else open(path, "rb", buffering=0)
    pool = None if is_gz else ThreadPoolExecutor(max_workers=max(1, int(read_threads)))
    state = dict(offset=0, eof=False)
    size = None if is_gz else os.fstat(fh.fileno()).st_size

    def fill(which, have):
        """Read up to block_bytes - have bytes behind the `have` carried-over bytes of buffer `which`;
        returns the bytes now in the buffer."""
        mv = pin_mvs[which]
        room = block_bytes - have
        got = 0
        if is_gz:
            k = C.c_uint64(0)
            check(lib.mrg_gz_read(gz, C.c_void_p(pins[which].data_ptr() + have), room, C.byref(k)))
            got = int(k.value)
            if got < room:
                state["eof"] = True
        else:
            got = max(0, min(room, size - state["offset"]))
            if got:
                n_parts = max(1, min(int(read_threads), got >> 22))
                step = (got + n_parts - 1) // n_parts
                fd, off0 = fh.fileno(), state["offset"]

                def part(i):
                    a, b = i * step, min(got, (i + 1) * step)
                    done = a
                    while done < b:
                        k = os.preadv(fd, [mv[have + done:have + b]], off0 + done)
                        if k <= 0:
                            raise IOError("short read from %s" % path)
                        done += k
                list(pool.map(part, range(n_parts)))
            state["offset"] += got
            state["eof"] = state["offset"] >= size
        return have + got

    try:
        cur = 0
        n_buf = fill(cur, 0)
        while n_buf:
            eof = state["eof"]
            pin, pin_np = pins[cur], pin_nps[cur]
            cut_at = C.c_uint64(0)
            check(lib.mrg_fastq_block_cut(pin.data_ptr(), n_buf, 1 if eof else 0, C.byref(cut_at)))
            n_blk = int(cut_at.value)
            if n_blk == 0:
                if eof:
                    break
                raise DeviceIngestUnsupported("no record boundary inside %d bytes of text" % n_buf)
            # the tail behind the last record boundary starts the next block, whose read begins now
            have = n_buf - n_blk
            nxt = cur ^ 1
            if have:
                pin_nps[nxt][:have] = pin_np[n_blk:n_buf]
            if base is None:
                # trim_file.py:104-110: the first record's qualities decide the base the workers trim with,
                # the first 1000 records what the report says
                head = bytes(pin_np[:min(n_blk, 1 << 20)]).split(b"\n")
                quals = [head[i].rstrip(b"\r") for i in range(3, min(len(head), 4000), 4)]
                base = 64 if quals and any(c > 74 for c in quals[0]) else 33
                any64 = any(c > 74 for q in quals[:1000] for c in q)
            d_text[:n_blk].copy_(pin[:n_blk], non_blocking=True)
            cap = n_blk // (2 * max(int(min_len), 1) + 6) + 1
            info = _native.FastqDeviceInfo()
            W = 1
            while True:
                words = torch.empty((W, cap), dtype=torch.int64, device=dev)
                lens = torch.empty(cap, dtype=torch.uint8, device=dev)
                nmask = torch.empty((W, cap), dtype=torch.int64, device=dev)
                check(lib.mrg_fastq_parse_device_ad(engine._h, d_text.data_ptr(), n_blk, base, int(qual_cutoff), int(min_len), cut,
                                                    ad_seqs, W, cap, words.data_ptr(), lens.data_ptr(), nmask.data_ptr(),
                                                    C.byref(info), stream))
                if info.status:
                    raise DeviceIngestUnsupported("record %d of a block is not a plain four-line record (status %d)"
                                                  % (info.bad_record, info.status))
                if info.n_long == 0:
                    break
                if W == _native.MRG_MAX_WORDS:
                    raise DeviceIngestUnsupported("reads beyond 255 nt")
                W *= 2
            k = int(info.n_kept)
            total += int(info.n_records)
            kept += k
            max_len = max(max_len, int(info.max_len))
            outs.append((words[:, :k].clone(), lens[:k].clone(), nmask[:, :k].clone() if info.has_n else None))
            del words, lens, nmask
            torch.cuda.current_stream(dev).synchronize()
            if eof:
                break
            n_buf = fill(nxt, have)
            cur = nxt
    finally:
        if fh is not None:
            fh.close()
        if gz:
            lib.mrg_gz_close(gz)
        if pool is not None:
            pool.shutdown()
    W = max([o[0].shape[0] for o in outs] + [1])
    any_n = any(o[2] is not None for o in outs)

    def widen(t, k):
        if t is None:
            t = torch.zeros((W, k), dtype=torch.int64, device=dev)
        elif t.shape[0] < W:
            t = torch.cat([t, torch.zeros((W - t.shape[0], k), dtype=torch.int64, device=dev)], dim=0)
        return t
    if outs:
        words = torch.cat([widen(o[0], o[0].shape[1]) for o in outs], dim=1)
        lens = torch.cat([o[1] for o in outs])
        nmask = torch.cat([widen(o[2], o[0].shape[1]) for o in outs], dim=1) if any_n else None
    else:
        words = torch.empty((1, 0), dtype=torch.int64, device=dev)
        lens = torch.empty(0, dtype=torch.uint8, device=dev)
        nmask = None
    return dict(words=words, lens=lens, nmask=nmask, total=total, kept=kept, packed=kept, phred=64 if any64 else 33,
                max_len=max_len, long_reads=[])


def collapse(engine, words, lens, nmask=None, sample=None, n_samples=1, max_len=0):
    """Collapse raw packed reads on the engine's GPU.

    words [W, n] uint64, lens [n] uint8, nmask like words or None, sample [n] uint16 or None.
    Returns dict(words [W, U], lens [U], nmask|None, quant [U, S] uint32,
    length_hist {length: [count per sample]}) with the arrays on the host."""
    import torch
    dev = engine.device
    words = np.ascontiguousarray(words, dtype=np.uint64)
    W, n = words.shape
    d_words = torch.from_numpy(words.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.uint8)).to(dev)
    d_nmask = None if nmask is None else torch.from_numpy(
        np.ascontiguousarray(nmask, dtype=np.uint64).view(np.int64)).to(dev)
    d_sample = None
    if n_samples > 1:
        d_sample = torch.from_numpy(np.ascontiguousarray(sample, dtype=np.uint16).view(np.int16)).to(dev)
    cap = max(n, 1)
    u_words = torch.empty((W, cap), dtype=torch.int64, device=dev)
    u_lens = torch.empty(cap, dtype=torch.uint8, device=dev)
    u_nmask = None if d_nmask is None else torch.empty((W, cap), dtype=torch.int64, device=dev)
    quant = torch.empty((cap, n_samples), dtype=torch.int32, device=dev)
    hist = torch.zeros((256, n_samples), dtype=torch.int64, device=dev)
    n_unique = C.c_uint64(0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(engine._lib.mrg_collapse_run(
        engine._h, d_words.data_ptr(), W, d_lens.data_ptr(),
        None if d_nmask is None else d_nmask.data_ptr(),
        None if d_sample is None else d_sample.data_ptr(), n, n_samples, int(max_len), cap,
        u_words.data_ptr(), u_lens.data_ptr(), None if u_nmask is None else u_nmask.data_ptr(),
        quant.data_ptr(), hist.data_ptr(), C.byref(n_unique), stream))
    U = int(n_unique.value)
    h = hist.cpu().numpy()
    return dict(
        words=u_words[:, :U].cpu().numpy().view(np.uint64),
        lens=u_lens[:U].cpu().numpy(),
        nmask=None if u_nmask is None else u_nmask[:, :U].cpu().numpy().view(np.uint64),
        quant=quant[:U].cpu().numpy().view(np.uint32),
        length_hist={int(L): [int(x) for x in h[L]] for L in np.nonzero(h.sum(axis=1))[0]})
