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
               threads=0):
    """Returns dict(words [W, n], lens, nmask|None, total, kept, packed, phred, max_len, long_reads):
    `kept` = reads that survive trimming (the reference's trimmedReads), `packed` = n of them in the
    arrays, `long_reads` = the kept reads beyond 128 nt (ASCII; too long for four packed words)."""
    lib = _native.load()
    h = C.c_void_p()
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
