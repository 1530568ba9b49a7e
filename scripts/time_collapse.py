#!/usr/bin/env python3
"""Time the GPU collapse (mrg_collapse_run) on synthetic raw reads: M reads/s."""
import ctypes as C
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mirge_amd import synth
from mirge_amd._native import check
from mirge_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
libs = synth.SynthLibraries(scale=0.05)
w = np.concatenate([synth.synth_reads_packed(libs, min(10_000_000, n - lo), seed=9 + lo)
                    for lo in range(0, n, 10_000_000)])
eng = Engine(0)
dev = eng.device
for n_samples in (1, 4):
    d_words = torch.from_numpy(w.view(np.int64)).to(dev)
    d_lens = torch.full((n,), 22, dtype=torch.uint8, device=dev)
    d_sample = torch.from_numpy((np.arange(n) % n_samples).astype(np.int16)).to(dev)
    u_words = torch.empty((1, n), dtype=torch.int64, device=dev)
    u_lens = torch.empty(n, dtype=torch.uint8, device=dev)
    quant = torch.empty((n, n_samples), dtype=torch.int32, device=dev)
    hist = torch.zeros((256, n_samples), dtype=torch.int64, device=dev)
    nu = C.c_uint64()
    for max_len in (22, 0):
        ts = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            check(eng._lib.mrg_collapse_run(eng._h, d_words.data_ptr(), 1, d_lens.data_ptr(), None,
                                            d_sample.data_ptr() if n_samples > 1 else None, n, n_samples, max_len, n,
                                            u_words.data_ptr(), u_lens.data_ptr(), None, quant.data_ptr(),
                                            hist.data_ptr(), C.byref(nu), None))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        best = min(ts[1:])
        print("collapse n=%d S=%d %s: %.1f ms -> %.0f M reads/s, %d uniques" %
              (n, n_samples, "keys-only (key = read + length [+ sample])" if max_len else "general (ids sorted column by column)", best * 1e3, n / best / 1e6,
               nu.value), flush=True)
