"""mrg_collapse_run alone on the headline read set (100 M raw 22-nt reads of bench.py's seeded mixture): wall time per
call, the option to compare the general path, and -- under rocprofv3 --kernel-trace --stats -- the time of each of its
kernels.  usage: python scripts/collapse_bench.py [--reads N] [--reps K] [--general] [--samples S]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--general", action="store_true")
    ap.add_argument("--samples", type=int, default=1)
    args = ap.parse_args()
    import torch
    from mirge_amd import synth
    from mirge_amd.engine import Engine
    libs = synth.SynthLibraries(seed=20181, scale=1.0)
    t0 = time.time()
    words, lens, _q = synth.global_read_slice(libs, args.reads, 0, args.reads)
    print("reads generated in %.1f s" % (time.time() - t0), file=sys.stderr)
    eng = Engine(0)
    if args.general:
        eng.set_option("collapse_fast", 0)
    dev = eng.device
    d_words = torch.from_numpy(words.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(lens).to(dev)
    S = args.samples
    d_sample = None
    if S > 1:
        d_sample = torch.from_numpy((np.arange(args.reads) % S).astype(np.int16)).to(dev)
    n = args.reads
    bufs = (torch.empty((1, n), dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.uint8, device=dev), None,
            torch.empty((n, S), dtype=torch.int32, device=dev))
    ts = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        urs, hist = eng.collapse(d_words, d_lens, None, d_sample, S, 22, out=bufs)
        ts.append(time.perf_counter() - t0)
    u = urs.words[0]
    ok = bool((u[1:] > u[:-1]).all()) and int(urs.quant.sum()) == n
    print(json.dumps(dict(raw=n, unique=urs.n, samples=S, general=args.general, ms=[round(t * 1e3, 3) for t in ts], sorted_and_complete=ok)))


if __name__ == "__main__":
    main()
