"""Does running the cascades of several CHUNKS of one read set on separate HIP streams overlap their launches
(the miss-bound large-library launch of one chunk beside the instruction-bound launches of another)?
  python scripts/overlap_probe.py [--reads N] [--chunks 1,2,4] [--grid-pct 100,50]
Prints ms per whole read set for every (chunks, grid_pct); results are not checked here (bench.py does that)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mirge_amd import synth  # noqa: E402
from mirge_amd._native import check  # noqa: E402
from mirge_amd.engine import Engine, ReadSet  # noqa: E402
from mirge_amd.index import FmIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--chunks", default="1,2,4,8")
    ap.add_argument("--grid-pct", default="100,60,40")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from concurrent.futures import ThreadPoolExecutor
    keys = list(synth.LIB_KEYS)
    libs = synth.SynthLibraries(seed=20181, scale=1.0)
    pool = ThreadPoolExecutor(max_workers=len(keys))
    futures = {k: pool.submit(FmIndex.build, *libs.libs[k]) for k in keys}
    words, lens, quant = synth.global_read_slice(libs, args.reads, 0, args.reads, workload="cascade", seed0=355)
    index = {k: f.result() for k, f in futures.items()}
    eng = Engine(0)
    for k in keys:
        eng.add_library(k, index[k])
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    passes = eng.mirge_passes()
    n_pass = len(passes)
    dev = eng.device
    rs = ReadSet(words, lens, None, quant, device=dev)
    eng.prepare(passes, rs.W, rs.min_len, rs.max_len)
    n = rs.n
    packed = torch.empty(n, dtype=torch.int32, device=dev)
    lib = eng._lib
    for n_chunks in [int(x) for x in args.chunks.split(",")]:
        bounds = [(n * i // n_chunks) // 4 * 4 for i in range(n_chunks)] + [n]
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_chunks)]
        wss, pcs = [], []
        for c in range(n_chunks):
            need = C.c_uint64()
            check(lib.mrg_cascade_workspace_bytes(bounds[c + 1] - bounds[c], C.byref(need)))
            wss.append(torch.empty(need.value, dtype=torch.uint8, device=dev))
            pcs.append(torch.zeros(2 * n_pass, dtype=torch.int64, device=dev))
        for pct in [int(x) for x in args.grid_pct.split(",")]:
            eng.set_option("grid_pct", pct)

            def run_all():
                for c in range(n_chunks):
                    lo, hi = bounds[c], bounds[c + 1]
                    eng.set_option("hint_min_len", rs.min_len)
                    eng.set_option("hint_max_len", rs.max_len)
                    check(lib.mrg_cascade_run_packed(
                        eng._h, rs.words.data_ptr() + 8 * lo, 1, rs.lens.data_ptr() + lo, None, hi - lo, passes, n_pass,
                        packed.data_ptr() + 4 * lo, pcs[c].data_ptr(), wss[c].data_ptr(), wss[c].numel(),
                        C.c_void_p(streams[c].cuda_stream)))
            run_all()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                run_all()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / args.reps
            print("chunks %d grid_pct %3d: %.3f ms per %d reads" % (n_chunks, pct, ms, n), flush=True)
        eng.set_option("grid_pct", 100)


if __name__ == "__main__":
    main()
