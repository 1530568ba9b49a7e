#!/usr/bin/env python3
"""Do H2D and D2H copies of pinned memory overlap on this box, and on which stream pairs?
(bench.py's e2e leg depends on it.)"""
import time
import torch

dev = "cuda:0"
n = 512 << 20
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device=dev)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)


def run(s1, s2, both=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        d_in.copy_(h_in, non_blocking=True)
    if both:
        with torch.cuda.stream(s2):
            h_out.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


a, b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
hi, lo = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)
streams = [torch.cuda.Stream(dev) for _ in range(6)]
run(a, b)
print("H2D alone             %.2f ms" % run(a, b, both=False))
print("H2D + D2H, two streams %.2f ms" % run(a, b))
print("H2D + D2H, priorities  %.2f ms" % run(hi, lo))
for i in range(1, 6):
    print("H2D on s0 + D2H on s%d   %.2f ms" % (i, run(streams[0], streams[i])))
print("H2D + D2H, same stream %.2f ms" % run(a, a))
