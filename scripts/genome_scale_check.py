#!/usr/bin/env python3
"""Genome-scale check of the -ai filters' engine path on one GPU.

Builds ONE genome part of --bases random bases (default 300 M: a human chromosome is 50-250 M;
the index is < 2^31 rows, the suffix array > 2 GB, the big jump table k = 14), loads it into HBM,
and runs mrg_count_best (`-n 1 -a -3 2` and `-n 0`) on --reads 22-nt reads cut from it with 0-2
substitutions plus random ones, both strands as a2i.EngineGenome submits them.  Checks, without
the oracle (an exhaustive scan of 3e8 bases per read is too slow):
  * a read cut from the genome unchanged has best_mm 0 and count >= 1 on its own strand;
  * its exact multiplicity equals bytes.count() of the 20-mer in the text for a sample;
  * a read with one planted substitution aligns with best_mm <= 1 under `-n 1`;
  * `-n 0` reports an alignment iff `-n 1` reports one with zero seed mismatches... (subset check).
Prints one JSON line with build time, HBM footprint and reads/s.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=int, default=300_000_000)
    ap.add_argument("--reads", type=int, default=2_000_000)
    args = ap.parse_args()
    import torch
    from mirge_amd.engine import Engine, ReadSet
    from mirge_amd.index import FmIndex
    rng = np.random.default_rng(2026)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    t0 = time.time()
    n_chr = 3
    per = args.bases // n_chr
    seqs = [acgt[rng.integers(0, 4, per)].tobytes() for _ in range(n_chr)]
    ix = FmIndex.build(["chr%d" % (i + 1) for i in range(n_chr)], [s.decode() for s in seqs])
    t_build = time.time() - t0
    inf = ix.info
    eng = Engine(0)
    eng.add_library("g", ix)
    L = 22
    n = args.reads
    codes = np.empty((n, L), dtype=np.uint8)
    planted = np.zeros(n, dtype=np.int32)
    src_chr = rng.integers(0, n_chr, n)
    src_off = rng.integers(0, per - L, n)
    from_genome = rng.random(n) < 0.8
    lut = np.zeros(256, dtype=np.uint8)
    lut[ord("C")], lut[ord("G")], lut[ord("T")] = 1, 2, 3
    arrs = [np.frombuffer(s, dtype=np.uint8) for s in seqs]
    for c in range(n_chr):
        m = np.nonzero(src_chr == c)[0]
        idx = src_off[m, None] + np.arange(L)[None, :]
        codes[m] = lut[arrs[c][idx]]
    rnd = np.nonzero(~from_genome)[0]
    codes[rnd] = rng.integers(0, 4, (len(rnd), L))
    n_sub = np.where(from_genome, rng.integers(0, 3, n), 0)
    for k in (1, 2):
        m = np.nonzero(n_sub >= k)[0]
        pos = rng.integers(0, L - 2, len(m))          # inside the 20 nt that remain after -3 2
        codes[m, pos] = (codes[m, pos] + rng.integers(1, 4, len(m))) % 4
    trimmed = codes[:, :L - 2]
    rc = (3 - trimmed[:, ::-1])
    both = np.concatenate([trimmed, rc]).astype(np.uint64)
    words = np.zeros((1, 2 * n), dtype=np.uint64)
    for i in range(L - 2):
        words[0] |= both[:, i] << np.uint64(2 * i)
    lens = np.full(2 * n, L - 2, dtype=np.uint8)
    rs = ReadSet(words, lens, None, None, device=eng.device)
    out = {}
    for n_seed in (1, 0):
        eng.count_best(rs, "g", seed_len=28, max_mm_seed=n_seed, max_mm_total=2)   # warm
        torch.cuda.synchronize()
        t1 = time.time()
        mm, cnt = eng.count_best(rs, "g", seed_len=28, max_mm_seed=n_seed, max_mm_total=2)
        dt = time.time() - t1
        out[n_seed] = (mm, cnt, dt)
    mm1, cnt1, dt1 = out[1]
    mm0, cnt0, dt0 = out[0]
    fwd = slice(0, n)
    exact = from_genome & (n_sub == 0)
    assert (mm1[fwd][exact] == 0).all() and (cnt1[fwd][exact] >= 1).all()
    assert (mm0[fwd][exact] == 0).all()
    ok = from_genome & (n_sub <= 1)      # the 20-nt read is all seed: -n 1 tolerates one substitution
    assert (mm1[fwd][ok] <= n_sub[ok]).all(), "a planted read must align within its substitutions"
    # -n 0 needs a mismatch-free seed (= the whole 20-mer, total <= 2 is then 0): aligned iff exact somewhere
    assert ((mm0 < 255) == (mm1 == 0)).all()
    # exact multiplicity against a plain substring count, forward strand, for a sample
    sample = np.nonzero(exact)[0][:40]
    for i in sample:
        q = bytes(acgt[trimmed[i]])
        want = sum(s.count(q) for s in seqs)   # non-overlapping count: fine for random 20-mers
        assert int(cnt1[i]) == min(want, 255) and int(cnt0[i]) == min(want, 255), (i, want, cnt1[i])
    hbm = inf.bytes_fm + inf.bytes_sa + sum((1 << (2 * k)) + 1 for k in inf.ftab_ks if k) * 4 + (inf.text_words * 4)
    print(json.dumps({
        "check": "genome part at scale", "bases": int(inf.n_bases), "ftab_ks": [int(k) for k in inf.ftab_ks],
        "index_build_s": round(t_build, 1), "hbm_bytes": int(hbm),
        "reads_both_strands": 2 * n, "n1_ms": round(dt1 * 1e3, 2), "n0_ms": round(dt0 * 1e3, 2),
        "n1_reads_per_s": round(2 * n / dt1), "n0_reads_per_s": round(2 * n / dt0),
        "aligned_n1": int((mm1 < 255).sum()), "unique_best_n1": int(((mm1 < 255) & (cnt1 == 1)).sum()),
        "note": "times include the device->host copy of the two uint8 result arrays"}))


if __name__ == "__main__":
    sys.exit(main())
