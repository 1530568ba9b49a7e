#!/usr/bin/env python3
"""FASTQ ingest throughput on this box's host cores: a
synthetic FASTQ of --reads reads (22 nt + Illumina adapter tail, 51 cycles) written once, then
mrg_fastq_load (csrc/fastq.cpp) timed for plain text, gzip and `-ad illumina`, over a range of
worker-thread counts.  Prints one JSON object (reads/s per configuration)."""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=8_000_000)
    ap.add_argument("--threads", default="8,16,32,64,128,256")
    args = ap.parse_args()
    from mirge_amd import ingest
    rng = np.random.default_rng(3)
    d = tempfile.mkdtemp(prefix="ingest_")
    plain, trimmed = os.path.join(d, "trimmed.fastq"), os.path.join(d, "raw.fastq")
    adapter = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
    t0 = time.time()
    n = args.reads
    codes = rng.integers(0, 4, (n, 22), dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = acgt[codes]                                    # [n, 22] bytes
    def write(path, with_adapter):
        L = 51 if with_adapter else 22
        rec = np.empty((n, 2 * L + 12 + 8), dtype=np.uint8)   # "@r" + 8 digits + \n + seq + \n+\n + qual + \n
        ids = np.char.zfill(np.arange(n).astype("U8"), 8).astype("S8").view(np.uint8).reshape(n, 8)
        rec[:, 0:2] = np.frombuffer(b"@r", dtype=np.uint8)
        rec[:, 2:10] = ids
        rec[:, 10] = 10
        rec[:, 11:33] = seqs
        if with_adapter:
            rec[:, 33:33 + 29] = np.frombuffer(adapter.encode(), dtype=np.uint8)
        rec[:, 11 + L] = 10
        rec[:, 12 + L] = ord("+")
        rec[:, 13 + L] = 10
        rec[:, 14 + L:14 + 2 * L] = ord("I")
        rec[:, 14 + 2 * L] = 10
        rec = rec[:, :15 + 2 * L]
        with open(path, "wb") as fh:
            fh.write(np.ascontiguousarray(rec).tobytes())
    write(plain, False)
    write(trimmed, True)
    gz = plain + ".gz"
    with open(plain, "rb") as fi, gzip.open(gz, "wb", compresslevel=4) as fo:
        fo.write(fi.read())
    gen_s = time.time() - t0
    out = dict(reads=n, cores=os.cpu_count(), generate_s=round(gen_s, 1), runs=[])
    for name, path, ad in (("plain", plain, "none"), ("gzip", gz, "none"), ("illumina_adapter", trimmed, "illumina")):
        for th in [int(x) for x in args.threads.split(",")]:
            if th > (os.cpu_count() or 1):
                continue
            best = None
            for _ in range(2):
                t1 = time.perf_counter()
                fq = ingest.load_fastq(path, adapter=ad, threads=th)
                dt = time.perf_counter() - t1
                assert fq["kept"] == n, (name, fq["kept"])
                best = dt if best is None else min(best, dt)
                del fq
            out["runs"].append(dict(input=name, threads=th, seconds=round(best, 3), m_reads_per_s=round(n / best / 1e6, 2)))
            print("[ingest] %s threads=%d: %.2f M reads/s" % (name, th, n / best / 1e6), file=sys.stderr, flush=True)
    # the device parser (csrc/ingest.hip): raw text blocks uploaded, records split / trimmed / packed on the GPU
    try:
        import torch
        if torch.cuda.is_available():
            from mirge_amd.engine import Engine
            eng = Engine(0)
            for name, path, ad in (("plain", plain, "none"), ("gzip", gz, "none"), ("illumina_adapter", trimmed, "illumina")):
                for th in (4, 16, 32):
                    best = None
                    for _ in range(2):
                        t1 = time.perf_counter()
                        fq = ingest.load_fastq_device(eng, path, adapter=ad, read_threads=th)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t1
                        assert fq["kept"] == n, (name, fq["kept"])
                        best = dt if best is None else min(best, dt)
                        del fq
                    out["runs"].append(dict(input=name + " (device parser)", threads=th, seconds=round(best, 3),
                                            m_reads_per_s=round(n / best / 1e6, 2)))
                    print("[ingest] %s, device parser, %d read threads: %.2f M reads/s" % (name, th, n / best / 1e6), file=sys.stderr,
                          flush=True)
                    if name == "gzip":
                        break
    except Exception as e:
        out["device_error"] = repr(e)
    for f in (plain, trimmed, gz):
        os.remove(f)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
