#!/usr/bin/env python3
"""The command line at scale: a FASTQ file of N reads, mostly distinct (the shape of BASELINE
configs[1]: "collapsed unique reads are mostly misses"), through `python -m mirge_amd annotate`.
Reports wall time per stage, peak host memory and the size of the tables; checks the row counts of
mapped.csv + unmapped.csv against the number of unique reads and the category totals against the
count vector.

    python scripts/cli_scale_check.py [n_reads=12000000] [scale=0.2] [device-ingest=0]
"""
import json
import os
import resource
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mirge_amd import cli, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
device_ingest = len(sys.argv) > 3 and sys.argv[3] == "1"
tmp = tempfile.mkdtemp(prefix="mrg_cli_scale_")
libs = synth.SynthLibraries(scale=scale)
libs.write_layout(os.path.join(tmp, "libs"), species="syn", db="miRBase")
t0 = time.time()
mix = dict(mirna_exact=0.10, isomir=0.03, trna=0.01, snorna=0.01, rrna_ncrna=0.01, mrna=0.01, polyt=0.01, random=0.82)
fq = os.path.join(tmp, "big.fastq")
L = 22
with open(fq, "wb") as fh:
    for lo in range(0, n, 4_000_000):
        m = min(4_000_000, n - lo)
        w = synth.synth_reads_packed(libs, m, seed=77 + lo, mix=mix)
        rec = np.empty((m, 3 + L + 3 + L + 1), dtype=np.uint8)   # "@r\n" SEQ "\n+\n" QUAL "\n"
        rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
        codes = ((w[:, None] >> (2 * np.arange(L, dtype=np.uint64))[None, :]) & np.uint64(3)).astype(np.uint8)
        rec[:, 3:3 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
        rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 6 + L:6 + 2 * L] = ord("I")
        rec[:, 6 + 2 * L] = ord("\n")
        fh.write(rec.tobytes())
gen_s = time.time() - t0
t0 = time.time()
out = cli.annotate_main(cli.build_parser().parse_args(
    ["annotate", "-s", fq, "-lib", os.path.join(tmp, "libs"), "-sp", "syn", "-o", tmp, "-di", "-cpu", "16"] +
    (["--device-ingest"] if device_ingest else [])))
wall = time.time() - t0
# a second run finds the indexes cached next to the library (`.mrgfm`): what a user's later samples see
t0 = time.time()
out = cli.annotate_main(cli.build_parser().parse_args(
    ["annotate", "-s", fq, "-lib", os.path.join(tmp, "libs"), "-sp", "syn", "-o", os.path.join(tmp, "second"), "-di", "-cpu", "16"] +
    (["--device-ingest"] if device_ingest else [])))
wall2 = time.time() - t0
rows = {}
for fn in ("mapped.csv", "unmapped.csv"):
    with open(os.path.join(out["outdir"], fn), "rb") as fh:
        rows[fn] = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b"")) - 1
qs = out["logDic"]["quantStats"][0]
assert rows["mapped.csv"] + rows["unmapped.csv"] == out["n_unique"] == qs["trimmedUniq"], (rows, out["n_unique"], qs["trimmedUniq"])
cats = sum(qs[k] for k in ("mirnaReads", "hairpinReads", "maturetrnaReads", "pretrnaReads", "snornaReads", "rrnaReads",
                           "ncrnaOthersReads", "mrnaReads", "remReads"))
assert cats == qs["trimmedReads"] == n, (cats, qs["trimmedReads"], n)
print(json.dumps(dict(
    reads=n, unique_reads=out["n_unique"], fastq_bytes=os.path.getsize(fq), fastq_generation_s=round(gen_s, 1),
    annotate_wall_s=round(wall, 1), annotate_wall_second_run_s=round(wall2, 1), device_ingest=device_ingest, peak_rss_gb=round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2),
    mapped_rows=rows["mapped.csv"], unmapped_rows=rows["unmapped.csv"],
    mapped_csv_bytes=os.path.getsize(os.path.join(out["outdir"], "mapped.csv")),
    unmapped_csv_bytes=os.path.getsize(os.path.join(out["outdir"], "unmapped.csv")),
    annot_stats=[(a["readsProcessed"], a["readsAligned"]) for a in out["logDic"]["annotStats"]])))
