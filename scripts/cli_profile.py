#!/usr/bin/env python3
"""Where does a FASTQ-to-tables run spend its time?  Writes two synthetic FASTQ samples (adapter
attached, as raw small-RNA reads have it), builds the library layout, runs `annotate` under
cProfile and prints the phase log plus the top cumulative entries.
    python scripts/cli_profile.py [reads per sample] [scale] [warm]"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mirge_amd import cli, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
tmp = tempfile.mkdtemp(prefix="mirge_cli_")
libs = synth.SynthLibraries(seed=20181, scale=scale)
libs.write_layout(os.path.join(tmp, "libs"), species="syn", db="miRBase")
ad = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
fastqs = []
t0 = time.time()
for si in range(2):
    codes = synth.synth_reads(libs, n, seed=100 + si)
    seqs = acgt[codes].view("S22").ravel()
    p = os.path.join(tmp, "s%d.fastq" % si)
    with open(p, "wb") as fh:
        tail = (ad + "ACGTACGT")[:28].encode()
        q = b"I" * 50
        fh.write(b"".join(b"@r%d\n%s%s\n+\n%s\n" % (i, s, tail, q) for i, s in enumerate(seqs)))
    fastqs.append(p)
print("wrote 2 x %d reads in %.1f s" % (n, time.time() - t0))
args = cli.build_parser().parse_args(["annotate", "-s"] + fastqs + ["-lib", os.path.join(tmp, "libs"), "-sp", "syn",
                                      "-o", tmp, "-ad", "illumina", "-cpu", "16", "-di"])
if len(sys.argv) > 3 and sys.argv[3] == "warm":     # a first, unprofiled run: the indexes are cached next to the library
    cli.annotate_main(args)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
out = cli.annotate_main(args)
pr.disable()
print("annotate: %.1f s for %d raw reads, %d unique" % (time.time() - t0, 2 * n, out["n_unique"]))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(34)
print(s.getvalue()[:6000])
