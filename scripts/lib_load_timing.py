#!/usr/bin/env python3
"""How long until a library is resident?  Builds the synthetic libraries at a scale, caches each index as .mrgfm,
then -- what a user's second sample sees -- loads the file and makes the library resident, with the stage times of
`load_index` / `mrg_ctx_add_library` (MIRGE_AMD_TIMING).
    python scripts/lib_load_timing.py [scale=1.0] [keys=mrna,ncrna_others,mirna]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIRGE_AMD_TIMING"] = "1"
from mirge_amd import synth
from mirge_amd.engine import Engine
from mirge_amd.index import FmIndex

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
keys = (sys.argv[2] if len(sys.argv) > 2 else "mrna,ncrna_others,mirna").split(",")
libs = synth.SynthLibraries(seed=20181, scale=scale)
tmp = tempfile.mkdtemp(prefix="mrg_load_")
eng = Engine(0)
for key in keys:
    names, seqs = libs.libs[key]
    t0 = time.time()
    ix = FmIndex.build(names, seqs)
    t1 = time.time()
    path = os.path.join(tmp, key + ".mrgfm")
    ix.save(path)
    del ix
    t2 = time.time()
    print("[load] %s: %d bp, build %.2f s, save %.2f s (%.0f MB)" % (key, sum(map(len, seqs)), t1 - t0, t2 - t1, os.path.getsize(path) / 1e6), file=sys.stderr)
    for rep in range(2):
        t0 = time.time()
        ix = FmIndex.load(path)
        t1 = time.time()
        eng.add_library("%s_%d" % (key, rep), ix, exact_dict=(key in ("mrna", "mirna", "pre_trna")))
        t2 = time.time()
        print("[load] %s run %d: load_index %.2f s, add_library %.2f s" % (key, rep, t1 - t0, t2 - t1), file=sys.stderr)
        del ix
