import sys, numpy as np
sys.path.insert(0, '.')
from oracle import model
from tests.util import LIB_ORDER, World
from mirge_amd.engine import Engine, ReadSet
model.build()
w = World(scale=0.05, n_fixed=20000, n_var=6000, with_n=False, max_var_len=32)
eng = Engine(0)
for k in LIB_ORDER:
    eng.add_library(k, w.index[k])
ref = model.fm_cascade(w.views, w.passes, w.words, w.lens, None)
for fuse in (1, 0):
    eng.set_option("fuse", fuse)
    rs = ReadSet(w.words, w.lens, None, None, device=eng.device)
    res = eng.cascade(rs, eng.mirge_passes())
    got = res.to_host()
    bad = np.nonzero(got[0] != ref["pass_id"])[0]
    print("fuse", fuse, "n", len(w.lens), "bad", len(bad), bad[:20], [(int(w.lens[i]), int(got[0][i]), int(ref["pass_id"][i])) for i in bad[:20]])
    print([ (s["processed"], s["aligned"]) for s in res.stats])
    print([ (int(a), int(b)) for a, b in ref["stats"][:, :2]])
