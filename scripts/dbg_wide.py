import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mirge_amd import pack
from mirge_amd.engine import Engine, ReadSet
from mirge_amd.index import FmIndex
from oracle import model
rng = np.random.default_rng(5)
def rnd(n): return "".join("ACGT"[c] for c in rng.integers(0, 4, n))
seqs = [rnd(60) + "A" * int(rng.integers(20, 60)) for _ in range(40)]
seqs += [("ACGT" * 30)[:int(rng.integers(40, 120))] for _ in range(10)]
seqs += [rnd(30) + "CACACACACACACACACACACACACACA" + rnd(10) for _ in range(10)]
seqs += [seqs[3], seqs[3], "A" * 200, "T" * 90]
seqs += [rnd(12) + "A" * 120 + rnd(5) for _ in range(60)]
names = ["rep%d" % i for i in range(len(seqs))]
ix = FmIndex.build(names, seqs)
reads = ["A" * L for L in (16, 22, 25, 30, 40)] + ["A" * 21 + "C", "C" + "A" * 21, "ACGT" * 6,
         "CGTA" * 5 + "CG", "CA" * 12, "AC" * 11 + "G", "T" * 22, "T" * 19 + "AAA", "G" * 22,
         seqs[3][40:62], seqs[3][50:75], "A" * 10 + "N" + "A" * 11]
reads += [s[int(o):int(o) + 22] for s in seqs[:30] for o in rng.integers(0, len(s) - 22, 3)]
reads = list(dict.fromkeys(reads))
w,l,nm = pack.pack_reads(reads)
print(len(reads), w.shape)
eng = Engine(0); eng.add_library("rep", ix)
lib = model.Library(names, seqs)
for wr in (100000, 256, 2):
    eng.set_option("wide_rows", wr)
    for cfg in ((28,0,2,0,0),(28,1,2,0,0),(1024,1,1,0,0),(1024,2,2,1,2)):
        passes = eng.make_passes([dict(lib="rep", seed_len=cfg[0], max_mm_seed=cfg[1], max_mm_total=cfg[2], trim5=cfg[3], trim3=cfg[4])])
        res = eng.cascade(ReadSet(w,l,nm,None,device=eng.device), passes)
        pid, ref, pos, mm = res.to_host()
        t5,t3=cfg[3],cfg[4]
        trimmed = [r[t5:len(r)-t3] if t3 else r[t5:] for r in reads]
        wr_, wp, wm = model.align_batch(lib, trimmed, *cfg[:3])
        bad = [(reads[i], (int(ref[i]),int(pos[i]),int(mm[i])) if pid[i]==0 else None, (int(wr_[i]),int(wp[i]),int(wm[i]))) for i in range(len(reads)) if ((int(ref[i]),int(pos[i]),int(mm[i])) if pid[i]==0 else (-1,-1,-1)) != (int(wr_[i]),int(wp[i]),int(wm[i]))]
        print(wr, cfg, 'bad', len(bad), bad[:4], res.stats[0]['candidates'])
