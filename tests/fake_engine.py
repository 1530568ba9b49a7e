"""An oracle-backed stand-in for mirge_amd.engine.Engine on CPU tensors (TEST INFRASTRUCTURE).

The multi-process path of the command line (`--gpus N`: rank 0 collapses, every rank annotates its
shard, one all-reduce, rank 0 writes the tables) has to be exercised without GPUs: gloo, world
size 2, and this class where the product would create an Engine.  It implements exactly the
methods cli.annotate_main calls, with the oracle's CPU port (oracle/fm_cpu.c) doing the matching.
"""
import numpy as np
import torch

from mirge_amd.engine import MIRGE_PASS_TABLE, ReadSet
from oracle import model


class _Result:
    def __init__(self, pass_id, ref_id, pos, mm, pass_counts, stats, n_pass):
        self.pass_id, self.ref_id, self.pos, self.mm = pass_id, ref_id, pos, mm
        self.pass_counts, self.stats, self.n_pass = pass_counts, stats, n_pass


class OracleEngine:
    def __init__(self, device=0):
        self.device = "cpu"
        self.libs, self.indexes = {}, {}

    def add_library(self, key, index):
        self.libs[key] = len(self.libs)
        self.indexes[key] = index
        return self.libs[key]

    def mirge_passes(self, spike_in=False):
        rows = MIRGE_PASS_TABLE[:10 if spike_in else 9]
        return [dict(lib=k, min_len=a, max_len=b, seed_len=s, max_mm_seed=ms, max_mm_total=mt, trim5=t5, trim3=t3,
                     poly_t=pt) for (k, a, b, s, ms, mt, t5, t3, pt) in rows]

    def counts_len(self, n_mirna, n_samples, n_pass):
        return 2 * n_mirna * n_samples + (n_pass + 2) * n_samples

    def collapse(self, words, lens, nmask=None, sample=None, n_samples=1, max_len=0):
        w = words.numpy().view(np.uint64)
        W, n = w.shape
        l = lens.numpy()
        nm = np.zeros_like(w) if nmask is None else nmask.numpy().view(np.uint64)
        key = np.concatenate([l[None, :].astype(np.uint64), w[::-1], nm[::-1]], axis=0).T   # (length, bases)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        U = uniq.shape[0]
        smp = np.zeros(n, dtype=np.int64) if sample is None else sample.numpy().astype(np.int64)
        quant = np.zeros((U, n_samples), dtype=np.int64)
        np.add.at(quant, (inv, smp), 1)
        hist = np.zeros((256, n_samples), dtype=np.int64)
        np.add.at(hist, (l.astype(np.int64), smp), 1)
        u_l = uniq[:, 0].astype(np.uint8)
        u_w = np.ascontiguousarray(uniq[:, 1:1 + W][:, ::-1].T)
        u_n = np.ascontiguousarray(uniq[:, 1 + W:][:, ::-1].T)
        rs = ReadSet.from_device(torch.from_numpy(u_w.view(np.int64).copy()), torch.from_numpy(u_l.copy()),
                                 None if nmask is None else torch.from_numpy(u_n.view(np.int64).copy()),
                                 torch.from_numpy(quant.astype(np.int32)), 0, int(max_len) if max_len else 255)
        return rs, torch.from_numpy(hist)

    def cascade(self, reads, passes, out=None):
        keys = list(self.libs)
        pd = [dict(p, lib=keys.index(p["lib"])) for p in passes]
        views = [self.indexes[k].view() for k in keys]
        w = reads.words.numpy().view(np.uint64)
        nm = None if reads.nmask is None else reads.nmask.numpy().view(np.uint64)
        ref = model.fm_cascade(views, pd, w, reads.lens.numpy(), nm, wstop=8, ftab=True)
        stats = [dict(processed=int(r[0]), aligned=int(r[1]), steps=int(r[2]), candidates=int(r[3]), lookups=int(r[4]),
                      ms=1.0, lds_bytes=0, lds_mode=0, group=i, kbits_log2=0, pair_anchor=0) for i, r in enumerate(ref["stats"])]
        pc = torch.from_numpy(ref["stats"][:, :2].astype(np.int64).reshape(-1).copy())
        return _Result(torch.from_numpy(ref["pass_id"]), torch.from_numpy(ref["ref_id"]), torch.from_numpy(ref["pos"]),
                       torch.from_numpy(ref["mm"]), pc, stats, len(passes))

    def tally(self, reads, result, n_mirna, canon_pass=0, isomir_pass=8, counts=None):
        c = model.tally(result.pass_id.numpy(), result.ref_id.numpy(), reads.quant.numpy().view(np.uint32), n_mirna,
                        result.n_pass, canon_pass, isomir_pass)
        t = torch.from_numpy(c.astype(np.int64))
        if counts is None:
            return t
        counts += t
        return counts

    def cascade_long(self, seqs, passes, pass_counts=None, stats=None):
        """Engine.cascade_long by the oracle's exhaustive scan (oracle.cascade.scan_cascade: no index, reads of any
        length)."""
        from oracle import cascade as ocas
        libs = {}
        for p in passes:
            ix = self.indexes[p["lib"]]
            if p["lib"] not in libs:
                libs[p["lib"]] = model.Library(ix.names, [ix.sequence(i) for i in range(ix.n_ref)])
        pass_id, ref_id, pos, mm, counts = ocas.scan_cascade(libs, passes, list(seqs))
        own = [dict(processed=c[0], aligned=c[1], steps=0, candidates=0, lookups=0, ms=0.0) for c in counts]
        if stats is not None:
            for a, b in zip(own, stats):
                for k, v in a.items():
                    b[k] += v
        return pass_id, ref_id, pos, mm, own
