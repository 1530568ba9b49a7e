"""Columnar host API over the C-ABI: one Engine per GPU.

This is the scale path (10^7..10^8 reads): packed reads, counts and
assignments live in HBM as flat arrays instead of the reference's dict of dicts
(`seqDic`, MAIN:318-321; infeasible at 10^8 entries).  `mirge_amd.annotate`
materialises the reference's dict shapes from these arrays for small inputs.

torch is used only for device memory, streams and (in mirge_amd.dist)
torch.distributed; every kernel is launched through libmirge_amd.so.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import PassCfg, PassStats, check

V_MODE_SEED = 1024  # "-v": the seed region is the whole read
DEFAULT_WSTOP = 8    # the context's default "wstop" option (mrg_ctx_set_option)

# The nine (ten) bowtie command lines of runAnnotationPipeline.py:577-599 / :688:
# (library key, min_len, max_len, seed_len, max_mm_seed, max_mm_total, trim5, trim3, poly_t)
MIRGE_PASS_TABLE = [
    ("mirna", 0, 25, 28, 0, 2, 0, 0, 0),                    # len < 26 ; -n 0
    ("hairpin", 26, 255, 28, 1, 2, 0, 0, 0),                # len > 25 ; -n 1
    ("mature_trna", 0, 255, V_MODE_SEED, 1, 1, 0, 0, 0),    # -v 1 -a --best --strata
    ("pre_trna", 0, 255, V_MODE_SEED, 0, 0, 0, 0, 1),       # poly-T rule ; -v 0 -a --best --strata
    ("snorna", 0, 255, 28, 1, 2, 0, 0, 0),                  # -n 1
    ("rrna", 0, 255, 28, 1, 2, 0, 0, 0),                    # -n 1
    ("ncrna_others", 0, 255, 28, 1, 2, 0, 0, 0),            # -n 1
    ("mrna", 0, 255, 28, 0, 2, 0, 0, 0),                    # -n 0
    ("mirna", 0, 255, V_MODE_SEED, 2, 2, 1, 2, 0),          # -5 1 -3 2 -v 2 --best
    ("spike-in", 0, 255, 28, 0, 2, 0, 0, 0),                # -n 0 (only with -spikeIn)
]
# libraries a pass searches WITHOUT seed mismatch: a large one of them gets an exact-match dictionary too
EXACT_LIBS = frozenset(row[0] for row in MIRGE_PASS_TABLE if row[4] == 0)
DICT_SMALL_BASES = 1 << 22   # csrc/dict_index.hpp: kDictSmallBases / kDictMaxBases
DICT_MAX_BASES = 1 << 30
CANON_PASS = 0   # annot slot 1 "exact miRNA"
ISOMIR_PASS = 8  # annot slot 9 "isomiR miRNA"


def _torch():
    import torch
    return torch


class ReadSet:
    """Packed reads resident in HBM."""

    def __init__(self, words, lens, nmask=None, quant=None, device="cuda:0"):
        torch = _torch()
        words = np.ascontiguousarray(words, dtype=np.uint64)
        self.W, self.n = words.shape
        self.device = torch.device(device)
        self.words = torch.from_numpy(words.view(np.int64)).to(self.device)
        lens = np.ascontiguousarray(lens, dtype=np.uint8)
        self.min_len = int(lens.min()) if lens.size else 0
        self.max_len = int(lens.max()) if lens.size else 255
        self.lens = torch.from_numpy(lens).to(self.device)
        self.nmask = None
        if nmask is not None:
            nm = np.ascontiguousarray(nmask, dtype=np.uint64)
            self.nmask = torch.from_numpy(nm.view(np.int64)).to(self.device)
        self.quant = None
        if quant is not None:
            q = np.ascontiguousarray(quant, dtype=np.uint32)
            if q.ndim == 1:
                q = q[:, None]
            self.quant = torch.from_numpy(q.view(np.int32)).to(self.device)

    @classmethod
    def from_device(cls, words, lens, nmask=None, quant=None, min_len=0, max_len=255):
        """Wrap tensors that already live in HBM (words int64 [W, n], lens uint8 [n], nmask like
        words or None, quant int32 [n, S] or None): nothing is copied.  min_len / max_len is the
        caller's knowledge of the batch's length range (0 / 255 = unknown)."""
        self = cls.__new__(cls)
        self.W, self.n = int(words.shape[0]), int(words.shape[1])
        self.device = words.device
        self.words, self.lens, self.nmask, self.quant = words, lens, nmask, quant
        self.min_len, self.max_len = int(min_len), int(max_len)
        return self

    @property
    def n_samples(self):
        return 0 if self.quant is None else int(self.quant.shape[1])


class CascadeResult:
    def __init__(self, pass_id, ref_id, pos, mm, pass_counts, engine, n_pass, packed=None):
        self.pass_id, self.ref_id, self.pos, self.mm = pass_id, ref_id, pos, mm
        self.packed = packed            # device int32 [n] (Engine.cascade_packed): then the four arrays are None
        self.pass_counts = pass_counts  # device int64 [2*n_pass]: processed, aligned
        self._engine = engine
        self.n_pass = n_pass
        self._stats = None
        self._run_id = engine._run_id() if engine is not None and hasattr(engine, "_run_id") else None

    @property
    def stats(self):
        """Synchronises; list of dicts per pass."""
        if self._stats is None:
            if self._run_id is not None and self._engine._run_id() != self._run_id:
                raise RuntimeError("the context has run another cascade since: read `stats` of a result before "
                                   "launching the next one (the counters and event times live in the context)")
            self._stats = self._engine._read_stats(self.n_pass)
        return self._stats

    def to_host(self):
        if self.packed is not None:   # (entries and offsets saturate: unpack_assignments)
            return unpack_assignments(self.packed.cpu().numpy())
        return (self.pass_id.cpu().numpy(), self.ref_id.cpu().numpy(), self.pos.cpu().numpy(),
                self.mm.cpu().numpy())


class Engine:
    def __init__(self, device=0):
        self._lib = _native.load()
        h = C.c_void_p()
        check(self._lib.mrg_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device_index = int(device)
        self.device = "cuda:%d" % int(device)
        self.libs = {}      # key -> lib id
        self.indexes = {}   # key -> FmIndex (kept alive; names for reports)
        self._ws = None
        n_cu = C.c_int32()
        hbm = C.c_uint64()
        arch = C.create_string_buffer(64)
        check(self._lib.mrg_ctx_device_info(self._h, C.byref(n_cu), C.byref(hbm), arch, 64))
        self.n_cu, self.hbm_bytes, self.arch = n_cu.value, hbm.value, arch.value.decode()
        # development aid: MIRGE_AMD_OPTS="key=value,key=value" sets context options on every new engine (A/B runs
        # of the whole test suite under another kernel variant)
        import os
        for kv in filter(None, os.environ.get("MIRGE_AMD_OPTS", "").split(",")):
            k, v = kv.split("=")
            self.set_option(k.strip(), int(v))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mrg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        check(self._lib.mrg_ctx_set_option(self._h, key.encode(), int(value)))

    def release_scratch(self):
        """Free the context's scratch arena (the collapse keeps 40 B per raw read otherwise)."""
        check(self._lib.mrg_ctx_release_scratch(self._h))

    def add_library(self, key, index, exact_dict=None):
        """Make a library resident.  exact_dict: give a LARGE library (> 4 Mbp) an exact-match dictionary
        too (16 B x 2..4 slots per base of HBM) -- by default the libraries the reference cascade searches
        without a seed mismatch (mRNA `-n 0`, runAnnotationPipeline.py:584/598; spike-in, :586): one 16-byte
        gather per read instead of a jump-table line and a suffix-array row line.  Small libraries always
        get theirs."""
        if exact_dict is None:
            exact_dict = key in EXACT_LIBS
        self.set_option("dict_max_bases", DICT_MAX_BASES if exact_dict else DICT_SMALL_BASES)
        lid = C.c_int32(-1)
        check(self._lib.mrg_ctx_add_library(self._h, index._h, C.byref(lid)))
        self.libs[key] = lid.value
        self.indexes[key] = index
        return lid.value

    def library_dict_stats(self, key):
        """(positions stored in the library's exact-match dictionary, HOME SLOTS whose chain overflowed -- every further
        position of such a home is left to the FM index: a count of homes, not of positions) -- mrg_ctx_library_stats."""
        out = (C.c_uint64 * 4)()
        check(self._lib.mrg_ctx_library_stats(self._h, self.libs[key], out))
        return int(out[0]), int(out[1])

    def check_tables(self, key):
        """Words in which a resident library's jump tables / row context / wide rows / seed buckets differ from the host
        functions' (None = the library has no such table) -- mrg_ctx_library_check_tables: 0 everywhere is the bar."""
        out = (C.c_uint64 * 4)()
        check(self._lib.mrg_ctx_library_check_tables(self._h, self.libs[key], self.indexes[key]._h, out))
        names = ("jump_tables", "row_context", "wide_rows", "seed_buckets")
        return {n: (None if int(v) == 2 ** 64 - 1 else int(v)) for n, v in zip(names, out)}

    # ------------------------------------------------------------------
    def mirge_passes(self, spike_in=False):
        """PassCfg array for the reference's cascade (runAnnotationPipeline.py:574-599)."""
        rows = MIRGE_PASS_TABLE[:10 if spike_in else 9]
        return self.make_passes([dict(lib=k, min_len=a, max_len=b, seed_len=s, max_mm_seed=ms,
                                      max_mm_total=mt, trim5=t5, trim3=t3, poly_t=pt)
                                 for (k, a, b, s, ms, mt, t5, t3, pt) in rows])

    def make_passes(self, rows):
        arr = (PassCfg * len(rows))()
        for i, r in enumerate(rows):
            lib = r["lib"]
            arr[i].lib = self.libs[lib] if isinstance(lib, str) else int(lib)
            arr[i].seed_len = int(r.get("seed_len", 28))
            arr[i].max_mm_seed = int(r.get("max_mm_seed", 0))
            arr[i].max_mm_total = int(r.get("max_mm_total", 2))
            arr[i].trim5 = int(r.get("trim5", 0))
            arr[i].trim3 = int(r.get("trim3", 0))
            arr[i].min_len = int(r.get("min_len", 0))
            arr[i].max_len = int(r.get("max_len", 255))
            arr[i].poly_t = int(r.get("poly_t", 0))
        return arr

    # ------------------------------------------------------------------
    def _stream_ptr(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, n):
        torch = _torch()
        need = C.c_uint64()
        check(self._lib.mrg_cascade_workspace_bytes(n, C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def cascade(self, reads, passes, out=None):
        """Run the cascade on a ReadSet; asynchronous on torch's current stream."""
        torch = _torch()
        n, n_pass = reads.n, len(passes)
        dev = self.device
        if out is None:
            out = (torch.empty(n, dtype=torch.int8, device=dev),
                   torch.empty(n, dtype=torch.int32, device=dev),
                   torch.empty(n, dtype=torch.int32, device=dev),
                   torch.empty(n, dtype=torch.uint8, device=dev),
                   torch.zeros(2 * n_pass, dtype=torch.int64, device=dev))
        pass_id, ref_id, pos, mm, pass_counts = out
        ws = self._workspace(n)
        # the length range of this batch (known on the host): passes whose window excludes it are
        # skipped; set before every run so that a hint never outlives its batch
        self.set_option("hint_min_len", reads.min_len)
        self.set_option("hint_max_len", reads.max_len)
        check(self._lib.mrg_cascade_run(
            self._h, reads.words.data_ptr(), reads.W, reads.lens.data_ptr(),
            reads.nmask.data_ptr() if reads.nmask is not None else None, n, passes, n_pass,
            pass_id.data_ptr(), ref_id.data_ptr(), pos.data_ptr(), mm.data_ptr(),
            pass_counts.data_ptr(), ws.data_ptr(), ws.numel(), self._stream_ptr()))
        return CascadeResult(pass_id, ref_id, pos, mm, pass_counts, self, n_pass)

    def cascade_long(self, seqs, passes, pass_counts=None, stats=None):
        """The cascade for reads of ANY length (mrg_cascade_run_long; RAP:543-554 offers every unannotated read,
        whatever its length, to every pass): `seqs` = ASCII reads the packed batches cannot describe (beyond 255 nt;
        any length works).  Returns host arrays (pass_id int8, ref_id int32, pos int32, mm uint8) and the per-pass
        dicts (processed, aligned, steps, candidates, lookups, ms) of these reads.  pass_counts: the device int64
        [2 n_pass] vector of the batch's own cascade -- the long reads' processed / aligned are ADDED to it;
        stats: the per-pass dicts of that cascade (CascadeResult.stats) -- added to in place.  Synchronises."""
        torch = _torch()
        from . import pack
        n, n_pass = len(seqs), len(passes)
        dev = self.device
        words, nmask, word_off, lens = pack.pack_ragged(list(seqs))

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev) if a is not None else None
        d_words = up(np.append(words, np.uint64(0)), np.int64)   # (one word of padding: an empty batch still has a buffer)
        d_nmask = up(None if nmask is None else np.append(nmask, np.uint64(0)), np.int64)
        d_off, d_lens = up(word_off, np.int64), up(lens, np.int32)
        out = (torch.empty(n, dtype=torch.int8, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
               torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))
        st = (PassStats * n_pass)()
        check(self._lib.mrg_cascade_run_long(
            self._h, d_words.data_ptr(), d_nmask.data_ptr() if d_nmask is not None else None, d_off.data_ptr(),
            d_lens.data_ptr(), n, passes, n_pass, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
            pass_counts.data_ptr() if pass_counts is not None else None, st, self._stream_ptr()))
        own = [dict(processed=int(s.processed), aligned=int(s.aligned), steps=int(s.steps), candidates=int(s.candidates),
                    lookups=int(s.lookups), ms=float(s.ms)) for s in st]
        if stats is not None:
            for mine, theirs in zip(own, stats):
                for k, v in mine.items():
                    theirs[k] += v
        return tuple(t.cpu().numpy() for t in out) + (own,)

    def pack_assignments(self, result, out=None):
        """The four assignment arrays of a CascadeResult as one int32 word per read (mrg_pack_assignments:
        pass + 1 in bits 28-31, mismatches 26-27, entry 8-25, offset 0-7, the last three saturating):
        4 bytes per read to move over PCIe instead of 10.  Asynchronous; returns the device tensor."""
        torch = _torch()
        n = int(result.pass_id.numel())
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        check(self._lib.mrg_pack_assignments(self._h, result.pass_id.data_ptr(), result.ref_id.data_ptr(), result.pos.data_ptr(),
                                             result.mm.data_ptr(), n, out.data_ptr(), self._stream_ptr()))
        return out

    def prepare(self, passes, words_per_read=1, min_len=0, max_len=255, has_n=False):
        """Build, ahead of the first timed run, whatever a cascade with these passes on batches of this
        shape derives lazily (the index over several small libraries searched with one policy, the
        anchor-pair tables of a large library for reads with short seed regions): a cascade over zero
        reads with the same length hints plans exactly the same launches."""
        torch = _torch()
        W = int(words_per_read)
        rs = ReadSet.from_device(torch.empty((W, 0), dtype=torch.int64, device=self.device),
                                 torch.empty(0, dtype=torch.uint8, device=self.device),
                                 torch.empty((W, 0), dtype=torch.int64, device=self.device) if has_n else None, None,
                                 int(min_len), int(max_len))
        self.cascade(rs, passes)
        torch.cuda.synchronize(self.device)

    def cascade_packed(self, reads, passes, out=None):
        """The cascade with ONE 4-byte word per read as its per-read output (mrg_cascade_run_packed: every
        kernel writes the packed assignment instead of pass_id / ref_id / pos / mm; SURVEY.md 8d's "4 B
        packed assignment out").  out = (packed int32 [n], pass_counts int64 [2 n_pass])."""
        torch = _torch()
        n, n_pass = reads.n, len(passes)
        if out is None:
            out = (torch.empty(n, dtype=torch.int32, device=self.device), torch.zeros(2 * n_pass, dtype=torch.int64, device=self.device))
        packed, pass_counts = out
        ws = self._workspace(n)
        self.set_option("hint_min_len", reads.min_len)
        self.set_option("hint_max_len", reads.max_len)
        check(self._lib.mrg_cascade_run_packed(
            self._h, reads.words.data_ptr(), reads.W, reads.lens.data_ptr(),
            reads.nmask.data_ptr() if reads.nmask is not None else None, n, passes, n_pass, packed.data_ptr(),
            pass_counts.data_ptr(), ws.data_ptr(), ws.numel(), self._stream_ptr()))
        return CascadeResult(None, None, None, None, pass_counts, self, n_pass, packed=packed)

    def _run_id(self):
        v = C.c_uint64()
        check(self._lib.mrg_cascade_run_id(self._h, C.byref(v)))
        return int(v.value)

    def _read_stats(self, n_pass):
        st = (PassStats * n_pass)()
        check(self._lib.mrg_cascade_stats(self._h, st, n_pass))
        return [dict(processed=int(s.processed), aligned=int(s.aligned), steps=int(s.steps),
                     candidates=int(s.candidates), lookups=int(s.lookups), ms=float(s.ms),
                     lds_bytes=int(s.lds_bytes), lds_mode=int(s.lds_mode), group=int(s.group),
                     n_launches=int(s.n_launches), kbits_log2=int(s.kbits_log2),
                     pair_anchor=int(s.pair_anchor), ms_rest=float(s.ms_rest), variant=int(s.variant))
                for s in st]

    def counts_len(self, n_mirna, n_samples, n_pass):
        ln = C.c_uint64()
        check(self._lib.mrg_tally_counts_len(n_mirna, n_samples, n_pass, C.byref(ln)))
        return int(ln.value)

    def tally(self, reads, result, n_mirna, canon_pass=CANON_PASS, isomir_pass=ISOMIR_PASS,
              counts=None):
        """summarize.py:34-66 on device; returns the fused int64 count vector (device)."""
        torch = _torch()
        if reads.quant is None:
            raise ValueError("ReadSet has no quant matrix")
        S = reads.n_samples
        ln = self.counts_len(n_mirna, S, result.n_pass)
        if counts is None:
            counts = torch.zeros(ln, dtype=torch.int64, device=self.device)
        if getattr(result, "packed", None) is not None:
            check(self._lib.mrg_tally_run_packed(
                self._h, result.packed.data_ptr(), reads.quant.data_ptr(), reads.n, S, n_mirna, result.n_pass, canon_pass,
                isomir_pass, counts.data_ptr(), self._stream_ptr()))
            return counts
        check(self._lib.mrg_tally_run(
            self._h, result.pass_id.data_ptr(), result.ref_id.data_ptr(), reads.quant.data_ptr(),
            reads.n, S, n_mirna, result.n_pass, canon_pass, isomir_pass, counts.data_ptr(),
            self._stream_ptr()))
        return counts

    # ---- torch-less multi-GPU (the C-ABI's own RCCL binding; mirge_amd.dist uses torch.distributed)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(_native.load().mrg_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        check(self._lib.mrg_comm_init(self._h, C.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def allreduce(self, counts):
        """In-place uint64 sum of a device tensor (int64 storage) over the ranks of comm_init."""
        check(self._lib.mrg_allreduce(self._h, counts.data_ptr(), counts.numel(), self._stream_ptr()))
        return counts

    def comm_destroy(self):
        check(self._lib.mrg_comm_destroy(self._h))

    EDIT_POSITIONS = 32

    def edit_counts_len(self, lib, n_samples, n_bins=None):
        ln = C.c_uint64()
        nb = self.indexes[lib].n_ref if n_bins is None else int(n_bins)
        check(self._lib.mrg_edit_counts_len(nb, int(n_samples), C.byref(ln)))
        return int(ln.value)

    def edit_tally(self, reads, result, lib="mirna", canon_pass=CANON_PASS, isomir_pass=ISOMIR_PASS, counts=None,
                   keep=None, remap=None, n_bins=None, from_base=0, to_base=2, isomir_trim5=1, flank5=2, flank3=6):
        """The per-read part of A2IEditing (writeDataToCSV.py:145-229) on device: per (miRNA bin, sample)
        count_true / seq_true / canonical and, per mature position, the counts of kept reads showing
        `to_base` where the mature sequence has `from_base` (A -> G by default).  keep: uint8 device
        tensor [n] or None; remap: int32 device tensor [entries] -> bin or None.  Returns the int64
        device vector laid out as mrg_edit_tally_run documents (split with split_edit_counts)."""
        torch = _torch()
        S = reads.n_samples
        nb = self.indexes[lib].n_ref if n_bins is None else int(n_bins)
        ln = self.edit_counts_len(lib, S, nb)
        if counts is None:
            counts = torch.zeros(ln, dtype=torch.int64, device=self.device)
        if getattr(result, "packed", None) is not None:
            check(self._lib.mrg_edit_tally_run_packed(
                self._h, reads.words.data_ptr(), reads.W, reads.lens.data_ptr(),
                reads.nmask.data_ptr() if reads.nmask is not None else None, result.packed.data_ptr(), reads.quant.data_ptr(),
                None if keep is None else keep.data_ptr(), None if remap is None else remap.data_ptr(), reads.n, S, nb,
                self.libs[lib], canon_pass, isomir_pass, isomir_trim5, flank5, flank3, from_base, to_base,
                counts.data_ptr(), self._stream_ptr()))
            return counts
        check(self._lib.mrg_edit_tally_run(
            self._h, reads.words.data_ptr(), reads.W, reads.lens.data_ptr(),
            reads.nmask.data_ptr() if reads.nmask is not None else None, result.pass_id.data_ptr(),
            result.ref_id.data_ptr(), result.pos.data_ptr(), reads.quant.data_ptr(),
            None if keep is None else keep.data_ptr(), None if remap is None else remap.data_ptr(), reads.n, S, nb,
            self.libs[lib], canon_pass, isomir_pass, isomir_trim5, flank5, flank3, from_base, to_base,
            counts.data_ptr(), self._stream_ptr()))
        return counts

    def collapse(self, words, lens, nmask=None, sample=None, n_samples=1, max_len=0, out=None):
        """quantReads (QNT:3-24) on device tensors: raw reads (words int64 [W, n], lens uint8 [n],
        nmask or None, sample int16 [n] or None) -> (ReadSet of the unique reads with their
        per-sample counts, all still in HBM and ordered by (length, bases); read-length histogram
        int64 [256, S]).  Synchronises (the number of uniques comes back to the host).
        out = (u_words [W, n], u_lens [n], u_nmask or None, quant [n, S]): caller-owned output
        buffers (a pipeline that collapses batch after batch keeps one set instead of asking the
        allocator for n-sized arrays every time)."""
        torch = _torch()
        dev = self.device
        W, n = int(words.shape[0]), int(words.shape[1])
        cap = max(n, 1)
        if out is not None:
            u_words, u_lens, u_nmask, quant = out
            if tuple(u_words.shape) != (W, cap) or u_lens.numel() != cap or tuple(quant.shape) != (cap, n_samples) or \
                    (nmask is not None and u_nmask is None):
                raise ValueError("collapse: output buffers do not match the input shape")
        else:
            u_words = torch.empty((W, cap), dtype=torch.int64, device=dev)
            u_lens = torch.empty(cap, dtype=torch.uint8, device=dev)
            u_nmask = None if nmask is None else torch.empty((W, cap), dtype=torch.int64, device=dev)
            quant = torch.empty((cap, n_samples), dtype=torch.int32, device=dev)
        hist = torch.zeros((256, n_samples), dtype=torch.int64, device=dev)
        n_unique = C.c_uint64(0)
        check(self._lib.mrg_collapse_run(
            self._h, words.data_ptr(), W, lens.data_ptr(), None if nmask is None else nmask.data_ptr(),
            None if sample is None or n_samples == 1 else sample.data_ptr(), n, n_samples, int(max_len), cap,
            u_words.data_ptr(), u_lens.data_ptr(), None if u_nmask is None else u_nmask.data_ptr(),
            quant.data_ptr(), hist.data_ptr(), C.byref(n_unique), self._stream_ptr()))
        U = int(n_unique.value)
        # the SoA stride of the cascade is the array's own row length: compact views
        rs = ReadSet.from_device(u_words[:, :U].contiguous() if U != cap else u_words, u_lens[:U],
                                 None if u_nmask is None else (u_nmask[:, :U].contiguous() if U != cap else u_nmask),
                                 quant[:U], 0, int(max_len) if max_len else 255)
        return rs, hist

    def expand_compact(self, bits, runs, quant8=None, esc=None, n_samples=1, out=None):
        """mrg_expand_compact: the compact wire form of a host-resident collapsed read set
        (pack.compact_read_set; bits int64 [n_words] and quant8 uint8 [n, S] / esc int32 [k, 2] already on
        the device, runs a host array of (length, count)) -> ReadSet in HBM.  Asynchronous.
        out = (words int64 [1, n], lens uint8 [n], quant int32 [n, S] or None): caller-owned buffers."""
        torch = _torch()
        dev = self.device
        runs = np.ascontiguousarray(np.asarray(runs, dtype=np.uint32).reshape(-1, 2))
        n = int(runs[:, 1].sum())
        if out is not None:
            words, lens, quant = out
        else:
            words = torch.empty((1, max(n, 1)), dtype=torch.int64, device=dev)[:, :n]
            lens = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)[:n]
            quant = None if quant8 is None else torch.empty((max(n, 1), n_samples), dtype=torch.int32, device=dev)[:n]
        check(self._lib.mrg_expand_compact(
            self._h, bits.data_ptr(), int(bits.numel()), runs.ctypes.data, runs.shape[0], None if quant8 is None else quant8.data_ptr(),
            None if esc is None or esc.shape[0] == 0 else esc.data_ptr(), 0 if esc is None else int(esc.shape[0]), n, n_samples,
            words.data_ptr(), lens.data_ptr(), None if quant8 is None else quant.data_ptr(), self._stream_ptr()))
        live = runs[:, 0][runs[:, 1] > 0]
        return ReadSet.from_device(words, lens, None, None if quant8 is None else quant, int(live.min()) if n else 0,
                                   int(live.max()) if n else 255)

    def count_best(self, reads, lib, seed_len=28, max_mm_seed=1, max_mm_total=2):
        """Best stratum of every read of a ReadSet against one library, forward strand:
        (fewest mismatches or 255, alignments reaching it, saturating) as host uint8 arrays.
        The -ai genome filters of writeDataToCSV.py:1263/:1488 (see mirge_amd.a2i)."""
        torch = _torch()
        mm = torch.empty(reads.n, dtype=torch.uint8, device=self.device)
        cnt = torch.empty(reads.n, dtype=torch.uint8, device=self.device)
        lid = self.libs[lib] if isinstance(lib, str) else int(lib)
        check(self._lib.mrg_count_best(
            self._h, reads.words.data_ptr(), reads.W, reads.lens.data_ptr(),
            reads.nmask.data_ptr() if reads.nmask is not None else None, reads.n, lid, int(seed_len),
            int(max_mm_seed), int(max_mm_total), mm.data_ptr(), cnt.data_ptr(), self._stream_ptr()))
        return mm.cpu().numpy(), cnt.cpu().numpy()

    def list_best(self, reads, lib, seed_len=28, max_mm_seed=0, max_mm_total=0):
        """`-a --best --strata` (RAP:577-599; parseAlignment3 RAP:41-52): every alignment of each
        read's best stratum.  Returns host arrays (best_mm[n], offsets[n+1], ref[total],
        pos[total]); read r owns ref/pos[offsets[r]:offsets[r+1]], sorted by (entry, offset)."""
        torch = _torch()
        n = reads.n
        mm = torch.empty(n, dtype=torch.uint8, device=self.device)
        off = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        lid = self.libs[lib] if isinstance(lib, str) else int(lib)
        nm = reads.nmask.data_ptr() if reads.nmask is not None else None
        total = C.c_uint64(0)
        check(self._lib.mrg_list_best_count(
            self._h, reads.words.data_ptr(), reads.W, reads.lens.data_ptr(), nm, n, lid, int(seed_len),
            int(max_mm_seed), int(max_mm_total), mm.data_ptr(), off.data_ptr(), C.byref(total),
            self._stream_ptr()))
        t = int(total.value)
        ref = torch.empty(max(t, 1), dtype=torch.int32, device=self.device)
        pos = torch.empty(max(t, 1), dtype=torch.int32, device=self.device)
        check(self._lib.mrg_list_best_fill(
            self._h, reads.words.data_ptr(), reads.W, reads.lens.data_ptr(), nm, n, lid, int(seed_len),
            int(max_mm_seed), int(max_mm_total), mm.data_ptr(), off.data_ptr(), t, ref.data_ptr(),
            pos.data_ptr(), self._stream_ptr()))
        off_h = off.cpu().numpy()
        ref_h, pos_h = ref.cpu().numpy()[:t], pos.cpu().numpy()[:t]
        # canonical order inside each read: (entry, offset)
        owner = np.repeat(np.arange(n, dtype=np.int64), np.diff(off_h))
        order = np.lexsort((pos_h, ref_h, owner))
        return mm.cpu().numpy(), off_h, ref_h[order], pos_h[order]

    # ------------------------------------------------------------------
    def annotate_host(self, words, lens, nmask, passes, quant=None, n_mirna=0,
                      canon_pass=CANON_PASS, isomir_pass=ISOMIR_PASS):
        """Host-buffer path of the C-ABI (mrg_annotate_host): numpy in, numpy out."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        W, n = words.shape
        lens = np.ascontiguousarray(lens, dtype=np.uint8)
        nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
        n_pass = len(passes)
        pass_id = np.empty(n, dtype=np.int8)
        ref_id = np.empty(n, dtype=np.int32)
        pos = np.empty(n, dtype=np.int32)
        mm = np.empty(n, dtype=np.uint8)
        st = (PassStats * n_pass)()
        counts = None
        q = None
        S = 0
        if quant is not None:
            q = np.ascontiguousarray(quant, dtype=np.uint32)
            if q.ndim == 1:
                q = q[:, None]
            S = q.shape[1]
            counts = np.zeros(self.counts_len(n_mirna, S, n_pass), dtype=np.uint64)
        check(self._lib.mrg_annotate_host(
            self._h, words.ctypes.data, W, lens.ctypes.data, None if nm is None else nm.ctypes.data,
            n, passes, n_pass, pass_id.ctypes.data, ref_id.ctypes.data, pos.ctypes.data,
            mm.ctypes.data, st, None if q is None else q.ctypes.data, S, n_mirna, canon_pass,
            isomir_pass, None if counts is None else counts.ctypes.data))
        stats = [dict(processed=int(s.processed), aligned=int(s.aligned), steps=int(s.steps),
                      candidates=int(s.candidates), lookups=int(s.lookups), ms=float(s.ms),
                     lds_bytes=int(s.lds_bytes), lds_mode=int(s.lds_mode), group=int(s.group),
                      n_launches=int(s.n_launches), kbits_log2=int(s.kbits_log2),
                     pair_anchor=int(s.pair_anchor), ms_rest=float(s.ms_rest), variant=int(s.variant))
                 for s in st]
        return dict(pass_id=pass_id, ref_id=ref_id, pos=pos, mm=mm, stats=stats, counts=counts)


PACKED_REF_SAT, PACKED_POS_SAT = 0x3FFFF, 0xFF


def unpack_assignments(packed):
    """Host side of Engine.pack_assignments: (pass_id int8, ref_id int32, pos int32, mm uint8) from the
    packed words (numpy); ref_id == PACKED_REF_SAT / pos == PACKED_POS_SAT / mm == 3 mean "at least"."""
    w = np.asarray(packed).view(np.uint32)
    code = (w >> 28).astype(np.int16)
    pass_id = (code - 1).astype(np.int8)
    none = code == 0
    ref = ((w >> 8) & 0x3FFFF).astype(np.int32)
    pos = (w & 0xFF).astype(np.int32)
    mm = ((w >> 26) & 3).astype(np.uint8)
    ref[none] = -1
    pos[none] = -1
    mm[none] = 0
    return pass_id, ref, pos, mm


def split_counts(counts, n_mirna, n_samples, n_pass):
    """Views into the fused count vector (layout of mrg_tally_run)."""
    M, S = n_mirna, n_samples
    c = np.asarray(counts).astype(np.int64)
    quant = c[:M * S].reshape(M, S)
    iscan = c[M * S:2 * M * S].reshape(M, S)
    cat = c[2 * M * S:2 * M * S + (n_pass + 1) * S].reshape(n_pass + 1, S)
    uniq = c[2 * M * S + (n_pass + 1) * S:2 * M * S + (n_pass + 2) * S]
    return quant, iscan, cat, uniq


def split_edit_counts(counts, n_bins, n_samples):
    """Views into the vector of mrg_edit_tally_run: (totals [bins, S, 3] = count_true, seq_true,
    canonical; positions [bins, 32, S])."""
    c = np.asarray(counts).astype(np.int64)
    k = n_bins * n_samples * 3
    return c[:k].reshape(n_bins, n_samples, 3), c[k:].reshape(n_bins, Engine.EDIT_POSITIONS, n_samples)
