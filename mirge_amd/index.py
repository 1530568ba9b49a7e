"""FM index of one reference library (host side).

Replaces what the reference gets from `bowtie-build` (offline, the `.ebwt`
files MAIN:262-281 checks for) and from `bowtie-inspect` at run time
(summarize.py:6-9 for the miRNA names that define the histogram bins;
runAnnotationPipeline.py:610-611,630 for name -> sequence dictionaries).
"""
import ctypes as C
import os

import numpy as np

from . import _native
from ._native import check


class FmIndex:
    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self._lib = _native.load()
        info = _native.IndexInfo()
        check(self._lib.mrg_index_get_info(self._h, C.byref(info)))
        self.info = info
        self._names = None

    # ---- construction --------------------------------------------------
    @classmethod
    def build(cls, names, seqs):
        lib = _native.load()
        n = len(names)
        if n != len(seqs):
            raise ValueError("names and seqs differ in length")
        arr_n = (C.c_char_p * n)(*[s.encode("ascii") for s in names])
        arr_s = (C.c_char_p * n)(*[s.encode("ascii") for s in seqs])
        h = C.c_void_p()
        check(lib.mrg_index_build(arr_n, arr_s, n, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_fasta(cls, path):
        lib = _native.load()
        h = C.c_void_p()
        check(lib.mrg_index_build_fasta(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_ebwt(cls, prefix):
        """The reference's own library file `<prefix>.1.ebwt` (bowtie 1, MAIN:262-281): names and
        sequences are read back out of the BWT (bowtie-inspect's job, SUM:6 / RAP:610-611) and
        indexed.  Restated without a bowtie-built sample: validated by round trip only."""
        lib = _native.load()
        h = C.c_void_p()
        check(lib.mrg_index_build_ebwt(os.fsencode(prefix), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path):
        lib = _native.load()
        h = C.c_void_p()
        check(lib.mrg_index_load(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    @classmethod
    def open_prefix(cls, prefix, cache=False):
        """Resolve a bowtie-style index prefix as the reference passes it around (MAIN:269-281):
        `<prefix>.mrgfm` if built (and not older than its source), else build from `<prefix>.fa`
        (what `bowtie-inspect <prefix>` would print) or from the reference's own `<prefix>.1.ebwt`
        (mirge_amd.ebwt reads names and sequences back out of it).  cache: save what had to be
        built as `<prefix>.mrgfm` (best effort: a read-only library directory is not an error)."""
        src = next((prefix + e for e in (".fa", ".fasta", ".1.ebwt") if os.path.isfile(prefix + e)), None)
        built = prefix + ".mrgfm"
        if os.path.isfile(built) and (src is None or os.path.getmtime(built) >= os.path.getmtime(src)):
            return cls.load(built)
        if src is None:
            raise FileNotFoundError(
                "no %s.mrgfm, %s.fa or %s.1.ebwt: build one with `python -m mirge_amd.build_index`"
                % (prefix, prefix, prefix))
        if src.endswith(".1.ebwt"):
            ix = cls.from_ebwt(prefix)
        else:
            ix = cls.from_fasta(src)
        # (an index read back from a `.1.ebwt` is not cached: that reader is validated by round trip
        # only, and a wrong guess must not outlive the run next to the library)
        if cache and not src.endswith(".1.ebwt"):
            tmp = "%s.%d.tmp" % (built, os.getpid())
            try:
                ix.save(tmp)
                os.replace(tmp, built)
            except Exception:
                try:
                    os.remove(tmp)
                except OSError:
                    pass
        return ix

    @classmethod
    def open_prefix_parts(cls, prefix):
        """A library too large for one 32-bit index (the genome, MAIN:277): `<prefix>.mrgfm` /
        `<prefix>.fa` if present, else every `<prefix>.partNNN.mrgfm` written by
        `python -m mirge_amd.build_index --max-bases`."""
        import glob
        if any(os.path.isfile(prefix + e) for e in (".mrgfm", ".fa", ".fasta")):
            return [cls.open_prefix(prefix)]
        parts = sorted(glob.glob(glob.escape(prefix) + ".part[0-9]*.mrgfm"))
        if not parts:
            raise FileNotFoundError("no %s.mrgfm, %s.fa or %s.partNNN.mrgfm" % (prefix, prefix, prefix))
        return [cls.load(p) for p in parts]

    def save(self, path):
        check(self._lib.mrg_index_save(self._h, os.fsencode(path)))

    def __del__(self):
        try:
            if self._h:
                self._lib.mrg_index_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- bowtie-inspect equivalents -------------------------------------
    @property
    def n_ref(self):
        return int(self.info.n_ref)

    @property
    def names(self):
        """`bowtie-inspect -n` (summarize.py:6-9)."""
        if self._names is None:
            out = []
            p = C.c_char_p()
            for i in range(self.n_ref):
                check(self._lib.mrg_index_name(self._h, i, C.byref(p)))
                out.append(p.value.decode("ascii"))
            self._names = out
        return self._names

    def sequence(self, i):
        """`bowtie-inspect`: sequence of entry i."""
        ln = C.c_uint32()
        check(self._lib.mrg_index_seq(self._h, i, None, 0, C.byref(ln)))
        buf = C.create_string_buffer(ln.value + 1)
        check(self._lib.mrg_index_seq(self._h, i, buf, ln.value + 1, C.byref(ln)))
        return buf.value.decode("ascii")

    def name_seq_dict(self):
        return {n: self.sequence(i) for i, n in enumerate(self.names)}

    # ---- raw arrays (tests, the oracle's CPU port) -----------------------
    def view(self):
        v = _native.IndexView()
        check(self._lib.mrg_index_get_view(self._h, C.byref(v)))
        inf = self.info

        def arr(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,))

        return dict(
            blocks=arr(v.blocks, inf.n_blocks * 4), super=arr(v.super, inf.n_super * 4),
            text=arr(v.text, inf.text_words),
            sa=arr(v.sa, inf.n_bases + 1), ftab=arr(v.ftab, sum((1 << (2 * k)) + 1 for k in inf.ftab_ks if k)),
            ftab_ks=[int(k) for k in inf.ftab_ks], seg_start=arr(v.seg_start, inf.n_seg + 1),
            seg_ref=arr(v.seg_ref, max(inf.n_seg, 1))[:inf.n_seg],
            seg_off=arr(v.seg_off, max(inf.n_seg, 1))[:inf.n_seg],
            chunk_seg=arr(v.chunk_seg, (inf.n_bases >> 5) + 2),
            ctx=arr(v.ctx, inf.n_bases + 1) if v.ctx else None,
            kbits=arr(v.kbits, (1 << 18) // 32) if v.kbits else None,
            n=int(inf.n_bases), primary=int(inf.primary), C=[int(c) for c in inf.C],
            _owner=self)

    def exact_dict(self, key_bases=16):
        """The library's exact-match dictionary (mrg_index_get_dict): dict(slots uint64 [2^log2, 2],
        log2_slots, key_bases, n_keys, n_overflow); built on first use, valid while the index lives."""
        v = _native.DictView()
        check(self._lib.mrg_index_get_dict(self._h, int(key_bases), C.byref(v)))
        n = 1 << v.log2_slots
        return dict(slots=np.ctypeslib.as_array(v.slots, shape=(n, 2)), log2_slots=int(v.log2_slots),
                    key_bases=int(v.key_bases), n_keys=int(v.n_keys), n_overflow=int(v.n_overflow), _owner=self)
