"""ctypes bindings of oracle/_build/liboracle.so (bowtie_model.c + fm_cpu.c + index_check.c).

TEST INFRASTRUCTURE, see oracle/__init__.py.  `build()` compiles the C sources
with gcc; nothing here touches a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/*.c -> oracle/_build/liboracle.so (gcc -O2 -fopenmp)."""
    srcs = [os.path.join(_HERE, f) for f in ("bowtie_model.c", "fm_cpu.c", "edit_tally.c", "index_check.c", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_align_batch.restype = None
        _lib.orc_align_all_best.restype = C.c_int
        _lib.orc_run_cascade.restype = None
        _lib.orc_tally.restype = None
    return _lib


class Library:
    """One reference library as the model sees it: names + ASCII sequences."""

    def __init__(self, names, seqs):
        self.names = list(names)
        self.seqs = [s.upper() for s in seqs]
        self.concat = "".join(self.seqs).encode("ascii")
        off = np.zeros(len(self.seqs) + 1, dtype=np.uint32)
        np.cumsum([len(s) for s in self.seqs], out=off[1:])
        self.off = off

    @classmethod
    def from_fasta(cls, path):
        names, seqs = [], []
        with open(path) as fh:
            for line in fh:
                line = line.strip()
                if not line:
                    continue
                if line[0] == ">":
                    names.append(line[1:].split()[0])
                    seqs.append([])
                else:
                    seqs[-1].append(line)
        return cls(names, ["".join(s) for s in seqs])


def align_batch(library, reads, seed_len, max_mm_seed, max_mm_total):
    """Exhaustive-scan alignment of ASCII reads; returns (ref, pos, mm) int32
    arrays with -1 for unaligned reads."""
    n = len(reads)
    out_ref = np.full(n, -1, dtype=np.int32)
    out_pos = np.full(n, -1, dtype=np.int32)
    out_mm = np.full(n, -1, dtype=np.int32)
    if n == 0:
        return out_ref, out_pos, out_mm
    blob = "".join(reads).upper().encode("ascii")
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    lib().orc_align_batch(
        C.c_char_p(library.concat), library.off.ctypes.data_as(C.c_void_p),
        C.c_uint32(len(library.names)), C.c_char_p(blob), off.ctypes.data_as(C.c_void_p),
        C.c_uint64(n), C.c_int(seed_len), C.c_int(max_mm_seed), C.c_int(max_mm_total),
        out_ref.ctypes.data_as(C.c_void_p), out_pos.ctypes.data_as(C.c_void_p),
        out_mm.ctypes.data_as(C.c_void_p))
    return out_ref, out_pos, out_mm


def align_all_best(library, read, seed_len, max_mm_seed, max_mm_total, cap=4096):
    """Every alignment of the best stratum, in (entry, offset) order."""
    refs = np.zeros(cap, dtype=np.int32)
    poss = np.zeros(cap, dtype=np.int32)
    mm = C.c_int32(-1)
    r = read.upper().encode("ascii")
    k = lib().orc_align_all_best(
        C.c_char_p(library.concat), library.off.ctypes.data_as(C.c_void_p),
        C.c_uint32(len(library.names)), C.c_char_p(r), C.c_int(len(r)), C.c_int(seed_len),
        C.c_int(max_mm_seed), C.c_int(max_mm_total), refs.ctypes.data_as(C.c_void_p),
        poss.ctypes.data_as(C.c_void_p), C.c_int(cap), C.byref(mm))
    k = min(k, cap)
    return [(int(refs[i]), int(poss[i])) for i in range(k)], int(mm.value)


def best_stratum(library, read, seed_len, max_mm_seed, max_mm_total):
    """(fewest mismatches, number of alignments reaching it) on the forward strand;
    (255, 0) when nothing aligns."""
    mm = C.c_int32(255)
    cnt = C.c_int64(0)
    r = read.upper().encode("ascii")
    f = lib().orc_best_stratum
    f.restype = None
    f(C.c_char_p(library.concat), library.off.ctypes.data_as(C.c_void_p), C.c_uint32(len(library.names)),
      C.c_char_p(r), C.c_int(len(r)), C.c_int(seed_len), C.c_int(max_mm_seed), C.c_int(max_mm_total),
      C.byref(mm), C.byref(cnt))
    return int(mm.value), int(cnt.value)


def list_valid(library, read, seed_len, max_mm_seed, max_mm_total, cap=65536):
    """[(entry, offset, mismatches)] of every valid forward-strand alignment (`bowtie -a`)."""
    refs = np.zeros(cap, dtype=np.int32)
    poss = np.zeros(cap, dtype=np.int32)
    mms = np.zeros(cap, dtype=np.int32)
    r = read.upper().encode("ascii")
    f = lib().orc_list_valid
    f.restype = C.c_int
    k = f(C.c_char_p(library.concat), library.off.ctypes.data_as(C.c_void_p), C.c_uint32(len(library.names)),
          C.c_char_p(r), C.c_int(len(r)), C.c_int(seed_len), C.c_int(max_mm_seed), C.c_int(max_mm_total),
          refs.ctypes.data_as(C.c_void_p), poss.ctypes.data_as(C.c_void_p), mms.ctypes.data_as(C.c_void_p),
          C.c_int(cap))
    return [(int(refs[i]), int(poss[i]), int(mms[i])) for i in range(min(k, cap))]


_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s):
    return "".join(_COMP.get(ch, "N") for ch in reversed(s))


class ScanGenome:
    """The two genome bowtie runs of the -ai path (writeDataToCSV.py:1263, :1488) answered by
    exhaustive scan: both strands, 3' 2 nt trimmed, -n 1 / -n 0."""

    def __init__(self, library):
        self.library = library

    def _best(self, read, max_mm_seed):
        t = read[:-2]
        a = best_stratum(self.library, t, 28, max_mm_seed, 2)
        b = best_stratum(self.library, revcomp(t), 28, max_mm_seed, 2)
        if a[0] < b[0]:
            return a
        if b[0] < a[0]:
            return b
        return (a[0], a[1] + b[1])

    def unique_best(self, reads):
        out = set()
        for r in set(reads):
            mm, cnt = self._best(r, 1)
            if mm < 255 and cnt == 1:
                out.add(r)
        return out

    def exact_hit(self, reads):
        return {r for r in set(reads) if self._best(r, 0)[0] < 255}


# ---------------------------------------------------------------------------
# fm_cpu.c: the host-core port of the seed-and-verify matcher
# ---------------------------------------------------------------------------
class _OrcLib(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("blocks", "super", "text", "sa", "ftab", "seg_start", "seg_ref", "seg_off",
                 "chunk_seg", "kbits")] + \
               [("n", C.c_uint32), ("primary", C.c_uint32), ("ftab_ks", C.c_uint8 * 4)]


class _OrcPass(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("lib", "seed_len", "max_mm_seed", "max_mm_total", "trim5", "trim3", "min_len",
                 "max_len", "poly_t", "reserved", "pair_anchor")]


def fm_cascade(lib_views, passes, reads, lens, nmask=None, wstop=0, threads=None,
               want_steps=False, ftab=False, kmer_filter=True):
    """Run the CPU port.

    lib_views: list of dicts with numpy arrays blocks/super/text/sa/seg_start/
               seg_ref/seg_off/chunk_seg and ints n, primary (as mirge_amd index views).
    passes   : list of dicts with the mrg_pass_cfg fields; optional `kbits_log2` (what
               mrg_pass_stats reports: 18/0 = the library's full 9-mer bitmap when it has one,
               13..17 = folded as a fused GPU launch stages it).
    reads    : uint64 [W, n] SoA words; lens uint8 [n]; nmask like reads or None.
    wstop / ftab / kmer_filter: the same search shortcuts the GPU context options select.
    Returns dict(pass_id, ref_id, pos, mm, stats[n_pass,5] = processed, aligned, steps,
    candidates, lookups; steps_per_read|None).
    """
    reads = np.ascontiguousarray(reads, dtype=np.uint64)
    W, n = reads.shape
    lens = np.ascontiguousarray(lens, dtype=np.uint8)
    keep = []
    libs = (_OrcLib * len(lib_views))()
    for i, v in enumerate(lib_views):
        for k in ("blocks", "super", "text", "sa", "ftab", "seg_start", "seg_ref", "seg_off",
                  "chunk_seg"):
            a = np.ascontiguousarray(v[k], dtype=np.uint64 if k == "sa" else np.uint32)
            keep.append(a)
            setattr(libs[i], k, a.ctypes.data)
        kb = v.get("kbits") if kmer_filter else None
        if kb is not None:
            kb = np.ascontiguousarray(kb, dtype=np.uint32)
            keep.append(kb)
            libs[i].kbits = kb.ctypes.data
        libs[i].n = int(v["n"])
        libs[i].primary = int(v["primary"])
        for t, k in enumerate(v["ftab_ks"]):
            libs[i].ftab_ks[t] = int(k)
    ps = (_OrcPass * len(passes))()
    for i, p in enumerate(passes):
        for k, _ in _OrcPass._fields_:
            setattr(ps[i], k, int(p.get(k, 0)))
        lg = int(p.get("kbits_log2", 0))
        ps[i].reserved = lg if (13 <= lg < 18 or lg == 255) else 0   # 255 = this pass ran unfiltered
    pass_id = np.empty(n, dtype=np.int8)
    ref_id = np.empty(n, dtype=np.int32)
    pos = np.empty(n, dtype=np.int32)
    mm = np.empty(n, dtype=np.uint8)
    stats = np.zeros((len(passes), 5), dtype=np.uint64)
    steps = np.zeros(n, dtype=np.uint32) if want_steps else None
    nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
    old = os.environ.get("OMP_NUM_THREADS")
    if threads:
        os.environ["OMP_NUM_THREADS"] = str(threads)
        try:
            C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
        except OSError:
            pass
    lib().orc_run_cascade(
        libs, ps, C.c_int(len(passes)), reads.ctypes.data_as(C.c_void_p), C.c_int(W),
        lens.ctypes.data_as(C.c_void_p), None if nm is None else nm.ctypes.data_as(C.c_void_p),
        C.c_uint64(n), C.c_uint32(wstop), C.c_int(1 if ftab else 0),
        pass_id.ctypes.data_as(C.c_void_p),
        ref_id.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
        mm.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p),
        None if steps is None else steps.ctypes.data_as(C.c_void_p))
    if threads and old is not None:
        os.environ["OMP_NUM_THREADS"] = old
    return dict(pass_id=pass_id, ref_id=ref_id, pos=pos, mm=mm, stats=stats, steps_per_read=steps)


_CHECKS = {1: "segment tables", 2: "text", 3: "chunk_seg", 4: "suffix array is not a permutation", 5: "suffix array order",
           6: "suffix-array row fields", 7: "BWT blocks / superblocks / primary", 8: "jump table", 9: "9-mer bitmap"}


def check_index(view, seqs, threads=None):
    """oracle/index_check.c: every array of an index view (mirge_amd FmIndex.view(): what the GPU uploads and
    what fm_cascade reads) against the library's strings, by definition -- text and segments, the suffix array a
    permutation in ascending suffix order with the right row fields, the BWT blocks it implies, every jump
    table, the 9-mer bitmap.  Returns dict(rows, jump_entries); raises AssertionError naming the array and the
    place of the first violation."""
    keep = []
    l = _OrcLib()
    for k in ("blocks", "super", "text", "sa", "ftab", "seg_start", "seg_ref", "seg_off", "chunk_seg"):
        a = np.ascontiguousarray(view[k], dtype=np.uint64 if k == "sa" else np.uint32)
        keep.append(a)
        setattr(l, k, a.ctypes.data)
    kb = view.get("kbits")
    if kb is not None:
        kb = np.ascontiguousarray(kb, dtype=np.uint32)
        keep.append(kb)
        l.kbits = kb.ctypes.data
    l.n = int(view["n"])
    l.primary = int(view["primary"])
    for t, k in enumerate(view["ftab_ks"]):
        l.ftab_ks[t] = int(k)
    concat = "".join(seqs).encode("ascii")
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    report = np.zeros(8, dtype=np.uint64)
    if threads:
        try:
            C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
        except OSError:
            pass
    f = lib().orc_check_index
    f.restype = C.c_int
    rc = f(C.byref(l), concat, off.ctypes.data_as(C.c_void_p), C.c_uint32(len(seqs)),
           C.c_uint32(len(view["seg_ref"])), report.ctypes.data_as(C.c_void_p))
    assert rc == 0, "index check failed: %s at %d (got %d, want %d)" % (_CHECKS.get(rc, rc), int(report[1]) & ((1 << 56) - 1),
                                                                        int(report[2]), int(report[3]))
    return dict(rows=int(report[4]), jump_entries=int(report[6]))


def tally(pass_id, ref_id, quant, n_mirna, n_pass, canon_pass, isomir_pass):
    quant = np.ascontiguousarray(quant, dtype=np.uint32)
    n, S = quant.shape
    counts = np.zeros(2 * n_mirna * S + (n_pass + 1) * S + S, dtype=np.uint64)
    lib().orc_tally(
        np.ascontiguousarray(pass_id, dtype=np.int8).ctypes.data_as(C.c_void_p),
        np.ascontiguousarray(ref_id, dtype=np.int32).ctypes.data_as(C.c_void_p),
        quant.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(S), C.c_uint32(n_mirna),
        C.c_uint32(n_pass), C.c_int(canon_pass), C.c_int(isomir_pass),
        counts.ctypes.data_as(C.c_void_p))
    return counts


def edit_tally(mirna_index, pass_id, ref_id, pos, words, lens, quant, canon_pass=0, isomir_pass=8, nmask=None,
               keep=None, remap=None, n_bins=None, from_base=0, to_base=2, isomir_trim5=1, flank5=2, flank3=6):
    """oracle/edit_tally.c: the A-to-I position tally by dash-padded strings.  `mirna_index` is a
    mirge_amd FmIndex (only its packed text and entry starts are read).  Returns uint64
    [n_bins * S * 3 + n_bins * 32 * S] laid out as mrg_edit_tally_run documents."""
    v = mirna_index.view()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    W, n = words.shape
    quant = np.ascontiguousarray(quant, dtype=np.uint32)
    if quant.ndim == 1:
        quant = quant[:, None]
    S = quant.shape[1]
    nb = mirna_index.n_ref if n_bins is None else int(n_bins)
    counts = np.zeros(nb * S * 35, dtype=np.uint64)
    text = np.ascontiguousarray(v["text"], dtype=np.uint32)
    seg_start = np.ascontiguousarray(v["seg_start"], dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.uint8)
    nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
    kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    rm = None if remap is None else np.ascontiguousarray(remap, dtype=np.uint32)
    a_pass = np.ascontiguousarray(pass_id, dtype=np.int8)
    a_ref = np.ascontiguousarray(ref_id, dtype=np.int32)
    a_pos = np.ascontiguousarray(pos, dtype=np.int32)
    f = lib().orc_edit_tally
    f.restype = None
    f(text.ctypes.data_as(C.c_void_p), seg_start.ctypes.data_as(C.c_void_p), words.ctypes.data_as(C.c_void_p),
      C.c_int(W), lens.ctypes.data_as(C.c_void_p), None if nm is None else nm.ctypes.data_as(C.c_void_p),
      a_pass.ctypes.data_as(C.c_void_p), a_ref.ctypes.data_as(C.c_void_p), a_pos.ctypes.data_as(C.c_void_p),
      quant.ctypes.data_as(C.c_void_p), None if kp is None else kp.ctypes.data_as(C.c_void_p),
      None if rm is None else rm.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(S), C.c_uint32(nb),
      C.c_int(canon_pass), C.c_int(isomir_pass), C.c_int(isomir_trim5), C.c_int(flank5), C.c_int(flank3),
      C.c_int(from_base), C.c_int(to_base), counts.ctypes.data_as(C.c_void_p))
    return counts
