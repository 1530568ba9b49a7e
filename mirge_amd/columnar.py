"""Columnar driver: the annotate path without the reference's dict of dicts.

The reference keeps every unique read in `seqDic[seq] = {'quant', 'annot', 'length'}` (QNT:13-15)
and walks that dict in runAnnotationPipeline, summarize and writeDataToCSV; at the 10^7-10^8
unique reads of BASELINE configs 3-5 that does not fit a Python process.  Here the per-read state
stays in arrays (HBM, then numpy) and only the M-sized tables and the SUBSETS of reads the
downstream consumers actually look at are turned into the reference's shapes:

  annotate_columns     cascade + tally on packed reads; pass_id / ref_id / pos / mm + counts
  tables_from_columns  mirDic / quantStats via summarize_from_counts -> miRNAmerge -> filter
                       (SUM:12-66, MRG:3-42, FLT:3-31; M-sized)
  write_read_tables    mapped.csv / unmapped.csv streamed from the arrays by the native writer
                       (W2C:582-619, :1172-1188; mrg_write_read_table)
  read_subset          seqDic-shaped records (+ alignments) of the reads claimed by chosen passes:
                       passes 0 / 8 feed the isomiR tables, the GFF and the -ai report, passes 2 / 3
                       the tRF tables -- none of them ever looks at another read
  isomir_dic           the grouping write_mapped_csv builds on the way (W2C:588-606)
  mirna_read_subset    the miRNA-claimed subset, optionally only what the -ai grouping keeps
  full_seq_dic         everything as the reference's seqDic (small inputs, tests)
"""
import ctypes as C
import os

import numpy as np

from . import _native, annotate, pack
from ._native import check
from .engine import CANON_PASS, ISOMIR_PASS, MIRGE_PASS_TABLE, ReadSet


def annotate_columns(engine, words, lens, nmask, quant, spike_in=False):
    """Packed reads (uint64 [W, n], uint8 [n], nmask or None, uint32 [n, S]) -> dict with host
    arrays pass_id, ref_id, pos, mm, the fused count vector and the per-pass stats."""
    rs = ReadSet(words, lens, nmask, quant, device=engine.device)
    res = engine.cascade(rs, engine.mirge_passes(spike_in=spike_in))
    counts = engine.tally(rs, res, engine.indexes["mirna"].n_ref, CANON_PASS, ISOMIR_PASS)
    pass_id, ref_id, pos, mm = res.to_host()
    return dict(pass_id=pass_id, ref_id=ref_id, pos=pos, mm=mm, counts=counts.cpu().numpy(), stats=res.stats,
                n_pass=res.n_pass)


def tables_from_columns(engine, cols, sampleList, merge_file, mirna_fa, canoRatio, spike_in=False):
    """(mirDic, logDic, mirNameSeqDic) as MAIN:381-386 leaves them, from the count vector."""
    log_dic = {"quantStats": [{"filename": s} for s in sampleList],
               "annotStats": [{"readsProcessed": st["processed"], "readsAligned": st["aligned"],
                               "cpuTime": st["ms"] * 1e-3} for st in cols["stats"]]}
    mir_dic, name_seq = {}, {}
    annotate.summarize_from_counts(cols["counts"], engine.indexes["mirna"].names, sampleList, log_dic, mir_dic,
                                   spike_in)
    annotate.miRNAmerge(merge_file, sampleList, mir_dic, mirna_fa, name_seq)
    annotate.filter(mir_dic, sampleList, log_dic, canoRatio)
    return mir_dic, log_dic, name_seq


def names_by_pass(engine, spike_in=False):
    """Entry names of the library each cascade pass aligns to (annot slot i + 1 holds names[i][ref])."""
    return [engine.indexes[row[0]].names for row in MIRGE_PASS_TABLE[:10 if spike_in else 9]]


def write_read_tables(outdir, annot_names, sample_list, words, lens, nmask, quant, pass_id, ref_id,
                      names_per_pass, extra=None):
    """mapped.csv and unmapped.csv from the host arrays, in array order, without a Python row loop.
    extra: (sequences, quant [k, S], pass_id, ref_id, ...) of the reads that never entered the packed arrays (reads
    beyond 255 nt, annotated by Engine.cascade_long): their rows are appended to the table they belong in."""
    lib = _native.load()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    W, n = words.shape
    lens = np.ascontiguousarray(lens, dtype=np.uint8)
    nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
    quant = np.ascontiguousarray(quant, dtype=np.uint32)
    S = quant.shape[1] if quant.ndim == 2 else 1
    pass_id = np.ascontiguousarray(pass_id, dtype=np.int8)
    ref_id = np.ascontiguousarray(ref_id, dtype=np.int32)
    n_slots = len(names_per_pass)
    flat = [nm_.encode("ascii") for names in names_per_pass for nm_ in names]
    arr = (C.c_char_p * max(len(flat), 1))(*flat)
    off = np.zeros(n_slots + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in names_per_pass], out=off[1:])
    header = ("uniqueSequence,annotFlag," + ",".join(annot_names) + "," + ",".join(sample_list) + "\n").encode()
    rows = {}
    for fn, mapped in (("mapped.csv", 1), ("unmapped.csv", 0)):
        k = C.c_uint64(0)
        check(lib.mrg_write_read_table(
            os.fsencode(os.path.join(outdir, fn)), mapped, header, 0, words.ctypes.data, W, n, lens.ctypes.data,
            None if nm is None else nm.ctypes.data, n, pass_id.ctypes.data, ref_id.ctypes.data, quant.ctypes.data, S,
            n_slots, arr, off.ctypes.data, C.byref(k)))
        rows[fn] = int(k.value)
    if extra is not None:
        e_seqs, e_quant, e_pass, e_ref = extra[:4]
        with open(os.path.join(outdir, "mapped.csv"), "a") as fm, open(os.path.join(outdir, "unmapped.csv"), "a") as fu:
            for k, seq in enumerate(e_seqs):
                p = int(e_pass[k])
                slots = [""] * n_slots
                if p >= 0:
                    slots[p] = names_per_pass[p][int(e_ref[k])]
                row = "%s,%d,%s,%s\n" % (seq, 1 if p >= 0 else 0, ",".join(slots), ",".join(str(int(x)) for x in e_quant[k]))
                (fm if p >= 0 else fu).write(row)
                rows["mapped.csv" if p >= 0 else "unmapped.csv"] += 1
    return rows


def add_long_records(records, align, extra, names_per_pass, passes=None, spike_in=False):
    """The reads of `extra` (write_read_tables) as seqDic-shaped records / alignments, as read_subset and
    full_seq_dic build them for the packed reads: those claimed by one of `passes` (None: every read)."""
    e_seqs, e_quant, e_pass, e_ref, e_pos, e_mm = extra
    width = 11 if spike_in else 10
    for k, seq in enumerate(e_seqs):
        p = int(e_pass[k])
        if passes is not None and p not in passes:
            continue
        annot = [1 if p >= 0 else 0] + [""] * (width - 1)
        if p >= 0:
            annot[p + 1] = names_per_pass[p][int(e_ref[k])]
            align[seq] = (p, int(e_ref[k]), int(e_pos[k]), int(e_mm[k]))
        records[seq] = {"quant": [int(x) for x in e_quant[k]], "annot": annot, "length": len(seq)}


def read_subset(words, lens, nmask, quant, pass_id, ref_id, pos, mm, names_per_pass, passes, spike_in=False,
                keep=None):
    """(records, alignments) of the reads claimed by one of `passes` (and selected by the optional
    boolean `keep`): records[seq] = {'quant', 'annot', 'length'} as quantReads.py:13-15 +
    updateAnnotDic (RAP:341-345) leave them; alignments[seq] = (pass, entry, 0-based offset,
    mismatches)."""
    sel = np.isin(pass_id, np.asarray(list(passes), dtype=pass_id.dtype))
    if keep is not None:
        sel &= keep
    idx = np.nonzero(sel)[0]
    sub_w = np.ascontiguousarray(np.asarray(words)[:, idx])
    sub_n = None if nmask is None else np.ascontiguousarray(np.asarray(nmask)[:, idx])
    seqs = pack.unpack_reads(sub_w, np.asarray(lens)[idx], sub_n)
    width = 11 if spike_in else 10
    quant = np.asarray(quant)
    records, align = {}, {}
    # (plain Python ints in one go: a numpy scalar per field and read was most of this loop's time at 10^5..10^6 reads)
    p_l, r_l, o_l, m_l = (np.asarray(a)[idx].tolist() for a in (pass_id, ref_id, pos, mm))
    q_l = quant[idx].tolist()
    blank = [1] + [""] * (width - 1)
    for k, seq in enumerate(seqs):
        p = p_l[k]
        annot = list(blank)
        annot[p + 1] = names_per_pass[p][r_l[k]]
        records[seq] = {"quant": q_l[k], "annot": annot, "length": len(seq)}
        align[seq] = (p, r_l[k], o_l[k], m_l[k])
    return records, align


def isomir_dic(records, n_samples):
    """The grouping write_mapped_csv builds while it writes mapped.csv (W2C:588-606): {miRNA (SNP
    suffix stripped): {'mirnas': {seq: counts}, 'isomirs': {seq: counts}}} over the reads claimed
    by the exact-miRNA or the isomiR pass."""
    out = {}
    for seq, rec in records.items():
        annot = rec["annot"]
        isomir, mirna = annot[9], annot[1]
        if isomir == "" and mirna == "":
            continue
        key, kind = (isomir, "isomirs") if isomir != "" else (mirna, "mirnas")
        if ".SNP" in key:
            key = key.split(".SNP")[0]
        slot = out.setdefault(key, {"mirnas": {}, "isomirs": {}})
        slot[kind][seq] = rec["quant"][:n_samples]
    return out


def write_isomir_tables(isomir_path, sample_path, sample_list, words, lens, nmask, quant, pass_id, ref_id, mirna_names, log_dic):
    """isomirs.csv + isomirs.samples.csv (W2C:1090-1170) from the arrays by the native writer (mrg_write_isomir_tables) -- the
    files report.write_isomir_tables(..., isomir_dic(read_subset(...)), ...) writes, byte for byte, without a Python record per
    miRNA read.  Returns the rows of isomirs.csv."""
    lib = _native.load()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    W, n = words.shape
    lens = np.ascontiguousarray(lens, dtype=np.uint8)
    nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
    quant = np.ascontiguousarray(quant, dtype=np.uint32)
    S = quant.shape[1] if quant.ndim == 2 else 1
    pass_id = np.ascontiguousarray(pass_id, dtype=np.int8)
    ref_id = np.ascontiguousarray(ref_id, dtype=np.int32)
    # a miRNA entry's group: its name with the SNP suffix stripped (W2C:599-600)
    gid, gnames = {}, []
    group_of = np.empty(len(mirna_names), dtype=np.int32)
    for e, name in enumerate(mirna_names):
        key = name.split(".SNP")[0] if ".SNP" in name else name
        if key not in gid:
            gid[key] = len(gnames)
            gnames.append(key)
        group_of[e] = gid[key]
    arr = (C.c_char_p * max(len(gnames), 1))(*[g.encode("ascii") for g in gnames])
    filtered = np.array([float(log_dic["quantStats"][i]["mirnaReadsFiltered"]) for i in range(S)], dtype=np.float64)
    h1 = "miRNA,sequence" + "".join("," + s for s in sample_list) + ",Entropy\n"
    h2 = "miRNA" + "".join(",%s isomir+miRNA Entropy,%s Canonical Sequence,%s Canonical RPM,%s Top Isomir RPM" % (s, s, s, s)
                           for s in sample_list) + "\n"
    k = C.c_uint64(0)
    check(lib.mrg_write_isomir_tables(
        os.fsencode(isomir_path), os.fsencode(sample_path), h1.encode(), h2.encode(), words.ctypes.data, W, n, lens.ctypes.data,
        None if nm is None else nm.ctypes.data, n, pass_id.ctypes.data, ref_id.ctypes.data, quant.ctypes.data, S,
        CANON_PASS, ISOMIR_PASS, group_of.ctypes.data, len(mirna_names), arr, len(gnames), filtered.ctypes.data, C.byref(k)))
    return int(k.value)


def mirna_read_subset(engine, cols, words, lens, nmask, quant, log_dic=None, spike_in=False, for_a2i=False):
    """seqDic-shaped records ({'quant', 'annot', 'length'}) of the reads claimed by the miRNA passes.
    for_a2i: keep an isomiR read only if its RPM (against mirnaReadsFiltered, W2C:1240-1247) is
    >= 1 in some sample, which is all the -ai block ever groups."""
    pass_id = cols["pass_id"]
    keep = None
    quant = np.asarray(quant)
    if for_a2i:
        total = np.array([q["mirnaReadsFiltered"] for q in log_dic["quantStats"]], dtype=np.float64)
        rpm_ok = (1000000.0 * quant.astype(np.float64) / total[None, :] >= 1).any(axis=1)
        keep = (pass_id == CANON_PASS) | rpm_ok
    records, _ = read_subset(words, lens, nmask, quant, pass_id, cols["ref_id"], cols["pos"], cols["mm"],
                             names_by_pass(engine, spike_in), (CANON_PASS, ISOMIR_PASS), spike_in, keep)
    return records


def full_seq_dic(words, lens, nmask, quant, pass_id, ref_id, names_per_pass, spike_in=False):
    """Every read as the reference's seqDic (for small inputs and the tests; O(n) Python objects)."""
    seqs = pack.unpack_reads(np.asarray(words), np.asarray(lens), nmask)
    width = 11 if spike_in else 10
    out = {}
    quant = np.asarray(quant)
    for i, s in enumerate(seqs):
        p = int(pass_id[i])
        annot = [1 if p >= 0 else 0] + [""] * (width - 1)
        if p >= 0:
            annot[p + 1] = names_per_pass[p][int(ref_id[i])]
        out[s] = {"quant": [int(x) for x in quant[i]], "annot": annot, "length": len(s)}
    return out
