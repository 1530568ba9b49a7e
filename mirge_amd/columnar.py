"""Columnar driver: the annotate path without the reference's dict of dicts.

The reference keeps every unique read in `seqDic[seq] = {'quant', 'annot', 'length'}` (QNT:13-15)
and walks that dict in runAnnotationPipeline, summarize and writeDataToCSV; at the 10^7-10^8
unique reads of BASELINE configs 3-5 that does not fit a Python process.  Here the per-read state
stays in arrays (HBM, then numpy) and only the M-sized tables and the SUBSETS of reads the
downstream consumers actually look at are turned into the reference's shapes:

  annotate_columns     cascade + tally on packed reads; pass_id / ref_id / pos / mm + counts
  tables_from_columns  mirDic / quantStats via summarize_from_counts -> miRNAmerge -> filter
                       (SUM:12-66, MRG:3-42, FLT:3-31; M-sized)
  mirna_read_subset    seqDic-shaped records of the reads claimed by pass 0 or 8, optionally only
                       those the -ai grouping keeps (W2C:1223-1250: exact miRNA, or isomiR with
                       RPM >= 1 in some sample -- at most ~10^6 reads per sample by construction)

`a2i.a_to_i_report`, `report.write_isomir_tables` and the GFF writer take those records
unchanged: they only ever read entries whose annot slot 1 or 9 is set.
"""
import numpy as np

from . import annotate, pack
from .engine import CANON_PASS, ISOMIR_PASS, ReadSet


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


def mirna_read_subset(engine, cols, words, lens, nmask, quant, log_dic=None, spike_in=False, for_a2i=False):
    """seqDic-shaped records ({'quant', 'annot', 'length'}) of the reads claimed by the miRNA passes.
    for_a2i: keep an isomiR read only if its RPM (against mirnaReadsFiltered, W2C:1240-1247) is
    >= 1 in some sample, which is all the -ai block ever groups."""
    pass_id, ref_id = cols["pass_id"], cols["ref_id"]
    keep = (pass_id == CANON_PASS) | (pass_id == ISOMIR_PASS)
    quant = np.asarray(quant)
    if for_a2i:
        total = np.array([q["mirnaReadsFiltered"] for q in log_dic["quantStats"]], dtype=np.float64)
        rpm_ok = (1000000.0 * quant.astype(np.float64) / total[None, :] >= 1).any(axis=1)
        keep &= (pass_id == CANON_PASS) | rpm_ok
    idx = np.nonzero(keep)[0]
    sub_w = np.ascontiguousarray(np.asarray(words)[:, idx])
    sub_n = None if nmask is None else np.ascontiguousarray(np.asarray(nmask)[:, idx])
    seqs = pack.unpack_reads(sub_w, np.asarray(lens)[idx], sub_n)
    names = engine.indexes["mirna"].names
    width = 11 if spike_in else 10
    out = {}
    for k, i in enumerate(idx):
        annot = [1] + [""] * (width - 1)
        annot[1 if pass_id[i] == CANON_PASS else 9] = names[int(ref_id[i])]
        out[seqs[k]] = {"quant": [int(x) for x in quant[i]], "annot": annot, "length": len(seqs[k])}
    return out
