"""Dict-shaped restatement of the reference's cascade + tally (TEST INFRASTRUCTURE).

Follows, function by function (paths under /root/reference/src/mirge/utils/):
  pass table / length filters      runAnnotationPipeline.py:574-599
  survivor selection               runAnnotationPipeline.py:543-554 (writeSeqToAnnot)
  main loop, per-pass counters     runAnnotationPipeline.py:636-705
  poly-T pass (index 3)            runAnnotationPipeline.py:664-701
  annot update / fan-out           runAnnotationPipeline.py:341-352
  summarize                        summarize.py:12-66
  miRNAmerge                       miRNAmerge.py:13-41
  filter                           filter.py:3-31
  seqDic record                    quantReads.py:9-16

Where the reference shells out to bowtie, this calls oracle.model.align_batch
(the exhaustive-scan model of bowtie's rules).  The state objects have the
reference's own shapes so golden vectors captured from the reference compare
directly:
  seqDic[seq] = {'quant': [per-sample counts], 'annot': [flag, slot1..slot9(10)], 'length': n}
  mirDic[name] = {'quant': [...], 'iscan': [...]}
  logDic = {'quantStats': [per-sample dict], 'annotStats': [per-pass dict]}
"""
import re

from . import model

BIG_SEED = 1 << 20  # "-v" mode: the whole read is the seed

# (library key, lengthFilter, seed_len, max_mm_seed, max_mm_total, trim5, trim3)
# one row per bowtie command line of runAnnotationPipeline.py:577-586 / :688
PASS_TABLE = [
    ("mirna", -26, 28, 0, 2, 0, 0),            # -n 0
    ("hairpin", 25, 28, 1, 2, 0, 0),           # -n 1
    ("mature_trna", 0, BIG_SEED, 1, 1, 0, 0),  # -v 1 -a --best --strata
    ("pre_trna", 0, BIG_SEED, 0, 0, 0, 0),     # -v 0 -a --best --strata, poly-T stripped reads
    ("snorna", 0, 28, 1, 2, 0, 0),             # -n 1
    ("rrna", 0, 28, 1, 2, 0, 0),               # -n 1
    ("ncrna_others", 0, 28, 1, 2, 0, 0),       # -n 1
    ("mrna", 0, 28, 0, 2, 0, 0),               # -n 0
    ("mirna", 0, BIG_SEED, 2, 2, 1, 2),        # -5 1 -3 2 -v 2 --best
    ("spike-in", 0, 28, 0, 2, 0, 0),           # -n 0, only with -spikeIn
]
RNA_LIBRARY_LABEL = ["miRNA", "hairpin", "mature tRNA", "precusor tRNA", "snoRNA", "rRNA",
                     "ncrna others", "mRNA", "isomiR", "spikeIn"]
POLY_T_PASS = 3


def new_seq_record(seq, n_samples, spike_in=False):
    """quantReads.py:12-15"""
    return {"quant": [0] * n_samples, "annot": [0] + [""] * (10 if spike_in else 9),
            "length": len(seq)}


def collapse(sample_reads, spike_in=False):
    """quantReads.py:3-24 for already-trimmed reads: sample_reads is a list (one
    per sample) of lists of sequences.  Returns (seqDic, readLengthDic)."""
    seq_dic, len_dic = {}, {}
    S = len(sample_reads)
    for si, reads in enumerate(sample_reads):
        for seq in reads:
            rec = seq_dic.get(seq)
            if rec is None:
                rec = seq_dic[seq] = new_seq_record(seq, S, spike_in)
            rec["quant"][si] += 1
            len_dic.setdefault(len(seq), [0] * S)[si] += 1
    return seq_dic, len_dic


def select_survivors(length_filter, seq_dic):
    """writeSeqToAnnot, runAnnotationPipeline.py:543-554 (the FASTA content)."""
    out = []
    for seq, rec in seq_dic.items():
        if rec["annot"][0] != 0:
            continue
        if length_filter < 0:
            if rec["length"] < -length_filter:
                out.append(seq)
        elif length_filter > 0:
            if rec["length"] > length_filter:
                out.append(seq)
        else:
            out.append(seq)
    return out


def _bowtie(library, reads, seed_len, mm_seed, mm_total, trim5, trim3):
    """Stand-in for one bowtie run: returns {read: (ref_idx, pos0, mm)} for the
    aligned reads plus (processed, aligned).  Trimming as `-5/-3` do."""
    trimmed = [r[trim5:len(r) - trim3] if trim3 else r[trim5:] for r in reads]
    ref, pos, mm = model.align_batch(library, trimmed, seed_len, mm_seed, mm_total)
    hits = {}
    for i, r in enumerate(reads):
        if ref[i] >= 0:
            hits[r] = (int(ref[i]), int(pos[i]), int(mm[i]))
    return hits, len(reads), int((ref >= 0).sum())


def run_annotation_pipeline(seq_dic, libraries, log_dic, spike_in=False, align_dic=None, n_passes=None):
    """runAnnotationPipeline.py:566-707 without -gff / -trf side products.

    libraries: {key: oracle.model.Library} for the keys of PASS_TABLE.
    align_dic (optional dict) receives seq -> (pass index, ref idx, pos0, mm).
    n_passes: only the first so many iterations of the loop (a truncated cascade)."""
    n_pass = (10 if spike_in else 9) if n_passes is None else int(n_passes)
    for i in range(n_pass):
        key, length_filter, seed_len, mm_seed, mm_total, t5, t3 = PASS_TABLE[i]
        lib = libraries[key]
        survivors = select_survivors(length_filter, seq_dic)
        if i != POLY_T_PASS:
            hits, processed, aligned = _bowtie(lib, survivors, seed_len, mm_seed, mm_total, t5, t3)
            for seq, (ref, pos, mm) in hits.items():           # updateAnnotDic :341-345
                seq_dic[seq]["annot"][0] = 1
                seq_dic[seq]["annot"][i + 1] = lib.names[ref]
                if align_dic is not None:
                    align_dic[seq] = (i, ref, pos, mm)
        else:
            # :664-686: reads ending in >= 3 T, all trailing T removed, >= 11 nt left;
            # the FASTA keeps one record per ORIGINAL read, so counters count originals
            stripped_of = {}
            fasta = []
            for seq in survivors:
                if re.search("T{3,}$", seq) is None:
                    continue
                sub = seq.rstrip("T")
                if len(sub) >= 11:
                    fasta.append(sub)
                    stripped_of.setdefault(sub, []).append(seq)
            hits, processed, aligned = _bowtie(lib, fasta, seed_len, mm_seed, mm_total, t5, t3)
            for sub, (ref, pos, mm) in hits.items():           # updateAnnotDic2 :347-352
                for seq in stripped_of[sub]:
                    seq_dic[seq]["annot"][0] = 1
                    seq_dic[seq]["annot"][i + 1] = lib.names[ref]
                    if align_dic is not None:
                        align_dic[seq] = (i, ref, pos, mm)
        log_dic["annotStats"].append({"readsProcessed": processed, "readsAligned": aligned})


def scan_cascade(libraries, passes, seqs):
    """The cascade of runAnnotationPipeline.py:636-705 for an ARBITRARY pass table and reads of ANY length, by
    exhaustive scan, one read at a time (writeSeqToAnnot :543-554 + the length filter of :574, the poly-T rule of
    :664-686, `-5` / `-3`, then the bowtie model): what the product's long-read lane and its random pass tables
    are checked against.  libraries: {key: oracle.model.Library}; passes: dicts with lib (a key of `libraries`),
    min_len, max_len (255 or more = no upper bound, as in include/mirge_amd.h), seed_len, max_mm_seed, max_mm_total,
    trim5, trim3, poly_t.  Returns (pass_id, ref_id, pos, mm, [(processed, aligned) per pass])."""
    import numpy as np
    n = len(seqs)
    pass_id = np.full(n, -1, np.int8)
    ref_id = np.full(n, -1, np.int32)
    pos = np.full(n, -1, np.int32)
    mm = np.zeros(n, np.uint8)
    counts = [[0, 0] for _ in passes]
    open_reads = list(range(n))
    for i, p in enumerate(passes):
        offered, subs = [], []
        for r in open_reads:
            seq = seqs[r]
            if len(seq) < p["min_len"] or (p["max_len"] < 255 and len(seq) > p["max_len"]):
                continue
            sub = seq
            if p["poly_t"]:
                if re.search("T{3,}$", seq) is None:
                    continue
                sub = seq.rstrip("T")
                if len(sub) < 11:
                    continue
            offered.append(r)
            subs.append(sub[p["trim5"]:len(sub) - p["trim3"]] if p["trim3"] else sub[p["trim5"]:])
        ref, ps, m = model.align_batch(libraries[p["lib"]], subs, p["seed_len"], p["max_mm_seed"], p["max_mm_total"])
        counts[i][0] = len(offered)
        claimed = set()
        for k, r in enumerate(offered):
            if ref[k] >= 0:
                counts[i][1] += 1
                pass_id[r], ref_id[r], pos[r], mm[r] = i, ref[k], ps[k], m[k]
                claimed.add(r)
        open_reads = [r for r in open_reads if r not in claimed]
    return pass_id, ref_id, pos, mm, counts


CATEGORY_KEYS = ["mirnaReads", "hairpinReads", "maturetrnaReads", "pretrnaReads", "snornaReads",
                 "rrnaReads", "ncrnaOthersReads", "mrnaReads"]


def summarize(seq_dic, sample_list, log_dic, mir_dic, mirna_names, spike_in=False):
    """summarize.py:12-66; mirna_names is what `bowtie-inspect -n` lists (:6-9)."""
    S = len(sample_list)
    for name in mirna_names:
        mir_dic[name] = {"quant": [0] * S, "iscan": [0] * S}
    for s in range(S):
        qs = log_dic["quantStats"][s]
        qs["trimmedUniq"] = 0
        for k in CATEGORY_KEYS:
            qs[k] = 0
        qs["remReads"] = 0
        if spike_in:
            qs["spikeInReads"] = 0
    for rec in seq_dic.values():
        annot = rec["annot"]
        for s in range(S):
            q = rec["quant"][s]
            if q == 0:
                continue
            qs = log_dic["quantStats"][s]
            qs["trimmedUniq"] += 1
            if annot[1] != "" or annot[9] != "":
                qs["mirnaReads"] += q
                if annot[1] != "":
                    mir_dic[annot[1]]["quant"][s] += q
                    mir_dic[annot[1]]["iscan"][s] += q
                else:
                    mir_dic[annot[9]]["quant"][s] += q
                continue
            for slot in range(2, 9):
                if annot[slot] != "":
                    qs[CATEGORY_KEYS[slot - 1]] += q
                    break
            else:
                if spike_in and annot[10] != "":
                    qs["spikeInReads"] += q
                else:
                    qs["remReads"] += q


def mirna_merge(merge_lines, sample_list, mir_dic):
    """miRNAmerge.py:13-41; merge_lines = lines of <sp>_merges_<db>.csv."""
    S = len(sample_list)
    doomed = []
    for line in merge_lines:
        fields = line.strip().split(",")
        target = fields[0]
        for member in fields[1:]:
            for s in range(S):
                if member not in mir_dic:
                    continue
                if mir_dic[member]["quant"][s] > 0:
                    if target not in mir_dic:
                        mir_dic[target] = {"quant": [0] * S, "iscan": [0] * S}
                        mir_dic[target]["quant"][s] = mir_dic[member]["quant"][s]
                        mir_dic[target]["iscan"][s] = mir_dic[member]["iscan"][s]
                    else:
                        mir_dic[target]["quant"][s] += mir_dic[member]["quant"][s]
                        mir_dic[target]["iscan"][s] += mir_dic[member]["iscan"][s]
                doomed.append(member)
    for name in set(doomed):
        mir_dic.pop(name, None)


class NoMirnaReads(Exception):
    """filter.py:25-31 prints a message and exits with status 1."""


def filter_mirnas(mir_dic, sample_list, log_dic, cano_ratio):
    """filter.py:3-31 (threshold arrives as a string, :5)."""
    S = len(sample_list)
    thr = float(cano_ratio)
    for rec in mir_dic.values():
        for s in range(S):
            q, c = rec["quant"][s], rec["iscan"][s]
            ratio = (float(c) / q) if q != 0 else 1.1
            if c < 2 or ratio < thr:
                rec["quant"][s] = 0
    for rec in mir_dic.values():
        for s in range(S):
            if rec["quant"][s] > 0:
                qs = log_dic["quantStats"][s]
                qs["mirnaReadsFiltered"] = qs.get("mirnaReadsFiltered", 0) + rec["quant"][s]
                qs["mirnaUniqFiltered"] = qs.get("mirnaUniqFiltered", 0) + 1
    for s in range(S):
        if log_dic["quantStats"][s].get("mirnaReadsFiltered", 0) == 0:
            raise NoMirnaReads(
                "No miRNA reads were found in sample %s. Please check your files and provided "
                "arguments.\n" % sample_list[s])
