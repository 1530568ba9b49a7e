"""Host-side mirror of the reference's operator interface for the hot path.

Same function names, argument order/meaning and error behaviour as
  runAnnotationPipeline   utils/runAnnotationPipeline.py:566
  summarize               utils/summarize.py:3
  miRNAmerge              utils/miRNAmerge.py:3
  filter                  utils/filter.py:3
with ONE substitution: where the reference takes `bowtieBinary` (a directory
holding bowtie/bowtie-inspect, used through os.system), these take a
`mirge_amd.engine.Engine` (one GPU).  `file_*` stay bowtie-style index prefixes
(MAIN:269-281); they are resolved by FmIndex.open_prefix.

The dict-shaped state (`seqDic`, `mirDic`, `logDic`) is exactly the reference's,
so a maintainer can swap the imports in __main__.py:375-385.
Inside, the dicts are flattened to packed columnar arrays, the cascade and the
tally run on the GPU through the C-ABI, and the results are written back into
the dicts.  For 10^7+ reads use the columnar Engine API directly.
"""
import os
import sys
import time

import numpy as np

from . import pack
from ._native import MirgeAmdError
from .engine import CANON_PASS, ISOMIR_PASS, MIRGE_PASS_TABLE, Engine, ReadSet, split_counts
from .index import FmIndex

RNA_LIBRARY_LABEL = ["miRNA", "hairpin", "mature tRNA", "precusor tRNA", "snoRNA", "rRNA",
                     "ncrna others", "mRNA", "isomiR", "spikeIn"]  # RAP:575/589
CATEGORY_KEYS = ["mirnaReads", "hairpinReads", "maturetrnaReads", "pretrnaReads", "snornaReads",
                 "rrnaReads", "ncrnaOthersReads", "mrnaReads"]  # SUM:22-29, by annot slot 1..8


def _ensure_libraries(engine, files, cache=False):
    """files: {library key: index prefix}.  Loads/builds (concurrently: the native calls release
    the GIL) and uploads each once.  cache: an index that had to be built from FASTA / .ebwt is
    saved as <prefix>.mrgfm for the next run."""
    tags = engine.__dict__.setdefault("_loaded_tags", {})
    todo = [(key, prefix) for key, prefix in files.items() if tags.get(key) != "%s@%s" % (key, prefix)]
    if not todo:
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(todo)) as pool:
        opened = list(pool.map(lambda kp: FmIndex.open_prefix(kp[1], cache=cache), todo))
    for (key, prefix), ix in zip(todo, opened):
        engine.add_library(key, ix)
        tags[key] = "%s@%s" % (key, prefix)


def runAnnotationPipeline(engine, seqDic, numCPU, phred64, annotNameList, outputdir, logDic,
                          file_mirna, file_hairpin, file_mature_tRNA, file_pre_tRNA, file_snoRNA,
                          file_rRNA, file_ncrna_others, file_mrna, spikeIn, file_spikeIn,
                          gff_output, miRNamePreNameDic, isomiRContentDic, miRNA_database,
                          trf_output, trnaStruDic, trfContentDic, sampleList):
    """RAP:566-707.  Mutates seqDic[*]['annot'] (RAP:341-352) and appends one
    {'cpuTime','readsProcessed','readsAligned'} dict per pass to
    logDic['annotStats'] (RAP:640-705).  Also leaves the located alignments in
    logDic['_alignments'] = {seq: (pass, entry index, 0-based offset, mismatches)}
    for the isomiR / A-to-I consumers."""
    files = {"mirna": file_mirna, "hairpin": file_hairpin, "mature_trna": file_mature_tRNA,
             "pre_trna": file_pre_tRNA, "snorna": file_snoRNA, "rrna": file_rRNA,
             "ncrna_others": file_ncrna_others, "mrna": file_mrna}
    if spikeIn:
        files["spike-in"] = file_spikeIn
    try:
        _ensure_libraries(engine, files)
        # one length byte per read in the packed batch: reads beyond 255 nt run the same passes through
        # mrg_cascade_run_long (the reference offers a read of any length to every pass: RAP:543-554)
        seqs = [s for s in seqDic.keys() if len(s) <= 255]
        long_seqs = [s for s in seqDic.keys() if len(s) > 255]
        words, lens, nmask = pack.pack_reads(seqs) if seqs else \
            (np.zeros((1, 0), np.uint64), np.zeros(0, np.uint8), None)
        passes = engine.mirge_passes(spike_in=bool(spikeIn))
        t0 = time.time()
        rs = ReadSet(words, lens, nmask, None, device=engine.device)
        res = engine.cascade(rs, passes)
        stats = res.stats
        pass_id, ref_id, pos, mm = res.to_host()
        if long_seqs:
            lp, lr, lo, lm, _ = engine.cascade_long(long_seqs, passes, stats=stats)
            seqs = seqs + long_seqs
            pass_id, ref_id = np.concatenate([pass_id, lp]), np.concatenate([ref_id, lr])
            pos, mm = np.concatenate([pos, lo]), np.concatenate([mm, lm])
        wall = time.time() - t0
    except MirgeAmdError as e:
        # RAP:661-663 / RAP:702-704: message + exit status 1
        print("Alignment to library %s exited with none-zero status.\n" % getattr(e, "library", "?"))
        print(str(e), file=sys.stderr)
        sys.exit(1)
    table = MIRGE_PASS_TABLE[:len(passes)]
    names = [engine.indexes[row[0]].names for row in table]
    align = logDic.setdefault("_alignments", {})
    for i, seq in enumerate(seqs):
        p = int(pass_id[i])
        if p < 0:
            continue
        rec = seqDic[seq]
        rec["annot"][0] = 1
        rec["annot"][p + 1] = names[p][int(ref_id[i])]
        align[seq] = (p, int(ref_id[i]), int(pos[i]), int(mm[i]))
    if gff_output:
        # RAP:609-619 + :653-656: after pass 0 and after pass 8, classify the claimed reads
        from . import isomir
        hairpin_seqs = engine.indexes["hairpin"].name_seq_dict()
        mirna_seqs = engine.indexes["mirna"].name_seq_dict()
        for pass_index in (0, 8):
            if pass_index >= len(passes):
                continue
            trim = 0 if pass_index == 0 else 3  # -5 1 -3 2 shortens the aligned read
            hits = {}
            for i, seq in enumerate(seqs):
                if int(pass_id[i]) == pass_index:
                    hits[seq] = (names[pass_index][int(ref_id[i])], int(pos[i]) + 1,
                                 "%dM" % (len(seq) - trim))
            isomir.build_isomir_content(isomiRContentDic, hits, pass_index, miRNamePreNameDic,
                                        hairpin_seqs, mirna_seqs, miRNA_database)
    if trf_output:
        # RAP:629-634, :657-660, :698-701: `-a --best --strata` listings of the two tRNA passes
        from . import trf
        trf.collect_trf_content(trfContentDic, seqDic, sampleList, trnaStruDic,
                                engine.indexes["pre_trna"].name_seq_dict(), trf.engine_lister(engine))
    gpu_ms = sum(s["ms"] for s in stats) or 1.0
    for s in stats:
        # the reference stores wall seconds per bowtie run (RAP:641-645); split ours by device time
        logDic["annotStats"].append({"cpuTime": wall * s["ms"] / gpu_ms,
                                     "readsProcessed": s["processed"], "readsAligned": s["aligned"]})


def summarize(seqDic, sampleList, logDic, mirDic, file_mirna, outputdir, spikeIn, engine):
    """SUM:3-66.  Bins = entry names of the miRNA index (`bowtie-inspect -n`, SUM:6-9)."""
    S = len(sampleList)
    _ensure_libraries(engine, {"mirna": file_mirna})
    mir_names = engine.indexes["mirna"].names
    M = len(mir_names)
    name_to_bin = {n: i for i, n in enumerate(mir_names)}
    n_pass = 10 if spikeIn else 9
    seqs = list(seqDic.keys())
    n = len(seqs)
    pass_id = np.full(n, -1, dtype=np.int8)
    ref_id = np.zeros(n, dtype=np.int32)
    quant = np.zeros((n, S), dtype=np.uint32)
    for i, seq in enumerate(seqs):
        rec = seqDic[seq]
        quant[i] = rec["quant"]
        annot = rec["annot"]
        # SUM:38-66: slot 1 | slot 9 first, then 2..8, then 10
        if annot[1] != "":
            pass_id[i], ref_id[i] = 0, name_to_bin[annot[1]]
        elif annot[9] != "":
            pass_id[i], ref_id[i] = 8, name_to_bin[annot[9]]
        else:
            for slot in range(2, 9):
                if annot[slot] != "":
                    pass_id[i] = slot - 1
                    break
            else:
                if spikeIn and annot[10] != "":
                    pass_id[i] = 9
    import torch
    dev = engine.device
    rs = ReadSet(np.zeros((1, n), np.uint64), np.zeros(n, np.uint8), None, quant, device=dev)

    class _R:
        pass
    r = _R()
    r.pass_id = torch.from_numpy(pass_id).to(dev)
    r.ref_id = torch.from_numpy(ref_id).to(dev)
    r.n_pass = n_pass
    counts = engine.tally(rs, r, M, CANON_PASS, ISOMIR_PASS).cpu().numpy()
    summarize_from_counts(counts, mir_names, sampleList, logDic, mirDic, spikeIn)


def summarize_from_counts(counts, mir_names, sampleList, logDic, mirDic, spikeIn=False):
    """The bookkeeping of SUM:12-66 from the fused count vector of mrg_tally_run
    ([mir_quant | mir_iscan | category totals | trimmedUniq]): fills mirDic and
    logDic['quantStats'][i] (which must exist)."""
    S, M = len(sampleList), len(mir_names)
    n_pass = 10 if spikeIn else 9
    q, c, cat, uniq = split_counts(np.asarray(counts), M, S, n_pass)
    for i, name in enumerate(mir_names):
        mirDic[name] = {"quant": [int(x) for x in q[i]], "iscan": [int(x) for x in c[i]]}
    for s in range(S):
        qs = logDic["quantStats"][s]
        qs["trimmedUniq"] = int(uniq[s])
        qs["mirnaReads"] = int(cat[0, s] + cat[8, s])
        for slot in range(2, 9):
            qs[CATEGORY_KEYS[slot - 1]] = int(cat[slot - 1, s])
        qs["remReads"] = int(cat[n_pass, s])
        if spikeIn:
            qs["spikeInReads"] = int(cat[9, s])


def miRNAmerge(mergeLibFile, sampleList, mirDic, mirna_fa_tmp, mirNameSeqDic):
    """MRG:3-42: load name->sequence of the 2-line miRNA FASTA, fold the members
    of each `merged,m1,m2,...` line into the merged bin and drop the members."""
    with open(mirna_fa_tmp, "r") as fh:
        rows = iter(fh)
        while True:
            header = next(rows, "")
            if header == "":
                break
            mirNameSeqDic[header.strip()[1:]] = next(rows, "").strip()
    if not os.path.isfile(mergeLibFile):
        print("Cannot find merges file, skipping merge step.\n")
        return
    S = len(sampleList)
    drop = set()
    with open(mergeLibFile, "r") as fh:
        for line in fh:
            if line == "":
                break
            fields = line.strip().split(",")
            target = fields[0]
            for member in fields[1:]:
                src = mirDic.get(member)
                if src is None:
                    continue
                for s in range(S):
                    if src["quant"][s] > 0:
                        dst = mirDic.get(target)
                        if dst is None:
                            dst = mirDic[target] = {"quant": [0] * S, "iscan": [0] * S}
                        dst["quant"][s] += src["quant"][s]
                        dst["iscan"][s] += src["iscan"][s]
                drop.add(member)
    for name in drop:
        mirDic.pop(name, None)


def filter(mirDic, sampleList, logDic, canoRatioTmp):
    """FLT:3-31: zero a miRNA's count in a sample when its canonical reads are < 2
    or their share is below the threshold; totals; abort when a sample is empty."""
    threshold = float(canoRatioTmp)
    S = len(sampleList)
    for rec in mirDic.values():
        for s in range(S):
            q, c = rec["quant"][s], rec["iscan"][s]
            ratio = float(c) / q if q != 0 else 1.1
            if c < 2 or ratio < threshold:
                rec["quant"][s] = 0
    for rec in mirDic.values():
        for s in range(S):
            if rec["quant"][s] > 0:
                qs = logDic["quantStats"][s]
                qs["mirnaReadsFiltered"] = qs.get("mirnaReadsFiltered", 0) + rec["quant"][s]
                qs["mirnaUniqFiltered"] = qs.get("mirnaUniqFiltered", 0) + 1
    for s in range(S):
        if logDic["quantStats"][s].get("mirnaReadsFiltered", 0) == 0:
            print("No miRNA reads were found in sample %s. Please check your files and provided "
                  "arguments.\n" % (sampleList[s]))
            sys.exit(1)


def quantReads(reads, seqDic, readLengthDic, sampleCount, sampleIndex, spikeIn=False):
    """Collapse of quantReads.py:3-24 for an in-memory list of trimmed reads
    (FASTQ ingest is SURVEY.md 8f rank 1)."""
    for seq in reads:
        rec = seqDic.get(seq)
        if rec is None:
            rec = seqDic[seq] = {"quant": [0] * sampleCount,
                                 "annot": [0] + [""] * (10 if spikeIn else 9), "length": len(seq)}
        rec["quant"][sampleIndex] += 1
        readLengthDic.setdefault(len(seq), [0] * sampleCount)[sampleIndex] += 1
