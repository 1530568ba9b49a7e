"""A-to-I editing tally for `-ai`.

Restates writeDataToCSV.py:1221-1615 and its helpers (:35-229 judgeAllign, removeDash,
refineName, checkSeqList, align2TargetSeq, A2IEditing; :344-351 is in report.py):

  1. group the reads claimed by pass 0 / pass 8 per (merged) miRNA          :1224-1249
  2. keep reads whose best genome hit (-n 1 -a -3 2, both strands) is unique :1251-1287
  3. per miRNA x sample with RPM >= 1: align reads to the canonical sequence, count
     A->G per position, binomial p (p_mismatch 0.001)                       :1293-1321, :145-229
  4. Benjamini-Hochberg style adjustment per sample                         :1353-1364
  5. a2IEditing.report.csv (filters: a significant sample; canonical RPM >= 1; not in the
     repetitive-element list; the edited sequence does not occur in the genome -n 0 -3 2),
     a2IEditing.report.newform.csv, a2IEditing.detail.txt                   :1373-1594

Two things are not the reference's own arithmetic:
  * `pairwise2.align.localms(target, seq, 2, -1, -20, -20)` (Biopython, absent from the
    image and from /root/reference): `local_pair` restates its local alignment with affine gaps
    and Biopython 1.70-1.76's traceback order (which alignment is listed first among equals);
    the golden fixture answers the reference's call with an independent restatement of the
    same published algorithm.  Unpinned against a real Biopython.
  * the two genome bowtie runs: answered by a `genome` object with
      unique_best(reads) -> set of reads whose best stratum (<= 1 seed mismatch, 3' 2 nt
                            trimmed, both strands) holds exactly one alignment
      exact_hit(reads)   -> set of reads that align with 0 seed mismatches (same trimming)
    (mirge_amd.a2i.EngineGenome on the GPU, an exhaustive scan in the oracle).
The mismatch-type table (mismatchCountAnalysis, :248-342) is computed by the reference
and then deleted (:1536-1537), so it is not produced here.
"""
import math
import os

from .report import py2_float_str

P_MISMATCH = 0.001   # W2C:146
TAIL_SHIFT = 5       # W2C:147: the last 5 nt of the miRNA are not scored
PERCENT_CUTOFF = 2.0  # W2C:567
# W2C:1541-1551: run accessions of the reference's own study are relabelled in the newform table
SAMPLE_LABELS = {"SRR837842": "Colon 1", "SRR837839": "Colon 2", "SRR5127219": "Colon cell",
                 "SRR1646473": "Colon cancer 1", "SRR1646493": "Colon cancer 2", "SRR1917324": "DKO1",
                 "SRR1917336": "DLD1", "SRR1917329": "DKS8", "SRR567638": "Placenta 2"}


# ------------------------------------------------------------------ alignment
def _local_scores(target, seq, match, mismatch, gap_open, gap_extend):
    """Smith-Waterman scores as pairwise2's local mode fills them: score[r][c] = best alignment
    ending with target[:r] / seq[:c] consumed (floored at 0), a gap of n costing
    open + (n - 1) * extend, EXCEPT in the last row and the last column, where a gap is free
    (end gaps are not penalised: the best score is passed along).  `stretched[r][c]` says that the
    cell's score can come from a gap of two or more."""
    la, lb = len(target), len(seq)
    score = [[0] * (lb + 1) for _ in range(la + 1)]
    stretched = [[False] * (lb + 1) for _ in range(la + 1)]
    down = [0] + [2 * gap_open + gap_extend * (c - 1) for c in range(1, lb + 1)]   # gap in seq, per column
    for r in range(1, la + 1):
        tr = target[r - 1]
        across = 2 * gap_open + gap_extend * (r - 1)                               # gap in target, this row
        row, above = score[r], score[r - 1]
        for c in range(1, lb + 1):
            pair = above[c - 1] + (match if tr == seq[c - 1] else mismatch)
            opened, longer = (row[c - 1], across) if r == la else (row[c - 1] + gap_open, across + gap_extend)
            across = opened if opened >= longer else longer
            opened_d, longer_d = (above[c], down[c]) if c == lb else (above[c] + gap_open, down[c] + gap_extend)
            down[c] = opened_d if opened_d >= longer_d else longer_d
            top = max(pair, across, down[c])
            row[c] = top if top > 0 else 0
            stretched[r][c] = (across == top and longer == across) or (down[c] == top and longer_d == down[c])
    return score, stretched


def local_pair(target, seq, match=2, mismatch=-1, gap_open=-20, gap_extend=-20):
    """The first alignment `pairwise2.align.localms(target, seq, 2, -1, -20, -20)` lists, as the
    two sequences padded with '-' (W2C:109-111 reads [0][0] and [0][1]).  Biopython's order
    (releases 1.70-1.76, the ones a Python-2.7 miRge2.0 runs with) decides between alignments of
    equal score, and this follows it: candidates are the best-scoring cells that end in a pair,
    tried from the LARGEST (target position, read position) down; from a cell the walk back
    prefers a gap in the target, then a pair, then a gap in the read, depth first; a walk that
    meets another best-scoring cell is a zero-score extension of that cell's alignment and is
    dropped, as is a gap in the read placed right behind a gap in the target.  With -20 per
    gap base almost every read lands on one ungapped diagonal; the rule shows on low-complexity
    miRNAs (several diagonals tie: the one ending furthest along the miRNA wins) and on merged
    families whose members differ by an indel (one gap beats the best ungapped run).
    tests/golden/a2i.json pins it against the generator's own restatement (make_golden.py:
    pairwise2_localms, no product code)."""
    la, lb = len(target), len(seq)
    if la == 0 or lb == 0:
        raise ValueError("local_pair: empty sequence")
    score, stretched = _local_scores(target, seq, match, mismatch, gap_open, gap_extend)
    best = max(max(row) for row in score)
    tops = [(r, c) for r in range(la + 1) for c in range(lb + 1) if score[r][c] == best]
    top_set = set(tops)

    def pairs_here(r, c):
        return score[r - 1][c - 1] + (match if target[r - 1] == seq[c - 1] else mismatch) == score[r][c]

    candidates = []
    for (r, c) in tops:
        if (r - 1, c - 1) in top_set:
            break
        if best > 0 and pairs_here(r, c):
            candidates.append((r, c))

    def walk(r, c, cols, read_gap_last, first):
        """cols: alignment columns collected back to front as (target char, read char)."""
        while True:
            if not first:
                if (r, c) in top_set:
                    return None
                if r == 0 or c == 0 or score[r][c] <= 0:
                    if c and read_gap_last:
                        return None
                    return cols, r, c
            here = score[r][c]
            if stretched[r][c] and not first:
                raise ValueError("local_pair: alignment through a gap of two or more (%r, %r)" % (target, seq))
            moves = []
            if not first and not read_gap_last and r < la and score[r][c - 1] + gap_open == here:
                moves.append((r, c - 1, ("-", seq[c - 1]), False))
            if pairs_here(r, c):
                moves.append((r - 1, c - 1, (target[r - 1], seq[c - 1]), False))
            if not first and c < lb and score[r - 1][c] + gap_open == here:
                moves.append((r - 1, c, (target[r - 1], "-"), True))
            if not moves:
                return None
            for (r2, c2, col, flag) in moves[:-1]:
                got = walk(r2, c2, cols + [col], flag, False)
                if got is not None:
                    return got
            r, c, col, read_gap_last = moves[-1]
            cols = cols + [col]
            first = False

    for (r, c) in reversed(candidates):
        got = walk(r, c, [], False, True)
        if got is None:
            continue
        cols, r0, c0 = got
        body_t = "".join(x for x, _ in reversed(cols))
        body_s = "".join(y for _, y in reversed(cols))
        head = max(r0, c0)
        tail = max(la - r, lb - c)
        tpad = "-" * (head - r0) + target[:r0] + body_t + target[r:] + "-" * (tail - (la - r))
        spad = "-" * (head - c0) + seq[:c0] + body_s + seq[c:] + "-" * (tail - (lb - c))
        return tpad, spad
    raise ValueError("local_pair: no local alignment of %r and %r" % (target, seq))


def dash_count(s):
    head = len(s) - len(s.lstrip("-"))
    tail = len(s) - len(s.rstrip("-")) if head < len(s) else 0
    return head, tail


def remove_dash(s):
    return s.strip("-")


def judge_align(target_pad, seq_pad):
    """judgeAllign (W2C:35-69): the read may start at most 1 nt after the miRNA, and over the
    miRNA minus its last 3 nt it may show at most 1 mismatch and must show enough matches."""
    mismatch_limit, head_shift, tail_shift = 1, 1, 3
    head_t, tail_t = dash_count(target_pad)
    head_s, tail_s = dash_count(seq_pad)
    len1 = len(target_pad) - head_t - tail_t
    match_limit = len1 - tail_shift - mismatch_limit
    end1 = len(target_pad) - head_t - 1 - tail_shift
    end2 = len(seq_pad) - tail_s - 1
    if head_s - head_t > head_shift:
        return False
    mism = mat = 0
    for pos in range(head_t, min(end1, end2) + 1):
        if seq_pad[pos] == "-":
            continue
        if target_pad[pos] != seq_pad[pos]:
            mism += 1
        else:
            mat += 1
    need = match_limit - head_shift if head_s - head_t == head_shift else match_limit
    return not (mism > mismatch_limit or mat < need)


def align_to_target(target, seqs):
    """align2TargetSeq (W2C:101-143): pad everything into one frame; returns
    ([target_pad, read_pad...], [judge state...])."""
    frame, states = [], []
    for seq in seqs:
        tpad, spad = local_pair(target, seq)
        states.append(judge_align(tpad, spad))
        if not frame:
            frame = [tpad, spad]
        elif tpad == frame[0]:
            frame.append(spad)
        else:
            h1 = tpad.index(target)
            t1 = len(tpad) - h1 - len(target)
            h2 = frame[0].index(target)
            t2 = len(frame[0]) - h2 - len(target)
            if h1 >= h2:
                frame = ["-" * (h1 - h2) + x for x in frame]
            else:
                spad = "-" * (h2 - h1) + spad
            if t1 >= t2:
                frame = [x + "-" * (t1 - t2) for x in frame]
            else:
                spad = spad + "-" * (t2 - t1)
            frame.append(spad)
    return frame, states


def binom_cdf(k, n, p):
    """P[X <= k], X ~ Binomial(n, p) (scipy.stats.binom.cdf, W2C:220)."""
    from scipy import stats
    return float(stats.binom.cdf(k, n, p))


def a2i_editing(target, seqs, counts, mir_name, detail, retained, start_base="A", end_base="G"):
    """A2IEditing (W2C:145-229)."""
    frame, states = align_to_target(target, seqs)
    tpad = frame[0]
    start = tpad.index(target)
    end = start + len(target) - 1
    head = dash_count(tpad)[0]
    canonical = count_true = seq_true = 0
    pos_count, positions, kept = {}, [], []
    for j, spad in enumerate(frame[1:]):
        if states[j] and remove_dash(spad) in retained:
            if seqs[j] in target:
                canonical += counts[j]
            kept.append(spad)
            seq_true += 1
            count_true += counts[j]
            for i in range(start, end + 1 - TAIL_SHIFT):
                if i < len(spad) and tpad[i] == start_base and spad[i] == end_base:
                    p = i + 1 - head
                    if p not in pos_count:
                        positions.append(p)
                        pos_count[p] = counts[j]
                    else:
                        pos_count[p] += counts[j]
    detail.write("Canonical_Seq of %s: %s\n" % (mir_name, target))
    detail.write("seqList size is: %d, %d\n" % (len(seqs), len(counts)))
    for k, spad in enumerate(frame[1:]):
        detail.write("\t".join([spad, str(counts[k]), str(states[k])]) + "\n")
    detail.write("****************\n")
    detail.write("retained seqList size is: %d\n" % seq_true)
    for k, spad in enumerate(frame[1:]):
        if states[k]:
            detail.write("\t".join([spad, str(counts[k]), str(states[k])]) + "\n")
    detail.write("****************\n")
    detail.write("retained sequences after filering are:\n")
    for k, spad in enumerate(frame[1:]):
        if states[k] and remove_dash(spad) in retained:
            detail.write("\t".join([spad, str(counts[k]), str(states[k])]) + "\n")
    ratio, pval = {}, {}
    for p in positions:
        ratio[p] = float(pos_count[p]) / count_true if count_true else 0
        rest = count_true - pos_count[p]
        pval[p] = binom_cdf(rest, count_true, 1 - P_MISMATCH) if rest >= 0 else 1.0
    return kept, positions, pos_count, ratio, pval, count_true, seq_true, canonical


def refine_name(name):
    """refineName (W2C:75-86): drop the first '.fastq'."""
    at = name.find(".fastq")
    return name if at < 0 else name[:at] + name[at + len(".fastq"):]


def _check_seq_list(seq_pads, seqDic):
    """checkSeqList (W2C:88-99): is at least one kept read an exact (pass-0) miRNA read?"""
    for spad in seq_pads:
        rec = seqDic.get(remove_dash(spad))
        if rec is not None and rec["annot"][1] != "":
            return True
    return False


# ------------------------------------------------------------------ driver
def a_to_i_report(outputdir, sampleList, logDic, seqDic, mirDic, mirNameSeqDic, mirMergedNameDic,
                  removedMiRNAList, genome, start_base="A", end_base="G"):
    """The `if a_to_i:` block of writeDataToCSV (W2C:1221-1594).  Writes
    a2IEditing.detail.txt, a2IEditing.report.csv, a2IEditing.report.newform.csv and returns
    the rows of the final report (header included)."""
    S = len(sampleList)

    def rpm(count, i):
        return 1000000.0 * count / logDic["quantStats"][i]["mirnaReadsFiltered"]

    groups = {}
    for seq, rec in seqDic.items():
        annot = rec["annot"]
        if annot[1] == "" and annot[9] == "":
            continue
        name = annot[1] if annot[1] != "" else annot[9]
        name = mirMergedNameDic.get(name, name)
        if annot[1] != "" or any(rpm(rec["quant"][i], i) >= 1 for i in range(S)):
            groups.setdefault(name, []).append(seq)

    retained = genome.unique_best([s for seqs in groups.values() for s in seqs])

    site_values, site_order = {}, []
    with open(os.path.join(outputdir, "a2IEditing.detail.txt"), "w") as detail:
        for name, seqs in groups.items():
            target = mirNameSeqDic[name]
            for i in range(S):
                if not rpm(mirDic[name]["quant"][i], i) >= 1:
                    continue
                sel = [s for s in seqs if rpm(seqDic[s]["quant"][i], i) >= 1 or
                       (seqDic[s]["annot"][1] != "" and seqDic[s]["quant"][i] > 0)]
                cnt = [seqDic[s]["quant"][i] for s in sel]
                if len(cnt) > 1 and min(cnt) > 0:
                    kept, positions, pos_count, ratio, pval, count_true, seq_true, canonical = \
                        a2i_editing(target, sel, cnt, name, detail, retained, start_base, end_base)
                    for p in positions:
                        site = "%s:%s" % (name, p)
                        if site not in site_values:
                            site_order.append(site)
                            site_values[site] = [[] for _ in range(S)]
                        site_values[site][i] = site_values[site][i] + [
                            kept, str(sum(cnt)), str(len([c for c in cnt if c > 0])), str(count_true),
                            str(seq_true), str(canonical), str(pos_count[p]), ratio[p], pval[p]]

    for i in range(S):  # W2C:1353-1364
        ranked = sorted([site_values[s][i][8], s] for s in site_order if len(site_values[s][i]) != 0)
        for rank, (p, site) in enumerate(ranked):
            site_values[site][i].append(p * len(ranked) / (rank + 1))

    header = "miRNA,A-to-I position in the miRNA,miRNA sequence"
    for s in sampleList:
        n = refine_name(s)
        header += "," + ",".join([n + ".readCount", n + ".readCount.canonical", n + ".RPM.canonical",
                                  n + ".readCount.mismatch", n + ".RPM.mismatch", n + ".AtoI.percentage",
                                  n + ".AtoI.adjusted.pValue"])
    rows = []
    for site in site_order:
        name, pos = site.split(":")[0].strip(), site.split(":")[1].strip()
        cells = [name, pos, mirNameSeqDic[name]]
        for i in range(S):
            v = site_values[site][i]
            if len(v) == 0:
                cells += ["NE"] * 7
                continue
            base = [v[3], v[5], "%.2f" % rpm(int(v[5]), i), v[6], "%.2f" % rpm(int(v[6]), i)]
            if _check_seq_list(v[0], seqDic):
                base.append("%.2f%%" % (v[7] * 100))
                base.append("%.2E" % v[9] if v[9] <= 0.05 else "NS")
            else:
                base += ["NE", "NE"]
            cells += base
        rows.append(cells)

    def is_float(x):
        try:
            float(x)
            return True
        except ValueError:
            return False
    # keep rows with a significant sample (W2C:1423-1442), sort by name then position (:1444)
    rows = [r for r in rows if any(is_float(r[9 + i * 7]) for i in range((len(r) - 9) // 7 + 1))]
    rows.sort(key=lambda r: (r[0], int(r[1])))

    edited = {}
    for r in rows:  # W2C:1450-1480
        pos = int(r[1])
        edited_seq = "".join(end_base if k == pos - 1 else ch for k, ch in enumerate(r[2]))
        canon_rpm = [float(r[5 + i * 7]) if is_float(r[5 + i * 7]) else 0 for i in range((len(r) - 9) // 7 + 1)]
        edited[(r[0], r[1])] = dict(seq=edited_seq, rpm_drop=not any(x >= 1 for x in canon_rpm),
                                    repeat_drop=r[0] in removedMiRNAList)
    in_genome = genome.exact_hit([e["seq"] for e in edited.values()])
    final = [r for r in rows if not (edited[(r[0], r[1])]["rpm_drop"] or edited[(r[0], r[1])]["repeat_drop"] or
                                     edited[(r[0], r[1])]["seq"] in in_genome)]
    with open(os.path.join(outputdir, "a2IEditing.report.csv"), "w") as out:
        out.write(header + "\n")
        for r in final:
            out.write(",".join(r) + "\n")

    # a2IEditing.report.newform.csv (W2C:1554-1605)
    samples = []
    for item in header.split(","):
        if ".AtoI.percentage" in item and item.split(".")[0] not in samples:
            samples.append(item.split(".")[0])
    kept_sites = []
    for r in final:
        site = ":".join(r[:2])
        cells = r[3:]
        per = []
        for k in range(0, len(cells), 7):
            item = cells[k:k + 7]
            per.append(("NA", "NA") if item[6] in ("NE", "NS") else
                       (item[5][:-1], py2_float_str(math.log(float(item[4]), 2))))
        if any(x[0] != "NA" and float(x[0]) >= PERCENT_CUTOFF for x in per):
            kept_sites.append((site, per))
    ranked = sorted(((sum(1 for x in per if x[0] != "NA"), site) for site, per in kept_sites), reverse=True)
    by_site = dict(kept_sites)
    with open(os.path.join(outputdir, "a2IEditing.report.newform.csv"), "w") as out:
        out.write("miRNA:position,sample,A-to-I percentage,log2RPM\n")
        for k, sample in enumerate(samples):
            for _, site in ranked:
                out.write(site + "," + SAMPLE_LABELS.get(sample, sample) + "," + ",".join(by_site[site][k]) + "\n")
    return [header.split(",")] + final


# ------------------------------------------------------------------ genome filters on the GPU
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(_COMP.get(ch, "N") for ch in reversed(s))


class EngineGenome:
    """The two genome bowtie runs of the -ai path on the GPU engine: the genome (one or several
    libraries, e.g. one per chromosome) is searched with the reads and with their reverse
    complements (no --norc in W2C:1263/:1488), after dropping the last 2 nt (-3 2)."""

    def __init__(self, engine, library_keys):
        self.engine = engine
        self.keys = list(library_keys)

    def _best(self, reads, max_mm_seed):
        import numpy as np
        from . import pack
        from .engine import ReadSet
        trimmed = [r[:-2] for r in reads]
        both = trimmed + [revcomp(r) for r in trimmed]
        n = len(reads)
        best_mm = np.full(n, 255, dtype=np.int32)
        count = np.zeros(n, dtype=np.int64)
        if n == 0:
            return best_mm, count
        words, lens, nmask = pack.pack_reads(both)
        rs = ReadSet(words, lens, nmask, None, device=self.engine.device)
        for key in self.keys:
            mm, cnt = self.engine.count_best(rs, key, seed_len=28, max_mm_seed=max_mm_seed, max_mm_total=2)
            for half in (slice(0, n), slice(n, 2 * n)):
                m, c = mm[half].astype(np.int32), cnt[half].astype(np.int64)
                better = m < best_mm
                same = (m == best_mm) & (m < 255)
                count = np.where(better, c, np.where(same, count + c, count))
                best_mm = np.where(better, m, best_mm)
        return best_mm, count

    def unique_best(self, reads):
        reads = list(dict.fromkeys(reads))
        mm, cnt = self._best(reads, 1)
        return {r for r, m, c in zip(reads, mm, cnt) if m < 255 and c == 1}

    def exact_hit(self, reads):
        reads = list(dict.fromkeys(reads))
        mm, _ = self._best(reads, 0)
        return {r for r, m in zip(reads, mm) if m < 255}
