"""tRF typing and tables for `-trf`.

Restates, on small host data:
  * the `-trf` table loading of __main__.py:164-251 (`load_trf_tables`);
  * `trfTypes` (runAnnotationPipeline.py:448-474) and the bookkeeping of
    `updatetrfContentDic2/3` (RAP:497-541) over the `-a --best --strata` listings that
    `parseAlignment3` (RAP:41-52) collects after the mature-tRNA pass (RAP:657-660) and the
    primary-tRNA-trailer pass (RAP:698-701);
  * the first part of writeDataToCSV's `if trf_output:` block (W2C:648-800):
    `tRFs.potential.report.tsv`, `discarded.reads.summary.assigningtRFs.csv`, `tRF.Counts.csv`,
    `tRF.RP100K.csv`, with `addDashNew`, `coordinate`, `getDistance2`, `assign_cluster`
    (W2C:353-392, :546-564).

The listings come from the GPU (`Engine.list_best` = mrg_list_best_count/fill); tests can pass
any `lister(reads, library_key, max_mismatches) -> [[(entry name, 0-based offset), ...], ...]`.

Not built (SURVEY.md section 2 row 9 marks it out of scope): the per-sample
`tRFs.samples.tmp/*` reports and their density-peak clustering (W2C:802-1088).

Where the reference is not deterministic it is pinned here:
  * `random.choice(candidatetRNAUniquelist)` (W2C:708) -> `choose`, default the
    lexicographically smallest name;
  * `list(set(...))` order of the de-duplicated hit columns (W2C:704) -> sorted.
"""
import os
import sys

from .isomir import NT2CODE, make_id


# ---------------------------------------------------------------- tables (MAIN:164-251)
def add_dash_new(seq, total_length, start, end):
    """W2C:353-355 (1-based inclusive start/end)."""
    return "-" * (start - 1) + seq + "-" * (total_length - end)


def load_trf_tables(library_path, species):
    """MAIN:164-251.  Returns a dict with trnaStruDic, trnaAAanticodonDic, duptRNA2UniqueDic,
    tRNAtrfDic, trfMergedNameDic, trfMergedList; a missing file prints the reference's message
    and exits 1."""
    ann = os.path.join(library_path, species, "annotation.Libs")

    def need(path):
        if not os.path.isfile(path):
            print("%s does not exsit. Please check it." % path, file=sys.stderr)
            sys.exit(1)
        return path

    stru = {}
    with open(need(os.path.join(ann, species + "_trna.str"))) as fh:
        line = fh.readline()
        while line != "":
            name = line.strip()[1:]
            seq = fh.readline().strip()
            st = fh.readline().strip()
            a = st.index("XXX") + 1          # 1-based
            stru[name] = {"seq": seq, "stru": st, "anticodonStart": a, "anticodonEnd": a + 2}
            line = fh.readline()
    aa = {}
    with open(need(os.path.join(ann, species + "_trna_aminoacid_anticodon.csv"))) as fh:
        for line in fh:
            f = line.strip().split(",")
            aa[f[0]] = {"aaType": f[1], "anticodon": f[2]}
    dup = {}
    with open(need(os.path.join(ann, species + "_trna_deduplicated_list.csv"))) as fh:
        fh.readline()
        for line in fh:
            if line == "":
                break
            f = line.strip().split(",")
            for item in f[1].split("/"):
                dup[item.strip()] = f[0].strip()
    clusters = {}
    with open(need(os.path.join(ann, species + "_tRF_infor.csv"))) as fh:
        fh.readline()
        for line in fh:
            f = line.strip().split(",")
            trna = f[0].split("_Cluster")[0]
            start, end = int(f[3].split("-")[0]), int(f[3].split("-")[1])
            clusters.setdefault(trna, {})[add_dash_new(f[4], len(f[5]), start, end)] = f[0]
    merged_name, merged_list = {}, []
    with open(need(os.path.join(ann, species + "_tRF_merges.csv"))) as fh:
        for line in fh:
            f = line.strip().split(",")
            merged_list.append(f[0])
            for item in f[1].split("/"):
                merged_name[item] = f[0]
    return {"trnaStruDic": stru, "trnaAAanticodonDic": aa, "duptRNA2UniqueDic": dup, "tRNAtrfDic": clusters,
            "trfMergedNameDic": merged_name, "trfMergedList": merged_list}


# ---------------------------------------------------------------- typing (RAP:448-541)
def trfTypes(seq, tRNAName, start, trnaStruDic, pretrnaNameSeqDic):
    """RAP:448-474; `start` 0-based."""
    if "pre_" in tRNAName:
        pretrnaNameSeqDic[tRNAName]          # the reference looks it up (KeyError if unknown)
        return "tRF-1"
    rec = trnaStruDic[tRNAName]
    n = len(rec["seq"])
    anticodon = rec["anticodonStart"] - 1
    last = start + len(seq) - 1
    if start == 0:
        if start + len(seq) == n:
            return "tRF-whole"
        if anticodon - 2 <= last <= anticodon + 1:
            return "5'-half"
        return "5'-tRF"
    if n - 3 <= last <= n - 1:
        if anticodon - 1 <= start <= anticodon + 2:
            return "3'-half"
        return "3'-tRF"
    return "i-tRF"


def update_trf_content_mature(trfContentDic, hits, trnaStruDic, pretrnaNameSeqDic, seqDic, sampleList):
    """updatetrfContentDic2 (RAP:497-516).  `hits` = {read: [(tRNA name, 0-based offset), ...]} in
    the order the SAM lists them (a later line for the same tRNA overwrites an earlier one)."""
    S = len(sampleList)
    for read, items in hits.items():
        rec = {"count": [int(seqDic[read]["quant"][i]) for i in range(S)], "uid": make_id(read, NT2CODE)}
        trfContentDic[read] = rec
        for name, start in items:
            rec[name] = {"tRFType": trfTypes(read, name, start, trnaStruDic, pretrnaNameSeqDic), "start": start,
                         "end": start + len(read) - 1, "cigar": "undifined"}


def update_trf_content_trailer(trfContentDic, hits, trnaStruDic, pretrnaNameSeqDic, seqDic, sampleList,
                               subseqSeqDic):
    """updatetrfContentDic3 (RAP:518-541).  `hits` is keyed by the T-stripped subsequence; every
    original read of subseqSeqDic[subsequence] gets the entry."""
    S = len(sampleList)
    for sub, items in hits.items():
        for read in subseqSeqDic[sub]:
            rec = {"count": [int(seqDic[read]["quant"][i]) for i in range(S)], "uid": make_id(read, NT2CODE)}
            trfContentDic[read] = rec
            tail_t = len(read) - len(read.rstrip("T"))
            for name, start in items:
                rec[name] = {"tRFType": trfTypes(read, name, start, trnaStruDic, pretrnaNameSeqDic),
                             "start": start, "end": start + len(read) - 1 - tail_t, "cigar": "undifined"}


def engine_lister(engine):
    """lister backed by the GPU: `-v V -a --best --strata --norc` against one library."""
    from . import pack
    from .engine import V_MODE_SEED, ReadSet

    def lister(reads, library_key, max_mismatches):
        if not reads:
            return []
        words, lens, nmask = pack.pack_reads(reads)
        rs = ReadSet(words, lens, nmask, None, device=engine.device)
        _, off, ref, pos = engine.list_best(rs, library_key, seed_len=V_MODE_SEED, max_mm_seed=max_mismatches,
                                            max_mm_total=max_mismatches)
        names = engine.indexes[library_key].names
        return [[(names[int(ref[k])], int(pos[k])) for k in range(int(off[i]), int(off[i + 1]))]
                for i in range(len(reads))]
    return lister


def strip_poly_t(read):
    """RAP:671-680: a read ending in >= 3 T, its T tail removed, if >= 11 nt remain; else None."""
    sub = read.rstrip("T")
    if len(read) - len(sub) >= 3 and len(sub) >= 11:
        return sub
    return None


def collect_trf_content(trfContentDic, seqDic, sampleList, trnaStruDic, pretrnaNameSeqDic, lister,
                        mature_key="mature_trna", pre_key="pre_trna"):
    """The `-trf` side products of the cascade (RAP:657-660, :698-701) after the cascade has set
    `annot`: reads claimed by pass 2 / pass 3 are listed against their library and typed.  Listings
    are consumed highest (entry, offset) first, so that for a tRNA hit twice by one read the lowest
    offset is kept -- the order in which the engine's tie rule would list them last."""
    mature = [s for s, r in seqDic.items() if r["annot"][3] != ""]
    lists = lister(mature, mature_key, 1)
    update_trf_content_mature(trfContentDic, {s: l[::-1] for s, l in zip(mature, lists)}, trnaStruDic,
                              pretrnaNameSeqDic, seqDic, sampleList)
    sub_reads = {}
    for s, r in seqDic.items():
        if r["annot"][4] != "":
            sub_reads.setdefault(strip_poly_t(s), []).append(s)
    subs = list(sub_reads)
    lists = lister(subs, pre_key, 0)
    update_trf_content_trailer(trfContentDic, {s: l[::-1] for s, l in zip(subs, lists)}, trnaStruDic,
                               pretrnaNameSeqDic, seqDic, sampleList, sub_reads)


# ---------------------------------------------------------------- tables (W2C:357-392, :546-564)
def coordinate(dashed):
    """W2C:357-371: 1-based first and last non-dash positions."""
    lead = len(dashed) - len(dashed.lstrip("-"))
    trail = len(dashed) - len(dashed.rstrip("-"))
    return lead + 1, len(dashed) - trail


def get_distance2(seq1, seq2):
    """W2C:373-392: |start shift| + |end shift| + substitutions (a base of seq1 past the end of
    seq2 counts as one: the reference's IndexError branch, reached only when seq1[k] is a base)."""
    c1, c2 = coordinate(seq1), coordinate(seq2)
    subst = 0
    for k in range(len(seq1)):
        if seq1[k] == "-":
            continue
        if k >= len(seq2) or (seq2[k] != "-" and seq1[k] != seq2[k]):
            subst += 1
    return abs(c1[0] - c2[0]) + abs(c1[1] - c2[1]) + subst


def assign_cluster(dashedSeq, tRNAName, tRNAtrfDic):
    """W2C:546-564."""
    if tRNAName not in tRNAtrfDic:
        return "Dele", 100, "Null"
    dis = sorted((get_distance2(dashedSeq, s), name) for s, name in tRNAtrfDic[tRNAName].items())
    if dis[0][0] <= 8.0:
        return dis[0][1], dis[0][0], dis[0][1]
    return "Undef", dis[0][0], dis[0][1]


def write_trf_tables(outputdir, sampleList, logDic, trfContentDic, tables, pretrnaNameSeqDic, choose=min):
    """W2C:648-800.  Mutates trfContentDic like the reference: adds 'RPM', keeps only the selected
    tRNA of every read, drops reads without a candidate."""
    stru, aa_dic = tables["trnaStruDic"], tables["trnaAAanticodonDic"]
    dup, clusters = tables["duptRNA2UniqueDic"], tables["tRNAtrfDic"]
    merged_name, merged_list = tables["trfMergedNameDic"], tables["trfMergedList"]
    S = len(sampleList)
    meta = ("uid", "RPM", "count")
    for rec in trfContentDic.values():
        rpm = []
        for i in range(S):
            q = logDic["quantStats"][i]
            denom = q["maturetrnaReads"] + q["pretrnaReads"]
            rpm.append(100000.0 * rec["count"][i] / denom if denom else 0.0)
        rec["RPM"] = rpm

    def info(read, name):
        e = trfContentDic[read][name]
        return ":".join([name, e["tRFType"], str(e["start"] + 1), str(e["end"] + 1)])

    def aa_of(name, type_from):
        t = aa_dic[type_from]["aaType"]
        return "pre:" + t if "pre_" in name else t

    rows = []
    trf_file = os.path.join(outputdir, "tRFs.potential.report.tsv")
    with open(trf_file, "w") as out:
        out.write("read sequence\tuid\tread count(%s)\tRP100K (%s)\tamino acid all hits\tamino acid-anticodon all "
                  "hits\ttRF information all hits\tamino acid all deduplicated hits\tamino acid-anticodon all "
                  "deduplicated hits\ttRF information all deduplicated hits\tamino acid one hit\tamino "
                  "acid-anticodon one hit\ttRF information one hit\n" % (",".join(sampleList), ",".join(sampleList)))
        for read in list(trfContentDic.keys()):
            rec = trfContentDic[read]
            infos, aa_anticodons, aa_types, candidates = [], [], [], []
            for name in rec:
                if name in meta:
                    continue
                infos.append(info(read, name))
                candidates.append(name)
                t = aa_of(name, name)
                if t not in ("Und", "pre:Und"):
                    pair = t + "-" + aa_dic[name]["anticodon"]
                    if pair not in aa_anticodons:
                        aa_anticodons.append(pair)
                    if t not in aa_types:
                        aa_types.append(t)
            unique = sorted(set(dup.get(c, c) for c in candidates))
            if not unique:
                del trfContentDic[read]
                continue
            selected = choose(unique)
            cols = [read, rec["uid"], ",".join(str(c) for c in rec["count"]),
                    ",".join("%.3f" % round(v, 3) for v in rec["RPM"]),
                    ",".join(aa_types), ",".join(aa_anticodons), ",".join(infos)]
            # W2C:727: the amino-acid type of every de-duplicated hit is read from the SELECTED tRNA
            d_types = [aa_of(t, selected) for t in unique]
            d_pairs = [a + "-" + aa_dic[t]["anticodon"] for a, t in zip(d_types, unique)]
            cols += [",".join(d_types), ",".join(d_pairs), ",".join(info(read, t) for t in unique)]
            one_type = aa_of(selected, selected)
            cols += [one_type, one_type + "-" + aa_dic[selected]["anticodon"], info(read, selected)]
            out.write("\t".join(cols) + "\n")
            rows.append((read, rec["count"], [float("%.3f" % round(v, 3)) for v in rec["RPM"]], selected,
                         rec[selected]["start"] + 1, rec[selected]["end"] + 1))
            for name in list(rec.keys()):
                if name not in meta and name != selected:
                    del rec[name]
    # W2C:749-800: counts per predefined tRF entity
    entity = {s: {m: [0, 0.0] for m in merged_list} for s in sampleList}
    summary = {s: [0, 0] for s in sampleList}
    for read, counts, rp100k, name, start, end in rows:
        length = len(pretrnaNameSeqDic[name]) if "pre" in name else len(stru[name]["seq"])
        assigned, _, _ = assign_cluster(add_dash_new(read, length, start, end), name, clusters)
        for i, s in enumerate(sampleList):
            summary[s][1] += counts[i]
            if assigned not in ("Undef", "Dele"):
                m = merged_name[assigned]
                entity[s][m][0] += counts[i]
                entity[s][m][1] += rp100k[i]
            else:
                summary[s][0] += counts[i]
    with open(os.path.join(outputdir, "discarded.reads.summary.assigningtRFs.csv"), "w") as out:
        out.write("sample name,percentage of discarded reads,details\n")
        for s in sampleList:
            d, t = summary[s]
            pct = round(float(d) / t * 100.0, 2) if t else 0.0
            out.write("%s,%.2f%%,%d\\%d\n" % (s, pct, d, t))
    with open(os.path.join(outputdir, "tRF.Counts.csv"), "w") as o1, \
            open(os.path.join(outputdir, "tRF.RP100K.csv"), "w") as o2:
        o1.write("entry name," + ",".join(sampleList) + "\n")
        o2.write("entry name," + ",".join(sampleList) + "\n")
        for m in merged_list:
            o1.write(m + "," + ",".join(str(entity[s][m][0]) for s in sampleList) + "\n")
            o2.write(m + "," + ",".join("%.2f" % round(entity[s][m][1], 2) for s in sampleList) + "\n")
    return rows
