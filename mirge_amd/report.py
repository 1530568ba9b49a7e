"""Output tables of the annotate path.

  mapped.csv / unmapped.csv        writeDataToCSV.py:582-619, :1172-1188
  miR.Counts.csv / miR.RPM.csv     writeDataToCSV.py:1190-1219
  isomirs.csv / isomirs.samples.csv (-di)   writeDataToCSV.py:582-619 (grouping),
                                            :1090-1170 (entropy), calcEntropy :344-351
  annotation.report.csv            generateReport.py:104-108, :191-200

Host-side: these are M-sized or N-row text tables, streamed row by row (no dict
copy), so they also work from the columnar arrays at 10^8 rows.

Float formatting: the reference runs on Python 2, whose str(float) keeps 12
significant digits (`'%.12g'`, with '.0' appended to integral values);
`py2_float_str` reproduces that so RPM / entropy columns are byte-compatible.
Row order of mapped/unmapped/isomirs tables follows dict iteration in the
reference (arbitrary under Python 2); here it is insertion order.
"""
import math
import os


def py2_float_str(x):
    """str(float) as Python 2.7 prints it."""
    x = float(x)
    if x != x:
        return "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    s = "%.12g" % x
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s


def calc_entropy(values):
    """calcEntropy, writeDataToCSV.py:344-351: Shannon entropy (bits) over the
    entries > 1 (sic), each weighted by its share of the FULL sum."""
    total = sum(values)
    h = 0
    for v in values:
        if v > 1:
            f = float(v) / total
            h = h + -1 * f * math.log(f, 2)
    return h


def _annot_width(spikeIn):
    return 11 if spikeIn else 10


def write_mapped_csv(path, annotNameList, sampleList, seqDic, spikeIn=False):
    """mapped.csv (writeDataToCSV.py:582-619).  Returns the isomirDic the reference
    builds in the same loop: {miRNA (SNP suffix stripped): {'mirnas': {seq: counts},
    'isomirs': {seq: counts}}}."""
    isomirDic = {}
    width = _annot_width(spikeIn)
    with open(path, "w") as out:
        out.write("uniqueSequence,annotFlag," + ",".join(annotNameList) + "," + ",".join(sampleList) + "\n")
        for seq, rec in seqDic.items():
            annot = rec["annot"]
            if not annot[0] > 0:
                continue
            isomir, mirna = annot[9], annot[1]
            if isomir != "" or mirna != "":
                key, kind = (isomir, "isomirs") if isomir != "" else (mirna, "mirnas")
                if ".SNP" in key:
                    key = key.split(".SNP")[0]
                slot = isomirDic.setdefault(key, {"mirnas": {}, "isomirs": {}})
                slot[kind][seq] = [rec["quant"][i] for i in range(len(sampleList))]
            row = [seq] + [str(annot[i]) for i in range(width)] + \
                  [str(rec["quant"][i]) for i in range(len(sampleList))]
            out.write(",".join(row) + "\n")
    return isomirDic


def write_unmapped_csv(path, annotNameList, sampleList, seqDic, spikeIn=False):
    """unmapped.csv (writeDataToCSV.py:1172-1188)."""
    width = _annot_width(spikeIn)
    with open(path, "w") as out:
        out.write("uniqueSequence,annotFlag," + ",".join(annotNameList))
        for s in sampleList:
            out.write("," + s)
        out.write("\n")
        for seq, rec in seqDic.items():
            if rec["annot"][0] == 0:
                out.write(seq)
                for i in range(width):
                    out.write("," + str(rec["annot"][i]))
                for i in range(len(sampleList)):
                    out.write("," + str(rec["quant"][i]))
                out.write("\n")


def write_counts_csv(path, sampleList, mirDic, logDic):
    """miR.Counts.csv (writeDataToCSV.py:1190-1205): miRNAtotal row, then names sorted."""
    with open(path, "w") as out:
        out.write("miRNA" + "".join("," + s for s in sampleList) + "\n")
        out.write("miRNAtotal" + "".join(
            "," + str(logDic["quantStats"][i]["mirnaReadsFiltered"]) for i in range(len(sampleList))) + "\n")
        for name in sorted(mirDic.keys()):
            out.write(name + "".join("," + str(mirDic[name]["quant"][i]) for i in range(len(sampleList))) + "\n")


def write_rpm_csv(path, sampleList, mirDic, logDic):
    """miR.RPM.csv (writeDataToCSV.py:1207-1219)."""
    with open(path, "w") as out:
        out.write("miRNA" + "".join("," + s for s in sampleList) + "\n")
        for name in sorted(mirDic.keys()):
            out.write(name)
            for i in range(len(sampleList)):
                tot = logDic["quantStats"][i]["mirnaReadsFiltered"]
                if tot > 0:
                    out.write("," + py2_float_str(1000000.0 * mirDic[name]["quant"][i] / tot))
                else:
                    out.write(",0")
            out.write("\n")


def write_isomir_tables(isomir_path, sample_path, sampleList, isomirDic, logDic):
    """isomirs.csv + isomirs.samples.csv (writeDataToCSV.py:1090-1170)."""
    S = len(sampleList)
    with open(isomir_path, "w") as f1, open(sample_path, "w") as f2:
        f1.write("miRNA,sequence" + "".join("," + s for s in sampleList) + ",Entropy\n")
        f2.write("miRNA")
        for s in sampleList:
            f2.write("," + s + " isomir+miRNA Entropy")
            f2.write("," + s + " Canonical Sequence")
            f2.write("," + s + " Canonical RPM")
            f2.write("," + s + " Top Isomir RPM")
        f2.write("\n")
        # (10^5..10^6 isomiR reads pass through this loop: the per-sample divisors are looked up once, and the one-sample
        # case -- entropy over one value is 0, its maximum log2(1) = 0: "NA" -- skips the entropy arithmetic)
        filtered = [logDic["quantStats"][i]["mirnaReadsFiltered"] for i in range(S)]
        lines = []
        for mirna, groups in isomirDic.items():
            per_sample_isomirs = {i: [] for i in range(S)}
            canon = [0] * S
            for counts in groups["mirnas"].values():
                for i in range(len(counts)):
                    canon[i] += counts[i]
            for seq, counts in groups["isomirs"].items():
                n_c = len(counts)
                if n_c == 1:
                    per_sample_isomirs[0].append(counts[0])
                    lines.append("%s,%s,%s,NA\n" % (mirna, seq, py2_float_str(counts[0] * 1000000.0 / filtered[0])))
                else:
                    for i in range(n_c):
                        per_sample_isomirs[i].append(counts[i])
                    h = calc_entropy(counts)
                    hmax = math.log(n_c, 2)
                    h_txt = "NA" if hmax == 0 else py2_float_str(h / hmax)
                    rpm = [py2_float_str(counts[i] * 1000000.0 / filtered[i]) for i in range(n_c)]
                    lines.append(",".join([mirna, seq] + rpm + [h_txt]) + "\n")
                if len(lines) >= 65536:
                    f1.write("".join(lines))
                    lines = []
            # isomirs.samples.csv: the reference appends to ONE row list across the
            # samples and writes it after each sample that has isomiRs (sic)
            row = [mirna]
            for lane in range(S):
                factor = 1000000.0 / logDic["quantStats"][lane]["mirnaReadsFiltered"]
                vals = per_sample_isomirs[lane]
                if len(vals) > 0:
                    top = max(vals) * factor
                    iso_sum = sum(vals) * factor
                    vals.append(canon[lane])
                    h_all = calc_entropy(vals)
                    canon_rpm = canon[lane] * factor
                    n = len(vals)
                    row.append(py2_float_str(h_all / math.log(n, 2)) if n > 1 else "NA")
                    combined = canon_rpm + iso_sum
                    row.append(py2_float_str(100.0 * canon_rpm / combined) if combined > 0 else "NA")
                    row.append(py2_float_str(canon_rpm))
                    row.append(py2_float_str(top))
                    f2.write(",".join(row) + "\n")
        f1.write("".join(lines))


ANNOTATION_REPORT_HEADER = ("File name(s),Total Input Reads,Trimmed Reads(all),Trimmed Reads(unique),"
                            "All miRNA Reads,Filtered miRNA Reads,Unique miRNAs,Hairpin miRNAs,"
                            "mature tRNA Reads,primary tRNA Reads,snoRNA Reads,rRNA Reads,ncRNA others,"
                            "mRNA Reads,Remaining Reads\n")


def write_annotation_report_csv(path, sampleList, logDic, spikeIn=False):
    """annotation.report.csv (generateReport.py:108, :191-200).  With -spikeIn the rows
    gain a spike-in column that the header does not name (reference quirk, kept)."""
    keys = ["totalReads", "trimmedReads", "trimmedUniq", "mirnaReads", "mirnaReadsFiltered",
            "mirnaUniqFiltered", "hairpinReads", "maturetrnaReads", "pretrnaReads", "snornaReads",
            "rrnaReads", "ncrnaOthersReads", "mrnaReads"]
    if spikeIn:
        keys.append("spikeInReads")
    keys.append("remReads")
    with open(path, "w") as out:
        out.write(ANNOTATION_REPORT_HEADER)
        for i, name in enumerate(sampleList):
            qs = logDic["quantStats"][i]
            out.write(name + "," + ",".join(str(qs[k]) for k in keys) + "\n")


def writeDataToCSV(outputdir, annotNameList, sampleList, isomirDiff, a_to_i, logDic, seqDic, mirDic,
                   mirNameSeqDic=None, mirMergedNameDic=None, spikeIn=False, gff_output=False,
                   isomiRContentDic=None, miRNA_database="miRBase", trf_output=False, genome=None,
                   removedMiRNAList=None, trfContentDic=None, trf_tables=None, pretrnaNameSeqDic=None):
    """The table-writing part of writeDataToCSV.py:566 (same leading arguments).  `genome`
    stands in for (bowtieBinary, genome_index): an object answering the two genome runs of
    the -ai block (mirge_amd.a2i.EngineGenome).  `trf_tables` = mirge_amd.trf.load_trf_tables(...)
    stands in for the six tRF table arguments, `pretrnaNameSeqDic` for (bowtieBinary,
    file_pre_tRNA).  Of the -trf branch the report / count tables are written (mirge_amd.trf),
    the per-sample clustering reports are not."""
    if trf_output and (trfContentDic is None or trf_tables is None or pretrnaNameSeqDic is None):
        raise ValueError("trf_output needs trfContentDic, trf_tables and pretrnaNameSeqDic")
    if a_to_i and genome is None:
        raise ValueError("a_to_i needs a genome (mirge_amd.a2i.EngineGenome)")
    isomirDic = write_mapped_csv(os.path.join(outputdir, "mapped.csv"), annotNameList, sampleList, seqDic,
                                 spikeIn)
    if gff_output:
        from .isomir import write_isomir_gff
        write_isomir_gff(outputdir, sampleList, isomiRContentDic, seqDic, miRNA_database)
    if trf_output:  # W2C:648
        from .trf import write_trf_tables
        write_trf_tables(outputdir, sampleList, logDic, trfContentDic, trf_tables, pretrnaNameSeqDic)
    if isomirDiff:
        write_isomir_tables(os.path.join(outputdir, "isomirs.csv"),
                            os.path.join(outputdir, "isomirs.samples.csv"), sampleList, isomirDic, logDic)
    write_unmapped_csv(os.path.join(outputdir, "unmapped.csv"), annotNameList, sampleList, seqDic, spikeIn)
    write_counts_csv(os.path.join(outputdir, "miR.Counts.csv"), sampleList, mirDic, logDic)
    write_rpm_csv(os.path.join(outputdir, "miR.RPM.csv"), sampleList, mirDic, logDic)
    if a_to_i:  # W2C:1221
        from .a2i import a_to_i_report
        a_to_i_report(outputdir, sampleList, logDic, seqDic, mirDic, mirNameSeqDic, mirMergedNameDic,
                      removedMiRNAList or [], genome)
    return isomirDic
