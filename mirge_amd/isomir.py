"""isomiR classification for the `-gff` output.

Host-side restatement, in coordinates rather than dash-padded strings, of
  make_id / NT2CODE          runAnnotationPipeline.py:214-235, :620-627
  make_cigar                 runAnnotationPipeline.py:180-212
  fillTerminal + analyzeAlignment   runAnnotationPipeline.py:86-172, :237-339
  inferPremiRName            runAnnotationPipeline.py:354-380
  updateIsomiRDic / updateIsomiRDic2  runAnnotationPipeline.py:382-446
  extractPreMiRName          extractPreMiRName.py:19-63
  the per-sample GFF writer  writeDataToCSV.py:621-646

Inputs are what the GPU cascade returns for reads claimed by pass 0 (exact
miRNA) or pass 8 (isomiR): library entry and 0-based offset (SAM POS - 1).
Everything lives on one axis, the precursor's: precursor P at [0, len P), the
library entry E (2 nt flank + mature + 6 nt flank, RAP:413) placed so that its
mature M sits on M's first occurrence in P, and the read R at the offset the
aligner reported inside E (one base earlier for pass 8, whose `-5 1` trimmed it).
The reference quirks are kept: the iso_snp position class is taken from the index
in the PADDED alignment (RAP:257-270), i.e. relative to the leftmost of P, E, R.
"""
import os

# 3-mer -> UID character (mirGFF3 / mirtop read-UID alphabet), codons in ACGT order
_UID_CHARS = "@fcoladsmkhwgebpvtDnx#yiCEGSrjqHT84FVXZ6KM$AWY35LNJzU9P07IuBQOR%"
NT2CODE = {a + b + c: _UID_CHARS[16 * i + 4 * j + k]
           for i, a in enumerate("ACGT") for j, b in enumerate("ACGT") for k, c in enumerate("ACGT")}


def make_id(seq, nt2code=NT2CODE):
    """Read UID: one character per 3-mer; a trailing 1- or 2-mer is padded with A and
    followed by the pad length; any non-ACGT 3-mer turns the whole UID into '.'."""
    n_full = len(seq) // 3
    out = []
    for t in range(n_full):
        ch = nt2code.get(seq[3 * t:3 * t + 3])
        if ch is None:
            return "."
        out.append(ch)
    rest = len(seq) - 3 * n_full
    if rest:
        pad = 3 - rest
        ch = nt2code.get(seq[3 * n_full:] + "A" * pad)
        if ch is None:
            return "."
        out.append(ch + str(pad))
    return "".join(out)


def make_cigar(seq, ref):
    """Column-wise: match 'M' (run-length encoded, a single M stays 'M'), substitution =
    the read's base, '-' in the read 'D', '-' in the reference 'I'."""
    ops = []
    for a, b in zip(seq, ref):
        if a == b:
            ops.append("M")
        elif a == "-":
            ops.append("D")
        elif b == "-":
            ops.append("I")
        else:
            ops.append(a)
    out = []
    run = 0
    for op in ops:
        if op == "M":
            run += 1
            continue
        if run:
            out.append("M" if run == 1 else "%dM" % run)
            run = 0
        out.append(op)
    if run:
        out.append("M" if run == 1 else "%dM" % run)
    return "".join(out)


def _snp_class(i):
    if 1 <= i <= 6:
        return "_seed"
    if i == 7:
        return "_central_offset"
    if 8 <= i <= 11:
        return "_central"
    if 12 <= i <= 16:
        return "central_supp"
    return ""


def classify_alignment(pre_seq, lib_seq, read, start, index_value):
    """fillTerminal + analyzeAlignment.  `start` = SAM POS (1-based) of the aligned
    read inside `lib_seq`; index_value 0 = exact-miRNA pass, anything else = isomiR pass.
    Returns (type, variant, pre_start, pre_end, cigar), or None when the mature
    sequence does not occur in the precursor (the reference then drops the read)."""
    mature = lib_seq[2:-6]
    m0 = pre_seq.find(mature)
    if m0 < 0:
        return None
    m1 = m0 + len(mature)
    e0 = m0 - 2
    r0 = e0 + (start - 1 if index_value == 0 else start - 2)
    r1 = r0 + len(read)
    # the padded frame of the reference starts at the leftmost of P, E (and R for isomiRs)
    frame0 = min(0, e0) if index_value == 0 else min(0, e0, r0)

    def pre_at(x):
        return pre_seq[x] if 0 <= x < len(pre_seq) else "-"

    def read_at(x):
        return read[x - r0] if r0 <= x < r1 else "-"

    def span(fn, a, b):
        return "".join(fn(x) for x in range(a, b))

    if r0 == m0 and r1 == m1 and read == mature:
        kind, variants = "ref_miRNA", ["NA"]
    else:
        kind, variants = "isomiR", []
        snp, snp_pos = False, ""
        for x in range(max(m0, r0), min(m1, r1)):
            if read[x - r0] != mature[x - m0]:
                snp, snp_pos = True, _snp_class(x - frame0)
                break
        add_pos = p5 = p3 = None
        if r0 > m0:
            p5 = str(m0 - r0)
        elif r0 < m0:
            p5 = "+" + str(m0 - r0)
            if span(read_at, r0, m0) != span(pre_at, r0, m0):
                snp, snp_pos = True, ""
        if r1 < m1:
            p3 = str(r1 - m1)
        elif r1 > m1:
            if span(read_at, m1, r1) != span(pre_at, m1, r1):
                add_pos = "+" + str(r1 - m1)
            else:
                p3 = "+" + str(r1 - m1)
        if snp:
            variants.append("iso_snp" + snp_pos)
        if add_pos is not None:
            variants.append("iso_add:" + add_pos)
        if p5 is not None:
            variants.append("iso_5p:" + p5)
        if p3 is not None:
            variants.append("iso_3p:" + p3)
    cigar = make_cigar(read, span(pre_at, r0, r1))
    return kind, ",".join(variants), r0 + 1, r1, cigar


def infer_premir_name(canonical, mirna_to_pre, database):
    """inferPremiRName: the gff3-derived map first, then the naming conventions."""
    if canonical in mirna_to_pre:
        return mirna_to_pre[canonical]
    if database == "miRBase":
        stem = "-".join(canonical.split("-")[:-1])
        for cand in (stem, stem + "-5p", stem + "-3p"):
            if cand in mirna_to_pre:
                return mirna_to_pre[cand]
        return canonical.replace("-5p", "").replace("-3p", "").replace("miR", "mir")
    name = canonical
    for suffix in ("_5p*", "_3p*", "_5p", "_3p"):
        name = name.replace(suffix, "")
    return name + "_pre"


def build_isomir_content(content, hits, index_value, mirna_to_pre, hairpin_seqs, mirna_lib_seqs,
                         database):
    """updateIsomiRDic + updateIsomiRDic2 for one pass.

    content      : isomiRContentDic, updated in place (read -> field dict)
    hits         : {read: (miRNA entry name, SAM POS as int, CIGAR of the aligner)}
    index_value  : 0 (exact pass) or 8 (isomiR pass)
    hairpin_seqs / mirna_lib_seqs : name -> sequence (`bowtie-inspect`, RAP:609-619)"""
    for read, (name, pos, aligner_cigar) in hits.items():
        canonical = name.split(".")[0] if "." in name else name
        rec = content.setdefault(read, {})
        rec["miRName"] = name
        rec["preMiRName"] = infer_premir_name(canonical, mirna_to_pre, database)
        rec["start"] = str(pos)
        rec["cigar"] = aligner_cigar if index_value == 0 else str(len(read)) + "M"
        rec["annot"] = 0
        rec["filter"] = "Pass"
        rec["uid"] = make_id(read)
    for read in list(content.keys()):
        rec = content[read]
        if rec["annot"] != 0:
            continue
        lib_seq = mirna_lib_seqs[rec["miRName"]]
        mature = lib_seq[2:-6]
        pre_seq = hairpin_seqs[rec["preMiRName"]]
        if ".SNP" in rec["miRName"] and ".SNPC" not in rec["miRName"]:
            # a SNP entry is compared with the precursor carrying the same SNP (RAP:417-432)
            base = rec["miRName"].split(".")[0]
            canon_mature = mirna_lib_seqs[base + ".SNPC"][2:-6]
            canon_pre = hairpin_seqs[infer_premir_name(base, mirna_to_pre, database)]
            at = canon_pre.find(canon_mature)
            if at < 0:
                raise ValueError("canonical mature of %s not found in its precursor" % rec["miRName"])
            pre_seq = canon_pre[:at] + mature + canon_pre[at + len(mature):]
        res = classify_alignment(pre_seq, lib_seq, read, int(rec["start"]), index_value)
        if res is None:
            del content[read]
            continue
        kind, variant, pre_start, pre_end, cigar = res
        rec.update(type=kind, pre_start=str(pre_start), pre_end=str(pre_end), variant=str(variant),
                   strand="+", annot=1, cigar=cigar)


def write_isomir_gff(outputdir, sampleList, content, seqDic, database):
    """<sample>_isomiRs.gff, one per sample (writeDataToCSV.py:621-646)."""
    source = "miRBase22" if database == "miRBase" else database + "2.0"
    for i, sample in enumerate(sampleList):
        name = os.path.splitext(sample)[0]
        with open(os.path.join(outputdir, name + "_isomiRs.gff"), "w") as out:
            out.write("# GFF3 adapted for miRNA sequencing data\n## VERSION 0.0.1\n## source-ontology: ")
            out.write(source + "\n")
            out.write("## COLDATA: %s\n" % name)
            for read, rec in content.items():
                count = seqDic[read]["quant"][i]
                if count >= 1:
                    rec["expression"] = str(count)
                    out.write("\t".join([rec["miRName"], source, rec["type"], rec["pre_start"],
                                         rec["pre_end"], ".", rec["strand"], "."]))
                    out.write("\t")
                    out.write(";".join(["Read " + read, " UID " + rec["uid"], " Name " + rec["miRName"],
                                        " Parent " + rec["preMiRName"], " Variant " + rec["variant"],
                                        " Cigar " + rec["cigar"], " Expression " + rec["expression"],
                                        " Filter " + rec["filter"]]))
                    out.write("\n")


def _pick_optimal(names):
    """pickOptimal, extractPreMiRName.py:9-17: lowest numeric suffix, else lexicographic."""
    tails = [n.split("-")[-1] for n in names]
    if all(t.isdigit() for t in tails):
        return sorted([int(t), n] for t, n in zip(tails, names))[0][-1]
    return sorted(names)[0]


def extract_premir_name(gff3_path, database):
    """extractPreMiRName.py:19-63: miRNA name -> precursor (stem-loop) name."""
    out = {}
    if database == "miRBase":
        order, derives, stem_name = [], {}, {}
        with open(gff3_path) as fh:
            for line in fh:
                if line[0] == "#":
                    continue
                f = line.strip().split("\t")
                if f[2] == "miRNA":
                    name = f[-1].split("Name=")[1].split(";")[0].strip()
                    parent = f[-1].split("Derives_from=")[1].split(";")[0].strip()
                    if name not in derives:
                        order.append(name)
                        derives[name] = []
                    if parent not in derives[name]:
                        derives[name].append(parent)
                if f[2] == "miRNA_primary_transcript":
                    alias = f[-1].split("Alias=")[1].split(";")[0].strip()
                    stem_name[alias] = f[-1].split("Name=")[1].split(";")[0].strip()
        for name in order:
            parents = derives[name]
            out[name] = _pick_optimal([stem_name[p] for p in parents]) if len(parents) > 1 \
                else stem_name[parents[0]]
    else:
        with open(gff3_path) as fh:
            for line in fh:
                if line[0] == "#":
                    continue
                f = line.strip().split("\t")
                if f[2] == "miRNA":
                    name = f[-1].split("ID=")[1].split(";")[0].strip()
                    out[name] = "_".join(name.split("_")[:-1]) + "_pre"
    return out
