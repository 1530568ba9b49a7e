"""Run-time probe for a real bowtie 1 and, when one is found, the reference-shaped cascade
(TEST INFRASTRUCTURE: used by bench.py's cpu_baseline leg and by tests only).

The reference's aligner is the external `bowtie` binary (v1.1.1/1.1.2, /root/reference/README.md:49)
which is neither vendored nor installed in the build image, so parity at the aligner boundary is
unpinned (oracle/__init__.py).  This module is the one place that can pin it: if `bowtie` and
`bowtie-build` are on the box that runs the benchmark, the libraries are indexed with
bowtie-build and the nine command lines of runAnnotationPipeline.py:577-599 / :688 are run
verbatim (`--threads N <index> <flags> -f --norc -S reads.fa 1> sam 2> log`), the SAM / log files
are parsed as parseAlignment (:20-28) and parseBowtieLog (:9-18) do, and the survivors go to the
next pass as writeSeqToAnnot (:543-554) and the poly-T step (:664-686) select them.

  find_bowtie()               -> (bowtie, bowtie_build) or None
  reference_cascade(...)      -> dict(pass_id, ref_name, pos, stats, seconds, threads)
  compare(...)                -> D1 / D2 / D3 disagreement counts
"""
import os
import re
import shutil
import subprocess
import tempfile
import time

import numpy as np

# (library key, lengthFilter, mode flags) -- runAnnotationPipeline.py:574-599, :688
PASSES = [
    ("mirna", -26, "-n 0 -f --norc -S"),
    ("hairpin", 25, "-n 1 -f --norc -S"),
    ("mature_trna", 0, "-v 1 -f -a --best --strata --norc -S"),
    ("pre_trna", 0, "-v 0 -f  -a --best --strata --norc -S"),   # on poly-T-stripped reads
    ("snorna", 0, "-n 1 -f --norc -S"),
    ("rrna", 0, "-n 1 -f --norc -S"),
    ("ncrna_others", 0, "-n 1 -f --norc -S"),
    ("mrna", 0, "-n 0 -f --norc -S"),
    ("mirna", 0, "-5 1 -3 2 -v 2 -f --norc --best -S"),
]
POLY_T_PASS = 3


def find_bowtie(extra_dirs=()):
    """Paths of `bowtie` and `bowtie-build` (bowtie 1) if both are runnable here, else None."""
    path = os.pathsep.join(list(extra_dirs) + [os.environ.get("PATH", "")])
    b, bb = shutil.which("bowtie", path=path), shutil.which("bowtie-build", path=path)
    if not b or not bb:
        return None
    return b, bb


def _parse_log(path):
    """parseBowtieLog, runAnnotationPipeline.py:9-18."""
    processed = aligned = 0
    with open(path) as fh:
        for line in fh:
            if "# reads processed:" in line:
                processed = int(line.strip().split(" ")[-1])
            if "# reads with at least one reported alignment:" in line:
                aligned = int(line.strip().split(" ")[-2])
    return processed, aligned


def _parse_sam(path):
    """parseAlignment, runAnnotationPipeline.py:20-28: last line per QNAME wins."""
    out = {}
    with open(path) as fh:
        for line in fh:
            if "@" not in line:
                f = line.strip().split("\t")
                out[f[0]] = (f[2], f[1], f[3], f[5])
    return out


def check_ebwt_reader(prefix, names, seqs, ebwt_reader, bowtie_inspect=None):
    """The one moment the product's `.1.ebwt` reader (bowtie-inspect's job: summarize.py:6,
    runAnnotationPipeline.py:610-611,630) can be PINNED: a bowtie-build of this box has just written
    `<prefix>.*.ebwt` from (names, seqs).  ebwt_reader(prefix) -> (names, seqs) is the product's reader;
    what it returns must be what was indexed (names up to their first blank, sequences upper-case) and,
    when `bowtie-inspect` exists too, what that prints.  Returns a dict(ok, entries, detail)."""
    want_names = [n.split()[0] if n.split() else n for n in names]
    want_seqs = [s.upper() for s in seqs]
    try:
        got_names, got_seqs = ebwt_reader(prefix)
    except Exception as e:  # the reader refuses files whose redundant fields it cannot re-derive
        return dict(ok=False, entries=len(names), detail="reader raised: %s" % e)
    got_names, got_seqs = list(got_names), [s.upper() for s in got_seqs]
    if got_names != want_names:
        k = next((i for i, (a, b) in enumerate(zip(got_names, want_names)) if a != b), min(len(got_names), len(want_names)))
        return dict(ok=False, entries=len(names), detail="names differ from the indexed FASTA at entry %d (%d vs %d entries)"
                    % (k, len(got_names), len(want_names)))
    if got_seqs != want_seqs:
        k = next(i for i, (a, b) in enumerate(zip(got_seqs, want_seqs)) if a != b)
        return dict(ok=False, entries=len(names), detail="sequence of entry %d (%s) differs from the indexed FASTA" % (k, want_names[k]))
    detail = "names + sequences = the FASTA bowtie-build indexed"
    if bowtie_inspect:
        out = subprocess.run([bowtie_inspect, prefix], capture_output=True, text=True)
        if out.returncode == 0:
            i_names, i_seqs, cur = [], [], None
            for line in out.stdout.splitlines():
                if line.startswith(">"):
                    i_names.append(line[1:].split()[0] if line[1:].split() else line[1:])
                    i_seqs.append([])
                elif i_seqs:
                    i_seqs[-1].append(line.strip())
            i_seqs = ["".join(x).upper() for x in i_seqs]
            if i_names != got_names or i_seqs != got_seqs:
                return dict(ok=False, entries=len(names), detail="differs from what bowtie-inspect prints")
            detail += " = what bowtie-inspect prints"
    return dict(ok=True, entries=len(names), detail=detail)


def reference_cascade(bowtie, bowtie_build, libraries, reads, threads=1, workdir=None, keep=False, ebwt_reader=None):
    """libraries: {key: (names, seqs)} for the keys of PASSES; reads: list of DISTINCT ASCII reads.
    Returns dict(pass_id int8 [n] (-1 = unannotated), ref_name [n], pos int32 [n] (SAM POS - 1),
    stats [(processed, aligned)] per pass, seconds (the nine bowtie runs + SAM parsing, index
    construction excluded), build_seconds, ebwt {key: check_ebwt_reader(...)} when ebwt_reader is given:
    every index bowtie-build wrote is handed to the product's `.1.ebwt` reader)."""
    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="mrg_bowtie_")
    os.makedirs(workdir, exist_ok=True)
    try:
        t0 = time.perf_counter()
        prefix = {}
        for key in dict.fromkeys(k for k, _, _ in PASSES):
            names, seqs = libraries[key]
            fa = os.path.join(workdir, key + ".fa")
            with open(fa, "w") as fh:
                for n, s in zip(names, seqs):
                    fh.write(">%s\n%s\n" % (n, s))
            prefix[key] = os.path.join(workdir, key)
            subprocess.run([bowtie_build, "-q", fa, prefix[key]], check=True, stdout=subprocess.DEVNULL)
        build_s = time.perf_counter() - t0
        ebwt = None
        if ebwt_reader is not None:
            inspect = shutil.which("bowtie-inspect", path=os.pathsep.join([os.path.dirname(bowtie), os.environ.get("PATH", "")]))
            ebwt = {key: check_ebwt_reader(prefix[key], libraries[key][0], libraries[key][1], ebwt_reader, inspect)
                    for key in prefix}
        n = len(reads)
        index_of = {r: i for i, r in enumerate(reads)}
        pass_id = np.full(n, -1, dtype=np.int8)
        ref_name = [""] * n
        pos = np.full(n, -1, dtype=np.int32)
        stats = []
        fasta = os.path.join(workdir, "SeqToAnnot.fasta")
        sam = os.path.join(workdir, "SeqToAnnot.sam")
        log = os.path.join(workdir, "SeqToAnnot.log")
        t1 = time.perf_counter()
        for i, (key, length_filter, flags) in enumerate(PASSES):
            # writeSeqToAnnot, :543-554
            survivors = []
            for r, pid in zip(reads, pass_id):
                if pid >= 0:
                    continue
                if length_filter < 0 and not len(r) < -length_filter:
                    continue
                if length_filter > 0 and not len(r) > length_filter:
                    continue
                survivors.append(r)
            fan = None
            if i == POLY_T_PASS:  # :664-686
                fan, stripped = {}, []
                for r in survivors:
                    if re.search("T{3,}$", r) is None:
                        continue
                    sub = r.rstrip("T")
                    if len(sub) >= 11:
                        stripped.append(sub)
                        fan.setdefault(sub, []).append(r)
                survivors = stripped
            with open(fasta, "w") as fh:
                for r in survivors:
                    fh.write(">%s\n%s\n" % (r, r))
            cmd = "%s --threads %d %s %s %s 1> %s 2> %s" % (bowtie, threads, prefix[key], flags, fasta, sam, log)
            rc = os.system(cmd)
            if rc != 0:
                raise RuntimeError("bowtie exited with status %d: %s" % (rc, cmd))
            stats.append(_parse_log(log))
            for q, (rname, _flag, p1, _cigar) in _parse_sam(sam).items():
                if rname == "*":
                    continue
                for r in (fan[q] if fan is not None else [q]):   # updateAnnotDic / updateAnnotDic2, :341-352
                    k = index_of[r]
                    pass_id[k] = i
                    ref_name[k] = rname
                    pos[k] = int(p1) - 1
        return dict(pass_id=pass_id, ref_name=ref_name, pos=pos, stats=stats,
                    seconds=time.perf_counter() - t1, build_seconds=build_s, threads=threads, ebwt=ebwt)
    finally:
        if own and not keep:
            shutil.rmtree(workdir, ignore_errors=True)


def compare(ref, pass_id, ref_name, pos):
    """Disagreement counts by determinism class: D1 = which pass claims the read
    (tie-break independent: must be 0), D2 = a different miRNA entry in pass 0 / 8, D3 = a
    different entry / offset string elsewhere (bowtie's tie-break is RNG-driven)."""
    d1 = int((np.asarray(pass_id) != ref["pass_id"]).sum())
    d2 = d3 = 0
    for i in range(len(pass_id)):
        if pass_id[i] < 0 or pass_id[i] != ref["pass_id"][i]:
            continue
        if ref_name[i] != ref["ref_name"][i] or int(pos[i]) != int(ref["pos"][i]):
            if pass_id[i] in (0, 8) and ref_name[i] != ref["ref_name"][i]:
                d2 += 1
            else:
                d3 += 1
    return dict(D1=d1, D2=d2, D3=d3)
