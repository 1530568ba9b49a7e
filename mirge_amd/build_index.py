"""`python -m mirge_amd.build_index lib.fa [more.fa ...] [-o prefix]`

The offline counterpart of `bowtie-build` for this engine: FASTA -> `<prefix>.mrgfm`
(FM index in the layout of mirge_amd/csrc/fm_index.hpp).  Host-only, no GPU needed.
`--max-bases N` splits a FASTA at entry boundaries into `<prefix>.partNNN.mrgfm` of at most
N bases each (the genome for -ai: one index addresses < 2^31 bases and mrg_count_best
wants <= 600 M per library so its rank table stays in LDS).
`<prefix>` defaults to the FASTA path without its extension, so an index sits next to
the bowtie-style prefix the reference passes around (MAIN:269-281).
"""
import argparse
import os
import sys
import time

from .index import FmIndex


def split_fasta(path, max_bases):
    """Yield (names, seqs) groups of whole entries with at most max_bases bases each."""
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    names, seqs, total = [], [], 0
    name, chunks, n = None, [], 0

    def close_entry():
        nonlocal names, seqs, total
        if name is None:
            return None
        if n > max_bases:
            raise ValueError("%s: entry %s has %d bases > --max-bases %d" % (path, name, n, max_bases))
        out = None
        if total + n > max_bases and names:
            out = (names, seqs)
            names, seqs, total = [], [], 0
        names.append(name)
        seqs.append("".join(chunks))
        total += n
        return out

    with opener(path, "rt") as fh:
        for line in fh:
            if line.startswith(">"):
                group = close_entry()
                if group:
                    yield group
                name, chunks, n = line[1:].strip(), [], 0
            else:
                t = line.strip()
                chunks.append(t)
                n += len(t)
    group = close_entry()
    if group:
        yield group
    if names:
        yield names, seqs


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m mirge_amd.build_index", description=__doc__.split("\n")[0])
    ap.add_argument("fasta", nargs="+")
    ap.add_argument("-o", "--output", help="index prefix (only with a single FASTA)")
    ap.add_argument("--max-bases", type=int, default=0,
                    help="split at entry boundaries into <prefix>.partNNN.mrgfm of at most this many bases")
    args = ap.parse_args(argv)
    if args.output and len(args.fasta) != 1:
        ap.error("-o needs exactly one FASTA")
    for fa in args.fasta:
        prefix = args.output or os.path.splitext(fa)[0]
        t0 = time.time()
        if args.max_bases:
            for k, (names, seqs) in enumerate(split_fasta(fa, args.max_bases)):
                ix = FmIndex.build(names, seqs)
                out = "%s.part%03d.mrgfm" % (prefix, k)
                ix.save(out)
                print("%s: part %d, %d entries, %d bp -> %s (%.1f s)" %
                      (fa, k, ix.info.n_ref, ix.info.n_bases, out, time.time() - t0))
            continue
        ix = FmIndex.from_fasta(fa)
        ix.save(prefix + ".mrgfm")
        inf = ix.info
        print("%s: %d entries, %d bp, %.1f MB index -> %s.mrgfm (%.1f s)" %
              (fa, inf.n_ref, inf.n_bases, (inf.bytes_fm + inf.bytes_sa) / 1e6, prefix, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
