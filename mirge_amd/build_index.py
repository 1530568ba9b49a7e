"""`python -m mirge_amd.build_index lib.fa [more.fa ...] [-o prefix]`

The offline counterpart of `bowtie-build` for this engine: FASTA -> `<prefix>.mrgfm`
(FM index in the layout of mirge_amd/csrc/fm_index.hpp).  Host-only, no GPU needed.
`<prefix>` defaults to the FASTA path without its extension, so an index sits next to
the bowtie-style prefix the reference passes around (MAIN:269-281).
"""
import argparse
import os
import sys
import time

from .index import FmIndex


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m mirge_amd.build_index", description=__doc__.split("\n")[0])
    ap.add_argument("fasta", nargs="+")
    ap.add_argument("-o", "--output", help="index prefix (only with a single FASTA)")
    args = ap.parse_args(argv)
    if args.output and len(args.fasta) != 1:
        ap.error("-o needs exactly one FASTA")
    for fa in args.fasta:
        prefix = args.output or os.path.splitext(fa)[0]
        t0 = time.time()
        ix = FmIndex.from_fasta(fa)
        ix.save(prefix + ".mrgfm")
        inf = ix.info
        print("%s: %d entries, %d bp, %.1f MB index -> %s.mrgfm (%.1f s)" %
              (fa, inf.n_ref, inf.n_bases, (inf.bytes_fm + inf.bytes_sa) / 1e6, prefix, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
