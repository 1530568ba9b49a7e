"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the one miRge2.0 path this repository accelerates (the
bowtie cascade of utils/runAnnotationPipeline.py:566-707 and the tally of
utils/summarize.py:3-66, utils/miRNAmerge.py:3-42, utils/filter.py:3-31).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package, and only as the checker / timed baseline.  Nothing under
mirge_amd/ imports it; the product path fails loudly without its HIP library.

Pinning status
  * bookkeeping around the aligner (survivor selection, length filters, the
    poly-T pass, annot slots, per-pass counters, summarize, merge, filter):
    PINNED by tests/golden/*.json, captured in the build container from the
    reference's own Python (lib2to3-converted in a scratch dir) -- see
    tests/golden/make_golden.py.
  * the aligner arithmetic itself (which reads align to which library under
    -n/-v): PARITY UNPINNED.  It belongs to bowtie 1 (v1.1.1/1.1.2,
    /root/reference/README.md:49), which is neither vendored in the reference
    nor installed here, and the reference has no tests.  bowtie_model.c
    restates bowtie's published rules by exhaustive scan.
  * the index arrays the CPU port (fm_cpu.c) shares with the kernels: checked
    by definition against the library strings by index_check.c (suffix order,
    permutation, row fields, BWT blocks, jump tables) -- in tests and, at full
    size, in bench.py's parity gate.
"""
