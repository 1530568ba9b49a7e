"""TEST INFRASTRUCTURE: Python restatement of the ingest path.

  quality_trim_3p / load_fastq : trim_file.py:24-66,89-134 with `-ad none` -- the only
      modifier is cutadapt's QualityTrimmer(0, 10, base).  cutadapt (<= 1.16) is a third-party
      dependency absent from the image, so this restates its published 3' algorithm
      (the BWA rule; parity unpinned against cutadapt itself).
  collapse : quantReads.py:3-24 is in oracle/cascade.py (pinned by tests/golden).
"""
import gzip


def quality_trim_3p(qualities, cutoff=10, base=33):
    """Index at which the 3' end is cut: walk back from the end summing (cutoff - q); stop
    when the sum goes negative; cut where it peaked."""
    s, best, stop = 0, 0, len(qualities)
    for i in range(len(qualities) - 1, -1, -1):
        s += cutoff - (ord(qualities[i]) - base)
        if s < 0:
            break
        if s > best:
            best, stop = s, i
    return stop


def load_fastq(path, cutoff=10, min_len=16):
    """Returns (kept sequences, total records, phred as the reference reports it)."""
    opener = gzip.open if path.endswith(".gz") else open
    kept, total, worker64, any64 = [], 0, False, False
    with opener(path, "rt") as fh:
        while True:
            name = fh.readline()
            if not name:
                break
            seq = fh.readline().rstrip("\r\n")
            fh.readline()
            qual = fh.readline().rstrip("\r\n")
            if total < 1000:
                hi = any(ord(c) > 74 for c in qual)
                any64 = any64 or hi
                if total == 0:
                    worker64 = hi  # the workers are created while record 0 is read (TRM:107-110)
            total += 1
            stop = quality_trim_3p(qual, cutoff, 64 if worker64 else 33)
            if stop >= min_len:
                kept.append(seq[:stop])
    return kept, total, 64 if any64 else 33
