"""TEST INFRASTRUCTURE: Python restatement of the ingest path.

  quality_trim_3p / load_fastq : trim_file.py:24-66,89-134 with `-ad none` -- the only
      modifier is cutadapt's QualityTrimmer(0, 10, base).  cutadapt (<= 1.16) is a third-party
      dependency absent from the image, so this restates its published 3' algorithm
      (the BWA rule; parity unpinned against cutadapt itself).
  locate_adapter_3p / trim_read : the AdapterCutter / UnconditionalCutter step of
      trim_file.py:34-41 (`-ad <adapter>` / `-ad +N`).  cutadapt's 3' adapter search is restated
      as a full dynamic-programming table (the product keeps one column): PARITY UNPINNED
      against cutadapt itself, which is absent.
  collapse : quantReads.py:3-24 is in oracle/cascade.py (pinned by tests/golden).
"""
import gzip


def locate_adapter_3p(adapter, read, max_error_rate=0.12, min_overlap=3):
    """cutadapt `-a ADAPTER` on one read.  None, or (read_start, read_stop, adapter_stop,
    matches, errors); the read is cut at read_start.

    Rules restated: an exact occurrence wins (leftmost).  Otherwise unit-cost edit alignment of
    an adapter PREFIX (whole adapter inside the read, or a shorter prefix ending exactly at the
    read's 3' end) against a read substring that may start anywhere.  Cell (i, j) = best way to
    align adapter[:i] ending at read[:j], carrying (cost, matches, origin = read start); on a
    mismatching cell the diagonal is preferred, then the cell above (adapter base inserted),
    then the cell to the left.  A row whose every reachable cost exceeds k = int(rate * len)
    stops being extended (cutadapt's band).  Candidates: (len(adapter), j) for every j in
    increasing order, then (i, len(read)) for every i; a candidate needs overlap i >= min_overlap
    and cost <= rate * i, and replaces the best so far only with more matches, or as many
    matches and a lower cost."""
    adapter, read = adapter.upper(), read.upper()
    m, n = len(adapter), len(read)
    if m == 0:
        return None
    p = read.find(adapter)
    if p >= 0:
        return (p, p + m, m, m, 0)
    min_overlap = min(min_overlap, m)
    k = int(max_error_rate * m)
    INF = None
    # table[j][i]
    table = [[INF] * (m + 1) for _ in range(n + 1)]
    for i in range(m + 1):
        table[0][i] = (i, 0, 0)
    last = min(m, k + 1)
    best = None   # (matches, cost, origin, adapter_stop, read_stop)

    def consider(i, j, cell):
        nonlocal best
        cost, matches, origin = cell
        if i >= min_overlap and cost <= i * max_error_rate and \
                (best is None or matches > best[0] or (matches == best[0] and cost < best[1])):
            best = (matches, cost, origin, i, j)
            return True
        return False

    done = False
    for j in range(1, n + 1):
        prev, cur = table[j - 1], table[j]
        cur[0] = (0, 0, j)
        for i in range(1, m + 1):
            if i > last:
                cur[i] = prev[i]          # outside the band: the stale value stays
                continue
            if adapter[i - 1] == read[j - 1]:
                d = prev[i - 1]
                cur[i] = (d[0], d[1] + 1, d[2])
            else:
                d, left, up = prev[i - 1], prev[i], cur[i - 1]
                cd, cl, cu = d[0] + 1, left[0] + 1, up[0] + 1
                if cd <= cl and cd <= cu:
                    cur[i] = (cd, d[1], d[2])
                elif cu <= cl:
                    cur[i] = (cu, up[1], up[2])
                else:
                    cur[i] = (cl, left[1], left[2])
        while last >= 0 and cur[last][0] > k:
            last -= 1
        if last < m:
            last += 1
        elif consider(m, j, cur[m]) and best[1] == 0 and best[0] == m:
            done = True
            break
    if not done:
        for i in range(m + 1):
            consider(i, n, table[n][i])
    if best is None:
        return None
    matches, cost, origin, a_stop, r_stop = best
    return (max(origin, 0), r_stop, a_stop, matches, cost)


def trim_read(seq, adapter):
    """The modifier after QualityTrimmer (trim_file.py:34-41); `adapter` as MAIN:123-127 leaves it."""
    if adapter in (None, "", "none"):
        return seq
    if adapter.startswith("+"):
        k = int(adapter)
        return seq[k:] if k > 0 else (seq[:k] if k < 0 else seq)
    best = None
    for a in adapter.split(","):
        if not a:
            continue
        m = locate_adapter_3p(a, seq)
        if m is not None and (best is None or m[3] > best[3]):
            best = m
    return seq if best is None else seq[:best[0]]


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


def load_fastq(path, cutoff=10, min_len=16, adapter="none"):
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
            trimmed = trim_read(seq[:stop], adapter)
            if len(trimmed) >= min_len:
                kept.append(trimmed)
    return kept, total, 64 if any64 else 33
