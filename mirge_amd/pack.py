"""2-bit packing of reads into the structure-of-arrays layout of the C-ABI.

Columnar counterpart of the reference's `seqDic` keys (quantReads.py:9-16): a
read set is `words` uint64 [W, n] (base i of read r in bits [2(i%32), +1] of
words[i//32, r]; A=0 C=1 G=2 T=3), `lens` uint8 [n] and, only when some read
holds a non-ACGT character, `nmask` uint64 [W, n] with bit 2(i%32) set for an N.
"""
import numpy as np

_CODE = np.full(256, 4, dtype=np.uint8)
for _ch, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("a", 0), ("c", 1), ("g", 2), ("t", 3)):
    _CODE[ord(_ch)] = _v
_LETTERS = np.frombuffer(b"ACGT", dtype=np.uint8)


def words_for(max_len):
    """Words per read the kernels are instantiated for (1, 2, 4 or 8)."""
    if max_len <= 32:
        return 1
    if max_len <= 64:
        return 2
    if max_len <= 128:
        return 4
    if max_len <= 255:
        return 8
    raise ValueError("reads longer than 255 nt are not supported (got %d)" % max_len)


def pack_codes(codes, W=None):
    """codes: uint8 [n, L] with values 0..3 (4 = N), fixed length L."""
    codes = np.asarray(codes, dtype=np.uint8)
    n, L = codes.shape
    W = W or words_for(L)
    words = np.zeros((W, n), dtype=np.uint64)
    nmask = None
    isn = codes > 3
    if isn.any():
        nmask = np.zeros((W, n), dtype=np.uint64)
    for i in range(L):
        w, sh = i >> 5, np.uint64((i & 31) * 2)
        col = codes[:, i] & 3 if nmask is None else np.where(isn[:, i], 0, codes[:, i])
        words[w] |= col.astype(np.uint64) << sh
        if nmask is not None:
            nmask[w] |= isn[:, i].astype(np.uint64) << sh
    lens = np.full(n, L, dtype=np.uint8)
    return words, lens, nmask


def pack_reads(seqs, W=None):
    """seqs: sequence of str (any lengths <= 255).  Returns (words, lens, nmask|None)."""
    n = len(seqs)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=n)
    max_len = int(lens.max()) if n else 1
    if max_len > 255:
        raise ValueError("reads longer than 255 nt cannot be described (uint8 lengths)")
    W = W or words_for(max_len)
    if max_len > 32 * W:
        raise ValueError("a read of %d nt does not fit %d words" % (max_len, W))
    words = np.zeros((W, n), dtype=np.uint64)
    nmask = np.zeros((W, n), dtype=np.uint64)
    any_n = False
    if n:
        blob = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8)
        codes = _CODE[blob]
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        for L in np.unique(lens):
            L = int(L)
            if L == 0:
                continue
            rows = np.nonzero(lens == L)[0]
            grid = codes[off[rows][:, None] + np.arange(L)[None, :]]
            w, _, nm = pack_codes(grid, W)
            words[:, rows] = w
            if nm is not None:
                nmask[:, rows] = nm
                any_n = True
    return words, lens.astype(np.uint8), (nmask if any_n else None)


def pack_ragged(seqs):
    """seqs: sequence of str of ANY length -> the ragged form of mrg_cascade_run_long: (words uint64 [total],
    nmask uint64 [total] | None, word_off uint64 [n + 1], lens uint32 [n]); read r is words[word_off[r] :
    word_off[r + 1]], ceil(len / 32) words (mrg_pack_reads_ragged; pure numpy, no library needed)."""
    n = len(seqs)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=n)
    nw = (lens + 31) // 32
    word_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(nw, out=word_off[1:].view(np.int64))
    total = int(word_off[n])
    words = np.zeros(total, dtype=np.uint64)
    nmask = np.zeros(total, dtype=np.uint64)
    any_n = False
    if total:
        codes = _CODE[np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8)]
        base_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=base_off[1:])
        read_of = np.repeat(np.arange(n), lens)            # the read of every base
        i = np.arange(codes.size) - base_off[read_of]      # its index in the read
        slot = word_off[read_of].astype(np.int64) + (i >> 5)
        sh = ((i & 31) * 2).astype(np.uint64)
        isn = codes > 3
        np.bitwise_or.at(words, slot, np.where(isn, 0, codes).astype(np.uint64) << sh)
        if isn.any():
            any_n = True
            np.bitwise_or.at(nmask, slot[isn], np.uint64(1) << sh[isn])
    return words, (nmask if any_n else None), word_off, lens.astype(np.uint32)


def unpack_reads(words, lens, nmask=None):
    """Inverse of pack_reads (reports, tests, the dict-shaped host path): vectorised over reads."""
    words = np.asarray(words, dtype=np.uint64)
    W, n = words.shape
    lens = np.asarray(lens).astype(np.int64)
    if n == 0:
        return []
    width = int(lens.max()) if n else 0
    if width == 0:
        return [""] * n
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    chars = np.empty((n, width), dtype=np.uint8)
    for w in range((width + 31) // 32):
        nb = min(32, width - 32 * w)
        sh = (2 * np.arange(nb)).astype(np.uint64)
        grid = letters[((words[w][:, None] >> sh[None, :]) & np.uint64(3)).astype(np.uint8)]
        if nmask is not None:
            isn = ((np.asarray(nmask, dtype=np.uint64)[w][:, None] >> sh[None, :]) & np.uint64(1)).astype(bool)
            grid[isn] = ord("N")
        chars[:, 32 * w:32 * w + nb] = grid
    blob = chars.tobytes()
    return [blob[r * width:r * width + L].decode("ascii") for r, L in enumerate(lens.tolist())]


COMPACT_ESCAPE = 255      # one-byte count meaning "look in the escape list"
COMPACT_MAX_RUNS = 64


def pack_bits(w, L):
    """Reads of L bases (uint64 packed words) as a bit stream of 2 L bits each, little-endian bit order:
    uint64 [ceil(len(w) * 2 L / 64)]."""
    w = np.ascontiguousarray(w, dtype=np.uint64)
    m, bits = w.shape[0], 2 * int(L)
    if m == 0 or bits == 0:
        return np.zeros(0, dtype=np.uint64)
    if bits == 64:
        return w.copy()
    g = int(np.gcd(bits, 64))
    period, wpb = 64 // g, bits // g          # reads and words of one block of the stream
    n_blk = -(-m // period)
    v = np.zeros(n_blk * period, dtype=np.uint64)
    v[:m] = w & np.uint64((1 << bits) - 1)
    v = v.reshape(n_blk, period)
    out = np.zeros((n_blk, wpb), dtype=np.uint64)
    for j in range(period):
        off = bits * j
        k, sh = off >> 6, off & 63
        out[:, k] |= v[:, j] << np.uint64(sh)
        if sh + bits > 64:
            out[:, k + 1] |= v[:, j] >> np.uint64(64 - sh)
    return np.ascontiguousarray(out.reshape(-1)[:-(-(m * bits) // 64)])


def compact_read_set(words, lens, quant=None):
    """The compact wire form of a host-resident collapsed read set (include/mirge_amd.h:
    mrg_expand_compact): one-word N-free reads grouped by length, 2 L bits each.
    -> dict(order (the reads' permutation into length groups, None = unchanged), bits uint64 [n_words]
            (every run starts a word; one word of padding at the end), runs uint32 [n_runs, 2] (length, count),
            quant8 uint8 [n, S] or None, esc uint32 [k, 2] (flat index, count) or None)."""
    words = np.asarray(words)
    lens = np.asarray(lens, dtype=np.uint8)
    if words.ndim != 2 or words.shape[0] != 1:
        raise ValueError("compact_read_set: one word per read")
    n = lens.shape[0]
    if n and int(lens.max()) > 32:
        raise ValueError("compact_read_set: reads of more than 32 nt")
    order = None
    if n and np.any(lens[1:] < lens[:-1]):
        order = np.argsort(lens, kind="stable")
        lens = lens[order]
    w = words[0] if order is None else words[0][order]
    runs = compact_runs(lens)
    parts, a = [], 0
    for L, count in runs:
        parts.append(pack_bits(w[a:a + int(count)], int(L)))
        a += int(count)
    bits = np.concatenate(parts + [np.zeros(1, dtype=np.uint64)])
    quant8 = esc = None
    if quant is not None:
        q = np.asarray(quant)
        q = q.reshape(n, q.shape[1] if q.ndim == 2 else 1)
        if order is not None:
            q = q[order]
        quant8, esc = compact_counts(q)
    return dict(order=order, bits=bits, runs=runs, quant8=quant8, esc=esc)


def compact_runs(lens):
    """(length, count) runs of a length array grouped by length: uint32 [n_runs, 2]."""
    lens = np.asarray(lens)
    if lens.shape[0] == 0:
        return np.zeros((0, 2), dtype=np.uint32)
    cut = np.flatnonzero(lens[1:] != lens[:-1]) + 1
    starts = np.concatenate(([0], cut))
    ends = np.concatenate((cut, [lens.shape[0]]))
    runs = np.stack([lens[starts].astype(np.uint32), (ends - starts).astype(np.uint32)], axis=1)
    if runs.shape[0] > COMPACT_MAX_RUNS:
        raise ValueError("compact_runs: %d length runs (at most %d): group the reads by length" % (runs.shape[0], COMPACT_MAX_RUNS))
    return np.ascontiguousarray(runs)


def compact_counts(quant):
    """uint32 counts [n, S] -> (uint8 [n, S] with 255 = escaped, uint32 [k, 2] (flat index, count))."""
    q = np.ascontiguousarray(np.asarray(quant)).astype(np.uint32, copy=False)
    big = q >= COMPACT_ESCAPE
    q8 = np.where(big, COMPACT_ESCAPE, q).astype(np.uint8)
    idx = np.flatnonzero(big.reshape(-1))
    esc = np.stack([idx.astype(np.uint32), q.reshape(-1)[idx]], axis=1) if idx.size else np.zeros((0, 2), dtype=np.uint32)
    return q8, np.ascontiguousarray(esc)
