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
    """Words per read the kernels are instantiated for (1, 2 or 4)."""
    if max_len <= 32:
        return 1
    if max_len <= 64:
        return 2
    if max_len <= 128:
        return 4
    raise ValueError("reads longer than 128 nt are not supported (got %d)" % max_len)


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
    """seqs: sequence of str (any lengths <= 128).  Returns (words, lens, nmask|None)."""
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
