"""Seeded synthetic miRge libraries and read sets (there is no network, and the
real `miRge.Libs` are not in the image).

Shapes follow SURVEY.md section 8(d): seven libraries sized like the human
miRBase-era set (miRNA entries are 2 nt 5' flank + mature + 6 nt 3' flank as
runAnnotationPipeline.py:413 assumes, cut out of their hairpins), plus the
companion merges CSV (miRNAmerge.py:15-36) and 2-line miRNA FASTA
(miRNAmerge.py:4-11).  Reads are fixed-length 22-mers in the stated mixture.
This module builds inputs only; it never computes an expected answer.
"""
import os

import numpy as np

_LETTERS = np.frombuffer(b"ACGT", dtype=np.uint8)

LIB_KEYS = ["mirna", "hairpin", "mature_trna", "pre_trna", "snorna", "rrna", "ncrna_others", "mrna"]

# entries, (min_len, max_len) -- SURVEY.md 8(d) defaults
FULL_SHAPES = {
    "mirna": 2800,
    "hairpin": (1900, 60, 120),
    "mature_trna": (450, 72, 76),
    "pre_trna": (600, 25, 60),
    "snorna": (1000, 70, 300),
    "rrna": [121, 157, 954, 1559, 1869, 5070, 1800, 3400],
    "ncrna_others": (20000, 100, 1000),
    "mrna": (50000, 500, 5000),
}


# "mouse" = the same generator with another seed and other sizes; BASELINE configs[4]
MOUSE_SEED = 10090
MOUSE_SHAPES = {
    "mirna": 2200,
    "hairpin": (1500, 60, 110),
    "mature_trna": (430, 72, 76),
    "pre_trna": (480, 25, 60),
    "snorna": (1100, 70, 260),
    "rrna": [121, 157, 954, 1559, 1870, 4730, 1580, 3300],
    "ncrna_others": (12000, 100, 1000),
    "mrna": (40000, 500, 4500),
}


def codes_to_str(codes):
    return _LETTERS[codes].tobytes().decode("ascii")


# what `repeats=True` does to a library: fractions of its entries (an entry may get several treatments)
REPEAT_RECIPES = {
    "mrna": dict(polya=0.05, polyt=0.01, tail=(20, 60), tandem=0.05, paralog=0.08, element=0.40, core_copies=2.0),
    "ncrna_others": dict(polya=0.02, polyt=0.0, tail=(15, 40), tandem=0.05, paralog=0.10, element=0.10, core_copies=0.0),
    "snorna": dict(polya=0.0, polyt=0.0, tail=(0, 0), tandem=0.02, paralog=0.20, element=0.0, core_copies=0.0),
}
_ELEMENT = np.random.default_rng(77).integers(0, 4, 280, dtype=np.uint8)   # the interspersed element (SINE-like), 280 nt
_CORE = (100, 160)                                                           # ... its conserved core: these 60 nt, copied exactly


def decorate_repeats(blob, off, rng, polya, polyt, tail, tandem, paralog, element, core_copies):
    """In place on the concatenated codes of a library (entry i = blob[off[i] : off[i + 1]]):
      paralog   the entry becomes a copy of another (as far as both reach) with 2 % substitutions
      element   280-nt interspersed element written at a random position, 5 % divergence outside its core
      core      the element's 60-nt core written `core_copies` times per entry on average, exactly
                (137 Mbp of mRNA at scale 1: ~10^5 copies -- every seed cut from it has 10^5 rows)
      tandem    a stretch of 50..400 nt filled with a motif of period 2..6
      polya/t   the last 20..60 nt (`tail`) become A / the first become T
    Returns the counts of each."""
    n = off.shape[0] - 1
    lens = np.diff(off)
    st = dict(paralog=0, element=0, core=0, tandem=0, polya=0, polyt=0)
    for i in np.nonzero(rng.random(n) < paralog)[0]:
        j = int(rng.integers(0, n))
        L = int(min(lens[i], lens[j]))
        if j == i or L < 50:
            continue
        seg = blob[off[j]:off[j] + L].copy()
        hit = rng.random(L) < 0.02
        seg[hit] = (seg[hit] + rng.integers(1, 4, int(hit.sum()))) & 3
        blob[off[i]:off[i] + L] = seg
        st["paralog"] += 1
    E = _ELEMENT.shape[0]
    for i in np.nonzero((rng.random(n) < element) & (lens > E + 20))[0]:
        p = int(off[i] + rng.integers(0, lens[i] - E))
        seg = _ELEMENT.copy()
        hit = rng.random(E) < 0.05
        hit[_CORE[0]:_CORE[1]] = False
        seg[hit] = (seg[hit] + rng.integers(1, 4, int(hit.sum()))) & 3
        blob[p:p + E] = seg
        st["element"] += 1
    C = _CORE[1] - _CORE[0]
    if core_copies > 0:
        k = rng.poisson(core_copies, n)
        for i in np.nonzero((k > 0) & (lens > 4 * C))[0]:
            for _ in range(int(k[i])):
                p = int(off[i] + rng.integers(0, lens[i] - C))
                blob[p:p + C] = _ELEMENT[_CORE[0]:_CORE[1]]
                st["core"] += 1
    for i in np.nonzero((rng.random(n) < tandem) & (lens > 120))[0]:
        period = int(rng.integers(2, 7))
        motif = rng.integers(0, 4, period, dtype=np.uint8)
        span = int(min(lens[i] - 20, rng.integers(50, 401)))
        p = int(off[i] + rng.integers(0, lens[i] - span))
        blob[p:p + span] = np.resize(motif, span)
        st["tandem"] += 1
    if tail[1] > 0:
        for frac, code, at_end, name in ((polya, 0, True, "polya"), (polyt, 3, False, "polyt")):
            for i in np.nonzero((rng.random(n) < frac) & (lens > 2 * tail[1]))[0]:
                t = int(rng.integers(tail[0], tail[1] + 1))
                if at_end:
                    blob[off[i + 1] - t:off[i + 1]] = code
                else:
                    blob[off[i]:off[i] + t] = code
                st[name] += 1
    return st


class SynthLibraries:
    """libs[key] = (names, seqs); codes[key] = (concatenated uint8 codes, starts)."""

    def __init__(self, seed=20181, scale=1.0, n_paralogs=60, n_snp=120, shapes=None, snpc=False, repeats=False):
        """repeats: the UNFRIENDLY variant of the large / unstructured libraries (the default is i.i.d.-uniform
        ACGT, what SURVEY.md 8d asks the headline for): what real rRNA / mRNA / ncRNA libraries hold and hash
        tables, seed buckets and suffix intervals dislike -- poly-A and poly-T tails on transcripts, tandem
        repeats, paralog families (copies of other entries with a few substitutions), and one interspersed
        element whose conserved core occurs ~10^5 times at full scale (`decorate_repeats`)."""
        rng = np.random.default_rng(seed)
        self.libs, self.codes = {}, {}
        FULL_SHAPES = dict(globals()["FULL_SHAPES"])
        if shapes:
            FULL_SHAPES.update(shapes)

        def sc(n, lo=4):
            return max(lo, int(round(n * scale)))

        # ---- hairpins with two embedded arms; miRNA entries are cut from them ----
        n_hp, hp_lo, hp_hi = FULL_SHAPES["hairpin"]
        n_hp = sc(n_hp)
        n_mir = sc(FULL_SHAPES["mirna"])
        hp_len = rng.integers(hp_lo, hp_hi + 1, n_hp)
        hp_codes = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in hp_len]
        mat_len_choices = np.array([18, 19, 20, 21, 22, 22, 22, 22, 23, 23, 24, 25])
        mir_names, mir_seqs, self.mirna_mature = [], [], []
        for j in range(n_mir):
            h = j % n_hp
            arm3 = (j // n_hp) % 2 == 1
            L = int(hp_len[h])
            mlen = int(mat_len_choices[rng.integers(0, len(mat_len_choices))])
            lo = 2 if not arm3 else L // 2 + 2
            hi = (L // 2 - mlen - 6) if not arm3 else (L - mlen - 6)
            start = int(rng.integers(lo, max(lo, hi) + 1)) if hi >= lo else lo
            start = min(start, L - mlen - 6)
            entry = hp_codes[h][start - 2:start + mlen + 6]
            mir_names.append("syn-miR-%d-%s" % (h + 1, "3p" if arm3 else "5p") +
                             ("" if j < 2 * n_hp else ".%d" % (j // (2 * n_hp))))
            mir_seqs.append(codes_to_str(entry))
        # paralogs: same mature, different flanks -> exact reads tie between entries;
        # the merges CSV folds each pair (miRNAmerge.py:15-36)
        self.merges = []
        n_par = min(sc(n_paralogs, 1), n_mir // 4)
        for k in range(n_par):
            a = int(rng.integers(0, n_mir))
            s = mir_seqs[a]
            flank5 = codes_to_str(rng.integers(0, 4, 2, dtype=np.uint8))
            flank3 = codes_to_str(rng.integers(0, 4, 6, dtype=np.uint8))
            name_b = mir_names[a] + "-par%d" % k
            mir_names.append(name_b)
            mir_seqs.append(flank5 + s[2:-6] + flank3)
            self.merges.append("%s/%s,%s,%s" % (mir_names[a], name_b, mir_names[a], name_b))
        # SNP entries: one substitution inside the mature (names carry .SNP, RAP:417-418)
        n_s = min(sc(n_snp, 1), n_mir // 4)
        for k in range(n_s):
            a = int(rng.integers(0, n_mir))
            s = list(mir_seqs[a])
            p = int(rng.integers(4, len(s) - 8))
            s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + int(rng.integers(0, 3))) % 4]
            mir_names.append(mir_names[a].split(".")[0] + ".SNP%d" % k)
            mir_seqs.append("".join(s))
            if snpc and mir_names[a].split(".")[0] + ".SNPC" not in mir_names:
                # the library convention the -gff code relies on (RAP:417-420): the canonical
                # sequence of a miRNA that has SNP entries is also listed as <name>.SNPC
                mir_names.append(mir_names[a].split(".")[0] + ".SNPC")
                mir_seqs.append(mir_seqs[a])
        self.libs["mirna"] = (mir_names, mir_seqs)
        self.libs["hairpin"] = (["syn-mir-%d" % (h + 1) for h in range(n_hp)],
                                [codes_to_str(c) for c in hp_codes])

        rep_rng = np.random.default_rng(seed + 977)   # (its own stream: the i.i.d. libraries stay what they were)
        self.repeat_stats = {}

        def uniform_lib(key, prefix, n, lo, hi, suffix=""):
            lens = rng.integers(lo, hi + 1, n)
            blob = rng.integers(0, 4, int(lens.sum()), dtype=np.uint8)
            if repeats and key in REPEAT_RECIPES:
                self.repeat_stats[key] = decorate_repeats(blob, np.concatenate([[0], np.cumsum(lens)]), rep_rng, **REPEAT_RECIPES[key])
            text = _LETTERS[blob].tobytes().decode("ascii")
            off = np.concatenate([[0], np.cumsum(lens)])
            self.libs[key] = (["%s-%d" % (prefix, i + 1) for i in range(n)],
                              [text[off[i]:off[i + 1]] + suffix for i in range(n)])

        n, lo, hi = FULL_SHAPES["mature_trna"]
        uniform_lib("mature_trna", "syn-tRNA", sc(n), lo, hi, suffix="CCA")
        n, lo, hi = FULL_SHAPES["pre_trna"]
        uniform_lib("pre_trna", "syn-pretRNA", sc(n), lo, hi)
        n, lo, hi = FULL_SHAPES["snorna"]
        uniform_lib("snorna", "syn-snoRNA", sc(n), lo, hi)
        rr = [max(60, int(L * min(1.0, max(scale, 0.05)))) for L in FULL_SHAPES["rrna"]]
        self.libs["rrna"] = (["syn-rRNA-%d" % (i + 1) for i in range(len(rr))],
                             [codes_to_str(rng.integers(0, 4, L, dtype=np.uint8)) for L in rr])
        n, lo, hi = FULL_SHAPES["ncrna_others"]
        uniform_lib("ncrna_others", "syn-ncRNA", sc(n), lo, hi)
        n, lo, hi = FULL_SHAPES["mrna"]
        uniform_lib("mrna", "syn-mRNA", sc(n), lo, hi)

        lut = np.zeros(256, dtype=np.uint8)
        for i, ch in enumerate(b"ACGT"):
            lut[ch] = i
        for key, (names, seqs) in self.libs.items():
            lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
            starts = np.concatenate([[0], np.cumsum(lens)])
            blob = lut[np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8)]
            self.codes[key] = (blob, starts)

    def total_bases(self, key):
        return int(self.codes[key][1][-1])

    def write_layout(self, root, species="syn", db="miRBase"):
        """Write the miRge.Libs directory layout with FASTA
        files where the reference keeps `.ebwt` indexes."""
        ix = os.path.join(root, species, "index.Libs")
        fa = os.path.join(root, species, "fasta.Libs")
        an = os.path.join(root, species, "annotation.Libs")
        for d in (ix, fa, an):
            os.makedirs(d, exist_ok=True)
        fname = {"mirna": "mirna_" + db, "hairpin": "hairpin_" + db}
        for key, (names, seqs) in self.libs.items():
            with open(os.path.join(ix, "%s_%s.fa" % (species, fname.get(key, key))), "w") as fh:
                for n, s in zip(names, seqs):
                    fh.write(">%s\n%s\n" % (n, s))
        names, seqs = self.libs["mirna"]
        with open(os.path.join(fa, "%s_mirna_SNP_pseudo_%s.fa" % (species, db)), "w") as fh:
            mature = {}
            for n, s in zip(names, seqs):
                mature[n] = s[2:-6]
                fh.write(">%s\n%s\n" % (n, s[2:-6]))
            # merged names are listed too (the -ai path looks them up, W2C:1295)
            for line in self.merges:
                f = line.split(",")
                fh.write(">%s\n%s\n" % (f[0], mature[f[1]]))
        with open(os.path.join(an, "%s_merges_%s.csv" % (species, db)), "w") as fh:
            for line in self.merges:
                fh.write(line + "\n")
        return os.path.join(ix, species + "_")


# read mixture of SURVEY.md 8(d): fraction per source
DEFAULT_MIX = dict(mirna_exact=0.55, isomir=0.15, trna=0.05, snorna=0.05, rrna_ncrna=0.05,
                   mrna=0.05, polyt=0.03, random=0.07)
EXACT_ONLY_MIX = dict(mirna_exact=0.02, random=0.98)  # config 2: "collapsed unique" is mostly misses
# config 5 (-ai): miRNA reads of which a part carries one A -> G substitution at a seeded position
A2I_MIX = dict(mirna_exact=0.40, mirna_a2g=0.20, isomir=0.12, trna=0.05, snorna=0.04, rrna_ncrna=0.04,
               mrna=0.05, polyt=0.03, random=0.07)


def _cut_packed(codes, base, L, cache=None):
    """2-bit pack codes[base : base+L] for every element of `base` (uint64 per read).
    `cache` (a dict owned by the library object) keeps the per-position windows of a
    small library between calls."""
    if codes.shape[0] <= (1 << 21) and cache is not None:
        # small library: pack a 32-base window at every text position once, then gather
        win = cache.get("win")
        if win is None:
            pad = np.concatenate([codes, np.zeros(32, dtype=np.uint8)]).astype(np.uint64)
            win = np.zeros(codes.shape[0], dtype=np.uint64)
            for i in range(32):
                win |= pad[i:i + codes.shape[0]] << np.uint64(2 * i)
            cache["win"] = win
        return win[base] & np.uint64((1 << (2 * L)) - 1)
    w = np.zeros(base.shape[0], dtype=np.uint64)
    for i in range(L):
        w |= codes[base + i].astype(np.uint64) << np.uint64(2 * i)
    return w


def _xor_at(words, rows, cols, delta):
    """Substitute base `cols` of reads `rows`: code ^= delta (delta in 1..3)."""
    words[rows] ^= delta.astype(np.uint64) << (2 * cols).astype(np.uint64)


def _set_at(words, rows, col, value):
    sh = np.uint64(2 * col)
    words[rows] = (words[rows] & ~(np.uint64(3) << sh)) | (value.astype(np.uint64) << sh)


def _set_at_cols(words, rows, cols, value):
    """Base `cols[i]` of read `rows[i]` := value[i]."""
    sh = (2 * cols).astype(np.uint64)
    words[rows] = (words[rows] & ~(np.uint64(3) << sh)) | (value.astype(np.uint64) << sh)


def synth_reads_packed(libs, n, seed=355, L=22, mix=None, zipf_s=1.1):
    """Packed reads (uint64 [n], L <= 32 bases each) drawn from the mixture, shuffled."""
    if L > 32:
        raise ValueError("synthetic reads are at most 32 nt")
    rng = np.random.default_rng(seed)
    mix = dict(DEFAULT_MIX if mix is None else mix)
    keys = list(mix)
    frac = np.array([mix[k] for k in keys], dtype=np.float64)
    counts = np.floor(frac / frac.sum() * n).astype(np.int64)
    counts[0] += n - counts.sum()
    out = np.empty(n, dtype=np.uint64)
    at = 0
    caches = libs.__dict__.setdefault("_window_caches", {})

    def cut(key, base, length):
        return _cut_packed(libs.codes[key][0], base, length, caches.setdefault(key, {}))

    def sample_sub(key, m, length, zipf=False):
        codes, starts = libs.codes[key]
        n_ent = len(starts) - 1
        lens = np.diff(starts)
        if zipf:
            ranks = np.minimum(rng.zipf(zipf_s, m) - 1, n_ent * 50) % n_ent
            perm = np.random.default_rng(seed + 1).permutation(n_ent)
            ent = perm[ranks]
        else:
            # entries weighted by length, so long transcripts dominate as in real data
            ent = np.searchsorted(starts, rng.integers(0, starts[-1], m), side="right") - 1
        room = lens[ent] - length
        bad = room < 0
        if bad.any():
            ok_ent = np.nonzero(lens >= length)[0]
            ent[bad] = ok_ent[rng.integers(0, ok_ent.size, int(bad.sum()))]
            room = lens[ent] - length
        off = (rng.random(m) * (room + 1)).astype(np.int64)
        return codes, starts, ent, off

    def substituted(words, frac_sub, length):
        pick = np.nonzero(rng.random(words.shape[0]) < frac_sub)[0]
        _xor_at(words, pick, rng.integers(0, length, pick.size), rng.integers(1, 4, pick.size))
        return words

    for k, m in zip(keys, counts):
        m = int(m)
        if m == 0:
            continue
        if k == "mirna_exact":
            codes, starts, ent, _ = sample_sub("mirna", m, L, zipf=True)
            lens = np.diff(starts)[ent]
            off = np.clip(2 + rng.integers(-2, 3, m), 0, lens - L)
            blk = cut("mirna", starts[ent] + off, L)
        elif k == "mirna_a2g":
            # canonical-offset miRNA reads; a seeded position inside the scored part of the mature
            # sequence is turned into G when it holds an A (about one pick in four does)
            codes, starts, ent, _ = sample_sub("mirna", m, L, zipf=True)
            lens = np.diff(starts)[ent]
            off = np.clip(2 + rng.integers(0, 2, m), 0, lens - L)
            blk = cut("mirna", starts[ent] + off, L)
            col = rng.integers(2, L - 7, m)
            is_a = ((blk >> (2 * col).astype(np.uint64)) & np.uint64(3)) == 0
            rows = np.nonzero(is_a)[0]
            _set_at_cols(blk, rows, col[rows], np.full(rows.size, 2))
        elif k == "isomir":
            codes, starts, ent, _ = sample_sub("mirna", m, L, zipf=True)
            lens = np.diff(starts)[ent]
            off = np.clip(2 + rng.integers(-1, 2, m), 0, lens - L)
            blk = cut("mirna", starts[ent] + off, L)
            kind = rng.integers(0, 3, m)
            # 0: non-templated 3' addition (last base replaced by A or T); 1: one internal
            # substitution; 2: both
            add = np.nonzero(kind != 1)[0]
            _set_at(blk, add, L - 1, np.where(rng.random(add.size) < 0.5, 0, 3))
            sub = np.nonzero(kind != 0)[0]
            _xor_at(blk, sub, rng.integers(3, L - 4, sub.size), rng.integers(1, 4, sub.size))
        elif k in ("trna", "snorna", "mrna"):
            key = {"trna": "mature_trna", "snorna": "snorna", "mrna": "mrna"}[k]
            codes, starts, ent, off = sample_sub(key, m, L)
            blk = substituted(cut(key, starts[ent] + off, L), 0.3, L)
        elif k == "rrna_ncrna":
            h = m // 2
            c1, s1, e1, o1 = sample_sub("rrna", h, L)
            c2, s2, e2, o2 = sample_sub("ncrna_others", m - h, L)
            blk = np.concatenate([cut("rrna", s1[e1] + o1, L), cut("ncrna_others", s2[e2] + o2, L)])
            blk = substituted(blk, 0.3, L)
        elif k == "polyt":
            tail = 4
            codes, starts, ent, off = sample_sub("pre_trna", m, L - tail)
            blk = cut("pre_trna", starts[ent] + off, L - tail)
            for i in range(L - tail, L):
                blk |= np.uint64(3) << np.uint64(2 * i)
        elif k == "random":
            blk = rng.integers(0, 1 << (2 * L), m, dtype=np.uint64)
        else:
            raise KeyError(k)
        out[at:at + m] = blk
        at += m
    return out[rng.permutation(n)]


VARLEN_MIX = dict(hairpin=0.35, mirna=0.25, snorna=0.1, ncrna_others=0.1, mrna=0.1, pre_trna_t=0.05, random=0.05)


def synth_reads_varlen(libs, n, seed=977, min_len=16, max_len=40, mix=None):
    """Variable-length reads ("a secondary run with lengths U{16..40}" so that the
    hairpin pass, the length windows and two-word reads are exercised; not the headline workload).
    Substrings of library entries with 0-1 substitution, poly-T trailers, random reads.
    Returns (words uint64 [2, n], lens uint8 [n])."""
    rng = np.random.default_rng(seed)
    mix = dict(VARLEN_MIX if mix is None else mix)
    keys = list(mix)
    frac = np.array([mix[k] for k in keys], dtype=np.float64)
    counts = np.floor(frac / frac.sum() * n).astype(np.int64)
    counts[0] += n - counts.sum()
    words = np.zeros((2, n), dtype=np.uint64)
    lens = rng.integers(min_len, max_len + 1, n).astype(np.int64)
    at = 0
    for k, m in zip(keys, counts):
        m = int(m)
        if m == 0:
            continue
        sl = slice(at, at + m)
        L = lens[sl]
        if k == "random":
            codes_src, base = None, None
            w0 = rng.integers(0, 1 << 63, m, dtype=np.uint64) << np.uint64(1) | rng.integers(0, 2, m, dtype=np.uint64)
            w1 = rng.integers(0, 1 << 32, m, dtype=np.uint64)
        else:
            key = "pre_trna" if k == "pre_trna_t" else k
            codes, starts = libs.codes[key]
            el = np.diff(starts)
            tail = rng.integers(3, 7, m) if k == "pre_trna_t" else np.zeros(m, dtype=np.int64)
            body = L - tail
            if k == "pre_trna_t":
                body = np.clip(body, 11, None)
                L = body + tail
                lens[sl] = L
            ent = np.searchsorted(starts, rng.integers(0, starts[-1], m), side="right") - 1
            bad = el[ent] < body
            if bad.any():
                # re-draw among entries long enough for the longest request
                ok_ent = np.nonzero(el >= body.max())[0]
                if ok_ent.size == 0:
                    ok_ent = np.array([int(np.argmax(el))])
                    body = np.minimum(body, el.max())
                    L = body + tail
                    lens[sl] = L
                ent[bad] = ok_ent[rng.integers(0, ok_ent.size, int(bad.sum()))]
            off = (rng.random(m) * (el[ent] - body + 1)).astype(np.int64)
            base = starts[ent] + off
            pad = np.concatenate([codes, np.zeros(64, dtype=np.uint8)])
            w0 = np.zeros(m, dtype=np.uint64)
            w1 = np.zeros(m, dtype=np.uint64)
            for i in range(int(L.max())):
                c = np.where(i < body, pad[base + i], 3).astype(np.uint64)   # beyond the body: the T tail
                c = np.where(i < L, c, 0).astype(np.uint64)
                if i < 32:
                    w0 |= c << np.uint64(2 * i)
                else:
                    w1 |= c << np.uint64(2 * (i - 32))
            sub = np.nonzero(rng.random(m) < 0.3)[0]
            col = (rng.random(sub.size) * body[sub]).astype(np.int64)
            delta = rng.integers(1, 4, sub.size).astype(np.uint64)
            lo = col < 32
            w0[sub[lo]] ^= delta[lo] << (2 * col[lo]).astype(np.uint64)
            w1[sub[~lo]] ^= delta[~lo] << (2 * (col[~lo] - 32)).astype(np.uint64)
        # clear everything beyond each read's length
        Lc = lens[sl]
        m0 = np.where(Lc >= 32, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << (2 * np.minimum(Lc, 31)).astype(np.uint64)) - np.uint64(1))
        m1 = np.where(Lc > 32, (np.uint64(1) << (2 * np.clip(Lc - 32, 0, 31)).astype(np.uint64)) - np.uint64(1), np.uint64(0))
        words[0, sl] = w0 & m0
        words[1, sl] = w1 & m1
        at += m
    perm = rng.permutation(n)
    return np.ascontiguousarray(words[:, perm]), lens[perm].astype(np.uint8)


def synth_reads(libs, n, seed=355, L=22, mix=None, zipf_s=1.1):
    """Same reads as uint8 codes [n, L] (small inputs: tests, smoke)."""
    w = synth_reads_packed(libs, n, seed=seed, L=L, mix=mix, zipf_s=zipf_s)
    sh = (2 * np.arange(L)).astype(np.uint64)
    return ((w[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.uint8)


def synth_quant(n, n_samples=1, seed=355):
    """Per-read, per-sample counts (uint32): mostly 1, geometric tail; with more
    than one sample some entries are zero, as after collapsing several FASTQs."""
    rng = np.random.default_rng(seed + 7)
    q = rng.geometric(0.6, size=(n, n_samples)).astype(np.uint32)
    if n_samples > 1:
        q[rng.random((n, n_samples)) < 0.3] = 0
        dead = q.sum(axis=1) == 0
        q[dead, 0] = 1
    return q


GLOBAL_CHUNK = 10_000_000


def global_read_slice(libs, n_total, lo, hi, workload="cascade", seed0=355, mix=None, n_samples=1,
                      chunk=GLOBAL_CHUNK):
    """Reads [lo, hi) of ONE seeded read set of n_total reads, without generating the rest: the set
    is defined chunk by chunk (chunk c of `chunk` reads is seeded with seed0 + c), so any rank can
    produce its own shard and the shards of every world size tile the same set (bench.py
    --scaling strong; tests/test_bench_shards.py).
    workload "varlen" -> reads of 16..40 nt (two words), else 22-mers from `mix`.
    Returns (words uint64 [W, hi - lo], lens uint8 [hi - lo], quant uint32 [hi - lo, n_samples])."""
    W = 2 if workload == "varlen" else 1
    m_out = hi - lo
    words = np.empty((W, m_out), dtype=np.uint64)
    lens = np.empty(m_out, dtype=np.uint8)
    quant = np.empty((m_out, n_samples), dtype=np.uint32)
    for c in range(lo // chunk, (max(hi, lo + 1) - 1) // chunk + 1 if hi > lo else 0):
        c_lo = c * chunk
        m = min(chunk, n_total - c_lo)
        a, b = max(lo, c_lo), min(hi, c_lo + m)
        if b <= a:
            continue
        if workload == "varlen":
            w, l = synth_reads_varlen(libs, m, seed=seed0 + 622 + c)
        else:
            w = synth_reads_packed(libs, m, seed=seed0 + c, mix=mix)[None, :]
            l = np.full(m, 22, dtype=np.uint8)
        q = synth_quant(m, n_samples, seed=seed0 + c)
        words[:, a - lo:b - lo] = w[:, a - c_lo:b - c_lo]
        lens[a - lo:b - lo] = l[a - c_lo:b - c_lo]
        quant[a - lo:b - lo] = q[a - c_lo:b - c_lo]
    return words, lens, quant
