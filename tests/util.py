"""Shared builders for the test suite: a small synthetic world (libraries,
indexes, reads) and adapters between the columnar product API and the oracle."""
import numpy as np

from mirge_amd import pack, synth
from mirge_amd.engine import MIRGE_PASS_TABLE
from mirge_amd.index import FmIndex

LIB_ORDER = list(synth.LIB_KEYS)


def pass_dicts(spike_in=False, order=LIB_ORDER):
    rows = MIRGE_PASS_TABLE[:10 if spike_in else 9]
    return [dict(lib=order.index(k), min_len=a, max_len=b, seed_len=s, max_mm_seed=ms,
                 max_mm_total=mt, trim5=t5, trim3=t3, poly_t=pt)
            for (k, a, b, s, ms, mt, t5, t3, pt) in rows]


def mixed_reads(libs, n_fixed=3000, n_var=500, seed=1, with_n=True, max_var_len=44):
    """22-mers from the standard mixture plus variable-length (16..44 nt) reads cut
    from hairpin / miRNA / ncRNA entries with 0-2 edits (some of them N)."""
    reads = [synth.codes_to_str(c) for c in synth.synth_reads(libs, n_fixed, seed=seed + 354)]
    rng = np.random.default_rng(seed)
    alphabet = "ACGTN" if with_n else "ACGT"
    for _ in range(n_var):
        key = ["hairpin", "mirna", "ncrna_others", "pre_trna"][int(rng.integers(0, 4))]
        seqs = libs.libs[key][1]
        s = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(16, max_var_len + 1))
        if len(s) < ln:
            continue
        o = int(rng.integers(0, len(s) - ln + 1))
        r = list(s[o:o + ln])
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, ln))] = alphabet[int(rng.integers(0, len(alphabet)))]
        if rng.random() < 0.2:
            r += list("T" * int(rng.integers(3, 7)))
        reads.append("".join(r)[:60 if max_var_len > 32 else 32])
    return list(dict.fromkeys(reads))


class World:
    def __init__(self, scale=0.03, seed=20181, n_fixed=3000, n_var=500, with_n=True, max_var_len=44):
        self.libs = synth.SynthLibraries(seed=seed, scale=scale)
        self.index = {k: FmIndex.build(*self.libs.libs[k]) for k in LIB_ORDER}
        self.views = [self.index[k].view() for k in LIB_ORDER]
        self.reads = mixed_reads(self.libs, n_fixed, n_var, with_n=with_n, max_var_len=max_var_len)
        self.words, self.lens, self.nmask = pack.pack_reads(self.reads)
        self.passes = pass_dicts()
        self.n_mirna = self.index["mirna"].n_ref


def big_library_case(seed=9, n_entries=12, entry_len=400_000, n_reads=6000):
    """A library past 4^11 bases, so its index carries the big (k = 12) jump table, and reads cut
    from it (exact, 1-2 substitutions, random).  Returns (names, seqs, reads)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [acgt[rng.integers(0, 4, entry_len)].tobytes().decode() for _ in range(n_entries)]
    names = ["big%d" % i for i in range(n_entries)]
    reads = []
    for _ in range(n_reads):
        s = seqs[int(rng.integers(0, n_entries))]
        L = int(rng.integers(16, 41))
        o = int(rng.integers(0, entry_len - L))
        r = list(s[o:o + L])
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, L))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(r))
    reads += ["".join("ACGT"[c] for c in rng.integers(0, 4, 22)) for _ in range(500)]
    # an N inside two entries (three N-free segments each side) and reads that would only align
    # across an entry boundary, across the N, right at an entry's first / last bases
    for e in (2, 7):
        s = seqs[e]
        seqs[e] = s[:123_457] + "N" + s[123_458:]
    for e in (2, 7):
        s = seqs[e]
        for L in (18, 22, 30):
            reads += [s[123_457 - L + 4:123_457 + 4].replace("N", "A"), s[123_458:123_458 + L],
                      s[123_457 - L:123_457], s[:L], s[-L:], s[1:L + 1], s[-L - 1:-1]]
    for e in range(3):
        for L in (18, 22, 30):
            reads.append(seqs[e][-(L // 2):] + seqs[e + 1][:L - L // 2])
    return names, seqs, reads


def special_reads_of_big_case(reads):
    """The boundary reads appended last by big_library_case."""
    return reads[-(2 * 3 * 7 + 3 * 3):]


BIG_PASSES = [dict(lib=0, seed_len=28, max_mm_seed=0, max_mm_total=2, min_len=0, max_len=25),     # -n 0: k = 12
              dict(lib=0, seed_len=28, max_mm_seed=1, max_mm_total=2, min_len=0, max_len=255),    # -n 1: 11-nt pieces
              dict(lib=0, seed_len=1024, max_mm_seed=2, max_mm_total=2, min_len=0, max_len=255)]  # -v 2: short pieces
