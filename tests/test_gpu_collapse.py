"""GPU (-m gpu): mrg_collapse_run (quantReads.py:3-24 on the device) -- the duplication-aware path (csrc/collapse.hip:
LDS aggregation, ordered partition, monotone-hash reduce) and the general radix-sort path (csrc/prims.hip) against a
dict keyed by the read, on skewed, multi-length, multi-sample and adversarial batches."""
import numpy as np
import pytest

from mirge_amd import ingest

pytestmark = pytest.mark.gpu


def _dict_collapse(words, lens, sample, S):
    want = {}
    for w, l, s in zip(words.tolist(), lens.tolist(), sample.tolist()):
        want.setdefault((l, w), [0] * S)[s] += 1
    return want


def _check(out, want, S, lens, sample):
    keys = list(zip(out["lens"].tolist(), out["words"][0].tolist()))
    assert keys == sorted(want), "uniques ordered by (length, packed bases), each once"
    assert out["quant"].tolist() == [want[k] for k in keys]
    hist = {}
    for l, s in zip(lens.tolist(), sample.tolist()):
        hist.setdefault(l, [0] * S)[s] += 1
    assert out["length_hist"] == hist


def _batch(rng, n, lens_choice, pool_size, zipf, S):
    """n reads drawn (Zipf) from a pool of random sequences of the given lengths + a tail of singletons."""
    pool_len = rng.choice(lens_choice, pool_size).astype(np.uint8)
    pool = rng.integers(0, 2 ** 63, pool_size, dtype=np.uint64) & ((np.uint64(1) << (2 * pool_len.astype(np.uint64))) - np.uint64(1))
    pick = rng.zipf(zipf, n) % pool_size
    words, lens = pool[pick].copy(), pool_len[pick].copy()
    single = rng.random(n) < 0.15
    slen = rng.choice(lens_choice, n).astype(np.uint8)
    sw = rng.integers(0, 2 ** 63, n, dtype=np.uint64) & ((np.uint64(1) << (2 * slen.astype(np.uint64))) - np.uint64(1))
    words[single], lens[single] = sw[single], slen[single]
    sample = rng.integers(0, S, n).astype(np.uint16)
    return words, lens, sample


@pytest.mark.parametrize("case", ["one_length", "lengths_16_29", "three_samples", "sixteen_samples", "tiny", "hot_key", "short_reads"])
def test_fast_path_equals_a_dict(native_lib, case):
    from mirge_amd.engine import Engine
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    eng = Engine(0)
    S = 1
    if case == "one_length":
        words, lens, sample = _batch(rng, 400_000, [22], 30_000, 1.2, 1)
    elif case == "lengths_16_29":
        words, lens, sample = _batch(rng, 300_000, list(range(16, 30)), 20_000, 1.3, 1)
    elif case == "three_samples":
        S = 3
        words, lens, sample = _batch(rng, 300_000, [18, 21, 22, 23, 25], 20_000, 1.3, 3)
    elif case == "sixteen_samples":
        S = 16
        words, lens, sample = _batch(rng, 200_000, [20, 22, 27], 5_000, 1.2, 16)   # 2 x 27 + 4 sample bits = 58
    elif case == "tiny":
        words, lens, sample = _batch(rng, 37, [22], 5, 1.5, 1)
    elif case == "hot_key":
        # one sequence is 60 % of the batch (far more copies per chunk than a pair's 14-bit count holds), another 20 %
        words, lens, sample = _batch(rng, 600_000, [22], 10_000, 1.5, 1)
        hot = rng.random(600_000)
        words[hot < 0.6] = np.uint64(0x2B3C4D5E6F7)
        words[(hot >= 0.6) & (hot < 0.8)] = np.uint64(0x2B3C4D5E6F6)
    else:
        words, lens, sample = _batch(rng, 100_000, [1, 3, 4, 5, 8, 12], 3_000, 1.3, 1)   # reads shorter than the 8 bucket bits
    want = _dict_collapse(words, lens, sample, S)
    out = ingest.collapse(eng, words[None, :], lens, None, sample if S > 1 else None, n_samples=S, max_len=int(lens.max()))
    _check(out, want, S, lens, sample)
    # and the general path agrees (same order, same counts)
    eng.set_option("collapse_fast", 0)
    out2 = ingest.collapse(eng, words[None, :], lens, None, sample if S > 1 else None, n_samples=S, max_len=int(lens.max()))
    assert np.array_equal(out["words"], out2["words"]) and np.array_equal(out["lens"], out2["lens"])
    assert np.array_equal(out["quant"], out2["quant"]) and out["length_hist"] == out2["length_hist"]


def test_batches_the_fast_path_hands_over_are_still_right(native_lib):
    """Adversarial for the ordered partition: 200 000 distinct 22-mers that share their 20 most significant bases (one
    final bucket, one run of occupied slots) overflow the reduce table -- the call answers through the general path;
    and a batch with reads of 30..32 nt or more than 16 distinct lengths never enters the fast path."""
    from mirge_amd.engine import Engine
    rng = np.random.default_rng(11)
    eng = Engine(0)
    n = 300_000
    low = rng.integers(0, 200_000, n, dtype=np.uint64)
    words = (np.uint64(0x2A5F1C2) << np.uint64(18)) | low       # 44-bit keys: the top 26 bits constant
    lens = np.full(n, 22, np.uint8)
    sample = np.zeros(n, np.uint16)
    out = ingest.collapse(eng, words[None, :], lens, None, None, n_samples=1, max_len=22)
    _check(out, _dict_collapse(words, lens, sample, 1), 1, lens, sample)
    words, lens, sample = _batch(rng, 100_000, list(range(10, 33)), 8_000, 1.3, 2)
    out = ingest.collapse(eng, words[None, :], lens, None, sample, n_samples=2, max_len=32)
    _check(out, _dict_collapse(words, lens, sample, 2), 2, lens, sample)


def test_collapse_errors(native_lib):
    import torch
    from mirge_amd._native import MirgeAmdError
    from mirge_amd.engine import Engine
    eng = Engine(0)
    rng = np.random.default_rng(2)
    words, lens, sample = _batch(rng, 5000, [22], 4000, 1.5, 2)
    sample[17] = 2   # not below n_samples
    with pytest.raises(MirgeAmdError):
        ingest.collapse(eng, words[None, :], lens, None, sample, n_samples=2, max_len=22)


def test_prims_scan_and_sort_through_the_general_path(native_lib):
    """The library's own prefix sums and radix sort (csrc/prims.hip) at sizes that cross their tile and level
    boundaries: the general path on 4097, 65 537 and 1.1 M records with two words, N masks and three samples."""
    from mirge_amd.engine import Engine
    eng = Engine(0)
    eng.set_option("collapse_fast", 0)
    rng = np.random.default_rng(3)
    for n in (1, 63, 4096, 4097, 65_537, 1_100_000):
        pool = rng.integers(0, 2 ** 62, (2, max(2, n // 7)), dtype=np.uint64)
        pool_l = rng.integers(33, 65, pool.shape[1]).astype(np.uint8)
        pool[1] &= (np.uint64(1) << (2 * (pool_l.astype(np.uint64) - np.uint64(32)))) - np.uint64(1)
        pool_n = np.zeros_like(pool)
        pool_n[0, ::5] = np.uint64(1) << np.uint64(2 * 7)
        pool[0, ::5] &= ~(np.uint64(3) << np.uint64(2 * 7))
        pick = rng.integers(0, pool.shape[1], n)
        words, nmask, lens = np.ascontiguousarray(pool[:, pick]), np.ascontiguousarray(pool_n[:, pick]), pool_l[pick]
        sample = rng.integers(0, 3, n).astype(np.uint16)
        want = {}
        for i in range(n):
            want.setdefault((int(lens[i]), int(nmask[1, i]), int(nmask[0, i]), int(words[1, i]), int(words[0, i])), [0, 0, 0])[sample[i]] += 1
        out = ingest.collapse(eng, words, lens, nmask, sample, n_samples=3, max_len=64)
        keys = [(int(out["lens"][i]), int(out["nmask"][1, i]), int(out["nmask"][0, i]), int(out["words"][1, i]), int(out["words"][0, i]))
                for i in range(out["lens"].size)]
        assert keys == sorted(want) and out["quant"].tolist() == [want[k] for k in keys], n
