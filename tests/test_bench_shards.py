"""bench.py --scaling strong: every rank builds its own contiguous shard of ONE seeded read set
(mirge_amd.synth.global_read_slice + mirge_amd.dist.shard_bounds); the shards of any world size
tile exactly the set a single process generates."""
import numpy as np

from mirge_amd import synth
from mirge_amd.dist import shard_bounds


def test_shards_tile_the_global_read_set():
    libs = synth.SynthLibraries(scale=0.02)
    n, chunk = 2500, 700   # several chunks, a ragged last one
    for workload in ("cascade", "varlen"):
        w1, l1, q1 = synth.global_read_slice(libs, n, 0, n, workload=workload, n_samples=2, chunk=chunk)
        assert w1.shape == ((2 if workload == "varlen" else 1), n)
        for world in (2, 3, 8):
            parts = [synth.global_read_slice(libs, n, *shard_bounds(n, r, world), workload=workload,
                                             n_samples=2, chunk=chunk) for r in range(world)]
            assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), w1)
            assert np.array_equal(np.concatenate([p[1] for p in parts]), l1)
            assert np.array_equal(np.concatenate([p[2] for p in parts]), q1)
    # an empty shard (more ranks than reads) is legal
    w, l, q = synth.global_read_slice(libs, 3, 3, 3)
    assert w.shape == (1, 0) and l.shape == (0,) and q.shape == (0, 1)
