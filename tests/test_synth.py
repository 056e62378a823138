"""Synthetic ray generators are deterministic and have the documented shapes (CPU only)."""
import numpy as np

from ohm_amd import synth


def test_generators_shapes_and_determinism():
    a = synth.rays_c1(n=4096)
    b = synth.rays_c1(n=4096)
    assert a.shape == (8192, 3) and np.array_equal(a, b)
    lengths = np.linalg.norm(a[1::2] - a[0::2], axis=1)
    assert lengths.min() >= 7.5 - 1e-9 and lengths.max() <= 30.0 + 1e-9
    c0 = synth.rays_c0(n=1000)
    assert np.allclose(np.linalg.norm(c0[1::2] - c0[0::2], axis=1), 10.0)
    c2 = synth.rays_c2(n=2048)
    assert np.linalg.norm(c2[1::2] - c2[0::2], axis=1).max() <= 30.0 + 0.2
    s0, s1 = synth.rays_c4_shard(0, n=128), synth.rays_c4_shard(1, n=128)
    assert not np.array_equal(s0, s1) and tuple(s0[0]) == synth.C4_ORIGINS[0]


def test_hash_prng_known_values():
    # splitmix64 reference vector (seed 0 -> first output 0xE220A8397B1DCDAF)
    assert int(synth.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF
    u = synth.uniform01(1, np.arange(10000, dtype=np.uint64), 0)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02
