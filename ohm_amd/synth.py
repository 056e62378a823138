"""Synthetic ray sets for the BASELINE configs (SURVEY.md 8d / BASELINE.md 2).

All randomness comes from an integer hash (splitmix64 -> double in [0,1)) so the same rays are produced on every
machine and numpy version.  Trig goes through numpy; both sides of every comparison consume the SAME generated array
in the same run, so last-bit libm differences between machines cannot affect parity.
"""
import numpy as np

SEED_BASE = 0x6F686D5F

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, index, stream=0):
    """Deterministic U[0,1) per (seed, stream, index)."""
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15))
        x = splitmix64(np.asarray(index, dtype=np.uint64) ^ key)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _pairs(origin, ends):
    n = ends.shape[0]
    rays = np.empty((2 * n, 3), dtype=np.float64)
    rays[0::2] = origin
    rays[1::2] = ends
    return rays


def rays_c0(n=100_000, origin=(0.05, 0.05, 0.05), length=10.0, seed=SEED_BASE + 0):
    """C0: uniform directions on the sphere, fixed length, one origin."""
    i = np.arange(n, dtype=np.uint64)
    z = 2.0 * uniform01(seed, i, 0) - 1.0
    phi = 2.0 * np.pi * uniform01(seed, i, 1)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    d = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)
    o = np.asarray(origin, dtype=np.float64)
    return _pairs(o, o + length * d)


def lidar_directions(n, beams=64, azimuths=15625, elev_min_deg=-24.8, elev_span_deg=26.8, first=0):
    idx = np.arange(first, first + n, dtype=np.int64)
    b = idx % beams
    k = (idx // beams) % azimuths
    elev = np.deg2rad(elev_min_deg + b * (elev_span_deg / (beams - 1)))
    az = 2.0 * np.pi * k / azimuths
    ce = np.cos(elev)
    return np.stack([ce * np.cos(az), ce * np.sin(az), np.sin(elev)], axis=1), idx


def rays_c1(n=1_000_000, origin=(0.05, 0.05, 0.05), max_range=30.0, seed=SEED_BASE + 1, first=0):
    """C1: 64-beam spinning lidar, ranges 0.25..1.0 * max_range."""
    d, idx = lidar_directions(n, first=first)
    u = uniform01(seed, idx.astype(np.uint64), 0)
    r = max_range * (0.25 + 0.75 * u)
    o = np.asarray(origin, dtype=np.float64)
    return _pairs(o, o + d * r[:, None])


def _room_range(d, half=20.0, max_range=30.0):
    with np.errstate(divide="ignore"):
        t = np.min(np.where(np.abs(d) > 1e-12, half / np.abs(d), np.inf), axis=1)
    return np.minimum(t, max_range)


def rays_c2(n=1_000_000, origin=(0.05, 0.05, 0.05), seed=SEED_BASE + 2, first=0, noise=0.02):
    """C2: same lidar pattern, ranges = hit on a 40 m box room (<= 30 m) + N(0, 2 cm) noise."""
    d, idx = lidar_directions(n, first=first)
    r = _room_range(d)
    u1 = np.maximum(uniform01(seed, idx.astype(np.uint64), 0), 1e-300)
    u2 = uniform01(seed, idx.astype(np.uint64), 1)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    r = r + noise * g
    o = np.asarray(origin, dtype=np.float64)
    return _pairs(o, o + d * r[:, None])


def rays_c3(n=4_000_000, origin=(0.05, 0.05, 0.05), seed=SEED_BASE + 3):
    """C3: 4 revolutions of the C2 pattern (TSDF config)."""
    return rays_c2(n=n, origin=origin, seed=seed)


C4_ORIGINS = [(x, y, 0.05) for y in (-20.0, 20.0) for x in (-60.0, -20.0, 20.0, 60.0)]


def rays_c4_shard(rank, n=1_000_000, seed=SEED_BASE + 4):
    """C4: the shard of sensor origin `rank` (0..7): the C1 pattern from that origin, own PRNG stream."""
    return rays_c1(n=n, origin=C4_ORIGINS[rank % len(C4_ORIGINS)], seed=seed + rank)


def random_rays(n, extent=10.0, seed=1, origin_spread=0.0):
    """Random segments in a cube (like tests/ohmtestgpu/GpuMapTest.cpp's PopulateSmall/Large ray sets)."""
    i = np.arange(n, dtype=np.uint64)
    ends = np.stack([(2.0 * uniform01(seed, i, s) - 1.0) * extent for s in range(3)], axis=1)
    starts = np.stack([(2.0 * uniform01(seed, i, 3 + s) - 1.0) * origin_spread + 0.05 for s in range(3)], axis=1)
    rays = np.empty((2 * n, 3), dtype=np.float64)
    rays[0::2] = starts
    rays[1::2] = ends
    return rays
