"""-m gpu: secondary voxel layers -- traversal, touch time, incident normal -- for GpuMap and GpuNdtMap
vs the CPU oracle.  Mirrors tests/ohmtestgpu/GpuTraversalTests / GpuTouchTimeTests / GpuIncidentsTests.
Touch time and the packed incident normal are integer fields: bit exact.  Traversal: the device sums a batch's ray
lengths per voxel in integer fixed point (k_region_traversal: LDS tile of 2^-28 m units with carries, deterministic)
where the CPU adds them to a float one ray at a time: equal to the rounding of the CPU's own running sum, held to 1e-5
relative here, and bit-identical run to run."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, OccupancyMap, RayFlag, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu

LAYERS = ("occupancy", "mean", "traversal", "touch_time", "incident_normal")


def _check(om, map_, layers):
    exact = [n for n in layers if n != "traversal"]
    stats = compare_maps(om.chunks(), map_.chunks, exact, rel=1e-5)
    assert_parity(stats)
    worst = 0.0
    for key, cpu in om.chunks().items():
        g = map_.chunks[key]["traversal"]
        c = cpu["traversal"]
        assert np.array_equal(c != 0, g != 0)
        nz = c != 0
        if nz.any():
            worst = max(worst, float(np.max(np.abs(g[nz] - c[nz]) / np.maximum(np.abs(c[nz]), 1e-3))))
    assert worst < 1e-5, worst


def test_occupancy_secondary_layers(gpu):
    rays = synth.rays_c1(n=30000, max_range=10.0)
    ts = 100.0 + 0.001 * np.arange(rays.shape[0] // 2, dtype=np.float64)
    map_ = OccupancyMap(0.1, layers=LAYERS)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    for i in range(0, rays.shape[0], 20000):
        chunk, tchunk = rays[i:i + 20000], ts[i // 2:(i + 20000) // 2]
        assert gm.integrateRays(chunk, timestamps=tchunk) == chunk.shape[0]
        om.integrate_occupancy(chunk, timestamps=tchunk)
    gm.syncVoxels()
    _check(om, map_, LAYERS)


def test_traversal_reference_rays(gpu):
    # tests/ohmtestcommon/TraversalTest.cpp:22-188: rays into / through the voxel at the origin, one call per ray
    map_ = OccupancyMap(0.1, layers=("occupancy", "traversal"))
    map_.setOrigin((-0.05, -0.05, -0.05))
    gm = GpuMap(map_)
    om = make_oracle(map_)
    dirs = [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (1, 1, 0), (-1, 0, 1), (1, 1, 1), (-1, 1, -1)]
    for d in dirs:
        for ray in (np.array([d, (0, 0, 0)], dtype=np.float64), np.array([d, tuple(-v for v in d)], dtype=np.float64)):
            gm.integrateRays(ray)
            om.integrate_occupancy(ray)
    gm.syncVoxels()
    stats = compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True)
    assert_parity(stats)
    for key, cpu in om.chunks().items():
        assert np.allclose(map_.chunks[key]["traversal"], cpu["traversal"], rtol=1e-5, atol=1e-6)


def test_ndt_secondary_layers(gpu):
    rays = np.concatenate([synth.rays_c2(n=12000, seed=600 + k) for k in range(2)])
    ts = 5.0 + 0.01 * np.arange(rays.shape[0] // 2, dtype=np.float64)
    map_ = OccupancyMap(0.2, layers=("occupancy", "traversal", "touch_time", "incident_normal"))
    gm = GpuNdtMap(map_)
    om = make_oracle(map_)
    om.set_ndt(adaptation_rate=gm.adaptation_rate)
    for i in range(0, rays.shape[0], 24000):
        chunk, tchunk = rays[i:i + 24000], ts[i // 2:(i + 24000) // 2]
        assert gm.integrateRays(chunk, timestamps=tchunk) == chunk.shape[0]
        om.integrate_ndt(chunk, timestamps=tchunk)
    gm.syncVoxels()
    _check(om, map_, list(map_.layers))


def test_traversal_is_deterministic(gpu):
    rays = synth.rays_c1(n=40000, max_range=8.0, seed=77)
    results = []
    for _ in range(3):
        map_ = OccupancyMap(0.1, layers=("occupancy", "traversal"))
        gm = GpuMap(map_)
        gm.integrateRays(rays)
        gm.integrateRays(rays[:30000])
        gm.syncVoxels()
        results.append({k: v["traversal"].copy() for k, v in map_.chunks.items()})
    for other in results[1:]:
        assert other.keys() == results[0].keys()
        for key, tile in results[0].items():
            assert np.array_equal(tile.view(np.uint32), other[key].view(np.uint32))


@pytest.mark.parametrize("flags", [int(RayFlag.kRfEndPointAsFree), int(RayFlag.kRfExcludeOrigin),
                                   int(RayFlag.kRfEndPointAsFree | RayFlag.kRfExcludeOrigin),
                                   int(RayFlag.kRfExcludeSample)])
def test_traversal_with_ray_flags(gpu, flags):
    """The traversal pass (traversal_kernels.h) walks the same segments as the count walk: an end voxel that is part of
    the walk exits at the ray's length, an excluded origin voxel gets nothing while its range still passes."""
    rays = synth.rays_c1(n=20000, max_range=9.0, seed=4100 + flags)
    map_ = OccupancyMap(0.1, layers=("occupancy", "traversal"))
    gm = GpuMap(map_)
    om = make_oracle(map_)
    assert gm.integrateRays(rays, ray_update_flags=flags) == rays.shape[0]
    om.integrate_occupancy(rays, flags=flags)
    gm.syncVoxels()
    _check(om, map_, ("occupancy", "traversal"))


@pytest.mark.parametrize("resolution,dims", [(5.0, (16, 16, 16)), (0.37, (20, 32, 9)), (0.02, (32, 32, 32))])
def test_traversal_voxel_sizes_and_region_shapes(gpu, resolution, dims):
    """The LDS tile counts 2^-28 m units unless a voxel diagonal would not fit 31 bits of them (5 m voxels: 2^-27)."""
    # (few rays per batch: the CPU's float32 running sum in the sensor's voxel drifts by 5e-5 after 32 000 adds of ~1 m
    # to a sum of 29 km -- the device's integer sum does not -- and this test is about the tile unit, not that)
    rays = synth.rays_c1(n=8000, max_range=60.0 * resolution, seed=int(resolution * 1000))
    map_ = OccupancyMap(resolution, dims, layers=("occupancy", "traversal"))
    gm = GpuMap(map_)
    om = make_oracle(map_)
    for k in range(2):
        part = rays[k * 8000:(k + 1) * 8000]
        assert gm.integrateRays(part) == part.shape[0]
        om.integrate_occupancy(part)
    gm.syncVoxels()
    _check(om, map_, ("occupancy", "traversal"))


def test_traversal_tile_words_wrap(gpu):
    """60 000 rays leave one voxel: its sum in one batch is ~3 km, 180 times what a 32-bit tile word holds (2^32 x 2^-28 m
    = 16 m) -- the carries to the 64-bit accumulator keep it exact.  Compared with the sum in float64 of the float lengths
    the CPU mapper would add, which the oracle's float running sum only approximates at this size."""
    rays = synth.rays_c1(n=60000, max_range=9.0, seed=99)
    map_ = OccupancyMap(0.1, layers=("occupancy", "traversal"))
    gm = GpuMap(map_)
    assert gm.integrateRays(rays) == rays.shape[0]
    gm.syncVoxels()
    om = make_oracle(map_)
    om.integrate_occupancy(rays)
    origin_key = om.voxel_key(tuple(rays[0]))
    got = float(map_.chunks[origin_key[0]]["traversal"][origin_key[1][0] + 32 * origin_key[1][1] + 1024 * origin_key[1][2]])
    ref = float(om.chunks()[origin_key[0]]["traversal"][origin_key[1][0] + 32 * origin_key[1][1] + 1024 * origin_key[1][2]])
    assert got > 16.0 * 20
    assert abs(got - ref) <= 2e-4 * ref  # (the float32 running sum of 60 000 terms is itself only good to ~1e-4)
    # exact reference for that voxel: every ray starts in it, so its share is the range of the ray's first step
    o = rays[0::2].astype(np.float64)
    d = rays[1::2] - o
    length = np.sqrt((d * d).sum(axis=1))
    res = 0.1
    lo = np.floor(o / res) * res  # the voxel [lo, lo + res) holding the sensor (voxel walls lie on multiples of res)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(d > 0, (lo + res - o) / d, np.where(d < 0, (lo - o) / d, np.inf))
    first = np.minimum(t.min(axis=1), 1.0) * length
    exact = float(first.astype(np.float32).astype(np.float64).sum())
    assert abs(got - exact) <= 1e-6 * exact, (got, exact)
