"""-m gpu: secondary voxel layers -- traversal, touch time, incident normal -- for GpuMap and GpuNdtMap
vs the CPU oracle.  Mirrors tests/ohmtestgpu/GpuTraversalTests / GpuTouchTimeTests / GpuIncidentsTests.
Touch time and the packed incident normal are integer fields: bit exact.  Traversal: the device sums a batch's ray
lengths per voxel exactly (fixed-point integer atomics, deterministic) where the CPU adds them to a float one ray at a
time: equal to the rounding of the CPU's own running sum, held to 1e-5 relative here, and bit-identical run to run."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, OccupancyMap, synth

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
