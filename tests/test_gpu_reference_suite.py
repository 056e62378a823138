"""-m gpu: restatements BY NAME of the reference's GPU test cases that had none yet:

* GpuMap.PopulateMultiple  (tests/ohmtestgpu/GpuMapTest.cpp:400-457): four GPU maps alive in one process, fed interleaved
  batches -- two persistent wrappers, one transient wrapper re-created around a persistent OccupancyMap for every batch,
  and a fully transient map -- must not disturb one another; the first three end up equal.
* GpuMap.CheckBadRays      (tests/ohmtestgpu/GpuMapTest.cpp:817-835): rays shorter than the walk's length epsilon whose
  ends lie in different voxels once sent the region walk into an infinite loop; the batch has to finish within 5 s and
  match the CPU mapper, voxel means included.

Same inputs as the reference where they are deterministic (the bad ray, resolutions, batch and cache sizes); its
std::mt19937 ray cloud is replaced by the repository's splitmix generator (same extents and count).  The bar is this
project's: bit-exact against the CPU oracle, which is stricter than the reference's compareMaps (1 % of voxels may be off
by half a hit there, GpuMapTest.cpp:211-212)."""
import time

import numpy as np
import pytest

from ohm_amd import GpuMap, OccupancyMap, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def test_populate_multiple(gpu):
    map_extents, resolution = 50.0, 0.25
    ray_count, batch_size = 1024 * 8, 1024 * 2            # batch_size counts POINTS, as in the reference
    cache_bytes = 200 << 20                                # GpuCache::kMiB * 200
    rays = synth.random_rays(ray_count, extent=map_extents, seed=4242)
    rays[0::2] = 0.05                                      # glm::dvec3(0.05) origins
    map1 = OccupancyMap(resolution, (32, 32, 32))
    gpu_map1 = GpuMap(map1, True, batch_size, cache_bytes)
    map2 = OccupancyMap(resolution, (32, 32, 32))
    gpu_map2 = GpuMap(map2, True, batch_size, cache_bytes)
    map3 = OccupancyMap(resolution, (32, 32, 32))           # persistent map, transient GpuMap wrapper per batch
    om = make_oracle(map1)
    for i in range(0, rays.shape[0], batch_size):
        batch = rays[i:i + batch_size]
        assert gpu_map1.integrateRays(batch) == batch.shape[0]
        assert gpu_map2.integrateRays(batch) == batch.shape[0]
        gpu_map3 = GpuMap(map3, True, batch_size, cache_bytes)   # uploads what map3 holds so far
        assert gpu_map3.integrateRays(batch) == batch.shape[0]
        gpu_map3.syncVoxels()
        gpu_map3.close()
        map4 = OccupancyMap(resolution, (32, 32, 32))            # fourth, fully transient map
        gpu_map4 = GpuMap(map4, True, batch_size, cache_bytes)
        assert gpu_map4.integrateRays(batch) == batch.shape[0]
        gpu_map4.syncVoxels()
        gpu_map4.close()
        single = make_oracle(map4)
        single.integrate_occupancy(batch)
        assert_parity(compare_maps(single.chunks(), map4.chunks, ["occupancy"], exact_float=True))
        om.integrate_occupancy(batch)
    gpu_map1.syncVoxels()
    gpu_map2.syncVoxels()
    gpu_map1.close()
    gpu_map2.close()
    expect = om.chunks()
    for other in (map1, map2, map3):                        # compareMaps(map1, map2); compareMaps(map1, map3) -- and the CPU
        assert_parity(compare_maps(expect, other.chunks, ["occupancy"], exact_float=True))
    assert len(expect) > 100


def test_check_bad_rays(gpu):
    rays = np.array([[-2.699077907025583, -1.5999031032475868, 1.0755428728082643],
                     [-2.6998157732186034, -1.6000298354709896, 1.0756803244026165]], dtype=np.float64)
    assert np.linalg.norm(rays[1] - rays[0]) < 1e-3         # under the walk's length epsilon ...
    assert not np.array_equal(np.floor(rays[0] / 0.1), np.floor(rays[1] / 0.1))  # ... yet in different voxels
    layers = ("occupancy", "mean")                          # params.voxel_means = true
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=layers)
    t0 = time.perf_counter()
    gm = GpuMap(map_, True, 32)                             # params.batch_size = 32
    assert gm.integrateRays(rays) == 2
    gm.syncVoxels()
    elapsed = time.perf_counter() - t0
    gm.close()
    assert elapsed < 5.0, "ASSERT_DURATION_LE(5, ...)"
    om = make_oracle(map_)
    om.integrate_occupancy(rays)
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))
    # the same ray among ordinary ones, and repeated: nothing hangs, nothing drifts
    many = np.concatenate([synth.rays_c0(n=500, length=3.0, seed=3)] + [rays] * 64)
    map2 = OccupancyMap(0.1, (32, 32, 32), layers=layers)
    gm2 = GpuMap(map2, True, 32)
    t0 = time.perf_counter()
    for i in range(0, many.shape[0], 32):
        gm2.integrateRays(many[i:i + 32])
    gm2.syncVoxels()
    assert time.perf_counter() - t0 < 5.0
    gm2.close()
    om2 = make_oracle(map2)
    om2.integrate_occupancy(many)
    assert_parity(compare_maps(om2.chunks(), map2.chunks, list(layers), exact_float=True))
