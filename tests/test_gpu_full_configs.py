"""-m gpu: the BASELINE.json configurations at FULL size.

C1 (1 M lidar rays, 0.1 m) and C2 (NDT, 1 M rays, 0.2 m) are compared voxel-for-voxel with the CPU oracle (a few
seconds of CPU each).  C3 (TSDF, 0.05 m, 4 M rays, 2.7 x 10^9 voxel visits) is compared voxel-for-voxel with the oracle
too (about a minute of CPU) and additionally checked through size-independent properties: the exact voxel-visit count
(closed form from the voxel keys), and batch-split invariance (integrating in one call or in four must give
bit-identical maps because the device applies every voxel's events in ray order)."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, GpuTsdfMap, OccupancyMap, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def expected_visits(rays, resolution, include_end):
    """Sum over rays of (Manhattan voxel distance + 1): what the CPU walk reports (ohm/LineWalkCompute.h:345-413)."""
    g = np.floor(rays / resolution).astype(np.int64)  # map origin 0: global voxel coordinate (ohm/MapCoord.h)
    manhattan = np.abs(g[1::2] - g[0::2]).sum(axis=1)
    return int(manhattan.sum() + (len(manhattan) if include_end else len(manhattan)))


def test_c1_occupancy_full_vs_oracle(gpu):
    rays = synth.rays_c1()
    map_ = OccupancyMap(0.1)
    gm = GpuMap(map_, gpu_mem_size=2 << 30)
    assert gm.integrateRays(rays) == rays.shape[0]
    gm.syncVoxels()
    st = gm.stats()
    assert st["voxel_visits"] == expected_visits(rays, 0.1, False)
    om = make_oracle(map_)
    om.integrate_occupancy(rays)
    assert om.visit_count() == st["voxel_visits"]
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))


def test_c2_ndt_full_vs_oracle(gpu):
    rays = synth.rays_c2()
    map_ = OccupancyMap(0.2)
    gm = GpuNdtMap(map_, gpu_mem_size=4 << 30)
    assert gm.integrateRays(rays) == rays.shape[0]
    gm.syncVoxels()
    om = make_oracle(map_)
    om.set_ndt(adaptation_rate=gm.adaptation_rate)
    om.integrate_ndt(rays)
    assert om.visit_count() == gm.stats()["voxel_visits"]
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean", "covariance"], rel=1e-5))


def test_c3_tsdf_full_vs_oracle_and_properties(gpu):
    rays = synth.rays_c3()
    assert rays.shape[0] == 8_000_000
    # full size, one call: bit exact against the CPU oracle, visit count equal to the closed form
    map_a = OccupancyMap(0.05, layers=("tsdf",))
    gm_a = GpuTsdfMap(map_a, gpu_mem_size=16 << 30)
    assert gm_a.integrateRays(rays) == rays.shape[0]
    assert gm_a.stats()["voxel_visits"] == expected_visits(rays, 0.05, True)
    gm_a.syncVoxels()
    gm_a.close()
    om = make_oracle(map_a)
    om.integrate_tsdf(rays)
    assert om.visit_count() == expected_visits(rays, 0.05, True)
    assert_parity(compare_maps(om.chunks(), map_a.chunks, ["tsdf"], exact_float=True))
    del om
    # one call vs four calls must agree bit for bit
    map_b = OccupancyMap(0.05, layers=("tsdf",))
    gm_b = GpuTsdfMap(map_b, gpu_mem_size=16 << 30)
    for i in range(0, rays.shape[0], 2_000_000):
        assert gm_b.integrateRays(rays[i:i + 2_000_000]) == 2_000_000
    gm_b.syncVoxels()
    gm_b.close()
    assert set(map_a.chunks) == set(map_b.chunks)
    for key, layers in map_a.chunks.items():
        assert np.array_equal(layers["tsdf"].view(np.uint32), map_b.chunks[key]["tsdf"].view(np.uint32))
    w = np.concatenate([c["tsdf"].reshape(-1, 2)[:, 0] for c in map_a.chunks.values()])
    d = np.concatenate([c["tsdf"].reshape(-1, 2)[:, 1] for c in map_a.chunks.values()])
    assert w.max() <= 1e4 and np.all(np.abs(d) <= np.float32(0.1))


def test_c4_shards_one_call_of_4m_rays_equals_four_calls(gpu):
    """C4's ray sets (one sensor origin per shard): four shards through ONE map in a single 4 M-ray call and in four
    1 M-ray calls.  Size-independent properties at a size the oracle is too slow for: the exact visit count and
    bit-identical maps whatever the batch split (every voxel's events are applied in ray order)."""
    shards = [synth.rays_c4_shard(r, n=1_000_000) for r in range(4)]
    rays = np.concatenate(shards)
    layers = ("occupancy", "mean")
    one, four = OccupancyMap(0.1, layers=layers), OccupancyMap(0.1, layers=layers)
    g1 = GpuMap(one, gpu_mem_size=8 << 30)
    g4 = GpuMap(four, gpu_mem_size=8 << 30)
    assert g1.integrateRays(rays) == rays.shape[0]
    visits = g1.stats()["voxel_visits"]
    total4 = 0
    for s in shards:
        assert g4.integrateRays(s) == s.shape[0]
        total4 += g4.stats()["voxel_visits"]
    # (the closed form of expected_visits() does not apply: these sensor origins sit exactly on voxel boundaries, where
    # ohm's region / local key arithmetic and a plain floor(p / res) round differently)
    assert visits == total4 and visits > 10 ** 9
    g1.syncVoxels()
    g4.syncVoxels()
    assert set(one.chunks) == set(four.chunks) and len(one.chunks) > 3000
    for key, c in one.chunks.items():
        for name in layers:
            assert np.array_equal(c[name].view(np.uint32), four.chunks[key][name].view(np.uint32)), (key, name)


def test_c4_shard_origin_on_voxel_boundaries_matches_oracle(gpu):
    """The C4 sensor origins (-60, -20, ...) sit exactly on voxel and region boundaries: ohm's key arithmetic
    (region = floor(p / R + 0.5), local = floor((p - region_min) / res) with its 1e-6 edge fix-ups, ohm/MapCoord.h:45-93)
    decides which voxel such a point belongs to, and the device has to decide the same way."""
    for shard in (0, 5):
        rays = synth.rays_c4_shard(shard, n=30000)
        map_ = OccupancyMap(0.1, layers=("occupancy", "mean"))
        gm = GpuMap(map_)
        assert gm.integrateRays(rays) == rays.shape[0]
        gm.syncVoxels()
        om = make_oracle(map_)
        om.integrate_occupancy(rays)
        assert om.visit_count() == gm.stats()["voxel_visits"]
        assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean"], exact_float=True))
