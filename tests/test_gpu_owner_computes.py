"""-m gpu: exact multi-GPU integration by region ownership (include/ohmhip.h: ohmhip_map_set_region_ownership;
ohm_amd/distributed.py "owner computes").  `world` maps on the one test GPU stand in for `world` ranks: each is given
the SAME ray stream and keeps only its own regions; their union must be the map a single device (and the CPU oracle)
builds from that stream -- same bar as the single-map tests: occupancy / mean / TSDF bit exact, NDT within 1e-5."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, GpuTsdfMap, OccupancyMap, RayFlag, synth
from ohm_amd import distributed as D

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def _union_of_owned(maps, world, shift):
    """Merge the per-rank host maps, checking the partition: a region with data may only live on its owner."""
    union = {}
    for rank, map_ in enumerate(maps):
        keys = np.array(sorted(map_.chunks.keys()), dtype=np.int16).reshape(-1, 3)
        owners = D.region_owner(keys, world, shift) if len(keys) else np.zeros(0, np.uint32)
        for key, owner in zip(map(tuple, keys.tolist()), owners):
            assert owner == rank, f"rank {rank} holds region {key} owned by {owner}"
            assert key not in union
            union[key] = map_.chunks[key]
    return union


@pytest.mark.parametrize("world,shift,flags", [(2, 0, 0), (3, 1, 0), (4, 0, int(RayFlag.kRfEndPointAsFree)),
                                                (2, 2, int(RayFlag.kRfExcludeOrigin))])
def test_occupancy_owner_computes_is_exact(gpu, world, shift, flags):
    rays = np.concatenate([synth.rays_c1(n=20000, seed=5), synth.random_rays(6000, extent=12.0, seed=77,
                                                                            origin_spread=5.0)])
    layers = ("occupancy", "mean")
    maps, gms = [], []
    for rank in range(world):
        map_ = OccupancyMap(0.1, (32, 32, 32), layers=layers)
        gm = GpuMap(map_)
        gm.setRegionOwnership(world, rank, shift)
        gm.setBatchCoalescing(0)  # one device batch per call: the last batch's statistics are compared below
        maps.append(map_)
        gms.append(gm)
    om = make_oracle(maps[0])
    step = 2 * 9000
    for i in range(0, rays.shape[0], step):
        for gm in gms:
            assert gm.integrateRays(rays[i:i + step], ray_update_flags=flags) == rays[i:i + step].shape[0]
        om.integrate_occupancy(rays[i:i + step], flags=flags)
    for gm in gms:
        gm.syncVoxels()
    union = _union_of_owned(maps, world, shift)
    assert_parity(compare_maps(om.chunks(), union, list(layers), exact_float=True))
    assert min(len(m.chunks) for m in maps) > 0, "every rank should own part of this map"
    # the walk work is partitioned, not replicated: the ranks' segment counts of the last batch add up to one map's
    last = rays[(rays.shape[0] - 1) // step * step:]
    single = GpuMap(OccupancyMap(0.1, (32, 32, 32), layers=layers))
    single.integrateRays(last, ray_update_flags=flags)
    single.wait()
    assert sum(gm.stats()["ray_region_segments"] for gm in gms) == single.stats()["ray_region_segments"]


def test_ownership_rejected_once_regions_exist(gpu):
    gm = GpuMap(OccupancyMap(0.1))
    gm.integrateRays(synth.rays_c0(n=100, length=2.0))
    with pytest.raises(Exception):
        gm.setRegionOwnership(2, 0)


def test_ndt_owner_computes(gpu):
    rays = synth.rays_c2(n=40000)
    world, shift = 2, 0
    maps, gms = [], []
    for rank in range(world):
        map_ = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
        gm = GpuNdtMap(map_)
        gm.setRegionOwnership(world, rank, shift)
        maps.append(map_)
        gms.append(gm)
    om = make_oracle(maps[0])
    g = gms[0]
    om.set_ndt(sensor_noise=g.sensor_noise, sample_threshold=g.sample_threshold, adaptation_rate=g.adaptation_rate,
               reinit_threshold=g.reinitialise_covariance_threshold,
               reinit_count=g.reinitialise_covariance_point_count, ndt_tm=False)
    for i in range(0, rays.shape[0], 30000):
        for gm in gms:
            gm.integrateRays(rays[i:i + 30000])
        om.integrate_ndt(rays[i:i + 30000])
    for gm in gms:
        gm.syncVoxels()
    union = _union_of_owned(maps, world, shift)
    assert_parity(compare_maps(om.chunks(), union, list(maps[0].layers), rel=1e-5))


def test_tsdf_owner_computes(gpu):
    rays = synth.rays_c2(n=20000)
    world, shift = 3, 0
    maps, gms = [], []
    for rank in range(world):
        map_ = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
        gm = GpuTsdfMap(map_, default_truncation_distance=0.1)
        gm.setRegionOwnership(world, rank, shift)
        maps.append(map_)
        gms.append(gm)
    om = make_oracle(maps[0])
    opts = gms[0].tsdf_options
    om.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    for i in range(0, rays.shape[0], 16000):
        for gm in gms:
            gm.integrateRays(rays[i:i + 16000])
        om.integrate_tsdf(rays[i:i + 16000])
    for gm in gms:
        gm.syncVoxels()
    union = _union_of_owned(maps, world, shift)
    assert_parity(compare_maps(om.chunks(), union, ["tsdf"], exact_float=True))
