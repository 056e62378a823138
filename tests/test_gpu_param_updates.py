"""-m gpu: parameters changed on the host map / mapper AFTER the GpuMap exists apply from the next batch, as in the
reference (its GpuMap reads the OccupancyMap's values at every launch, ohmgpu/GpuMap.cpp:1036-1191; GpuNdtMap::
setSensorNoise, GpuTsdfMap option setters).  Oracle maps get the same changes between the same calls."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, GpuTsdfMap, OccupancyMap, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def test_probabilities_clamps_and_filter_changed_between_batches(gpu):
    layers = ("occupancy", "mean")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    gm.setBatchCoalescing(1 << 20)  # a pending batch keeps the values it was presented under
    rays = synth.rays_c1(n=24000, max_range=10.0, seed=3)
    oracles = []
    om = make_oracle(map_)
    om.integrate_occupancy(rays[:16000])
    gm.integrateRays(rays[:16000])
    map_.setHitProbability(0.7)
    map_.setMissProbability(0.3)
    map_.min_voxel_value, map_.max_voxel_value = -1.0, 1.5
    map_.ray_filter = ("clip", 6.0)
    gm.integrateRays(rays[16000:32000])
    om2 = make_oracle(map_)  # same parameters as the map has NOW, continuing on the first oracle's voxels
    # the oracle has no "continue with new parameters" entry point: replay both halves into one map via its setters
    from oracle.oracle import lib as _olib
    _olib.oracle_map_set_hit_value(om.handle, float(map_.hit_value))
    _olib.oracle_map_set_miss_value(om.handle, float(map_.miss_value))
    _olib.oracle_map_set_min_max(om.handle, float(map_.min_voxel_value), float(map_.max_voxel_value))
    om.set_ray_filter("clip", 6.0)
    om.integrate_occupancy(rays[16000:32000])
    map_.saturate_at_max_value = True
    _olib.oracle_map_set_saturation(om.handle, 0, 1)
    gm.integrateRays(rays[32000:])
    om.integrate_occupancy(rays[32000:])
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))
    del om2, oracles


def test_ndt_sensor_noise_and_tsdf_options_changed_between_batches(gpu):
    rays = synth.rays_c2(n=30000)
    map_n = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
    gn = GpuNdtMap(map_n)
    on = make_oracle(map_n)

    def ndt_params():
        on.set_ndt(sensor_noise=gn.sensor_noise, sample_threshold=gn.sample_threshold,
                   adaptation_rate=gn.adaptation_rate, reinit_threshold=gn.reinitialise_covariance_threshold,
                   reinit_count=gn.reinitialise_covariance_point_count, ndt_tm=False)

    ndt_params()
    gn.integrateRays(rays[:30000])
    on.integrate_ndt(rays[:30000])
    gn.setSensorNoise(0.11)
    gn.sample_threshold = 5
    assert gn.sensorNoise() == pytest.approx(0.11)
    ndt_params()
    gn.integrateRays(rays[30000:])
    on.integrate_ndt(rays[30000:])
    gn.syncVoxels()
    assert_parity(compare_maps(on.chunks(), map_n.chunks, list(map_n.layers), rel=1e-5))

    map_t = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
    gt = GpuTsdfMap(map_t, default_truncation_distance=0.2)
    ot = make_oracle(map_t)
    ot.set_tsdf(max_weight=gt.tsdf_options[0], trunc=gt.tsdf_options[1], dropoff=gt.tsdf_options[2],
                sparsity=gt.tsdf_options[3])
    gt.integrateRays(rays[:20000])
    ot.integrate_tsdf(rays[:20000])
    gt.tsdf_options = (3.0, 0.2, 0.0, 1.0)  # a weight cap below what many voxels already hold
    ot.set_tsdf(max_weight=3.0, trunc=0.2, dropoff=0.0, sparsity=1.0)
    gt.integrateRays(rays[20000:40000])
    ot.integrate_tsdf(rays[20000:40000])
    # a new truncation distance on a populated map is refused, not approximated; the map keeps working as it was
    gt.setDefaultTruncationDistance(0.3)  # (the reference's setter names; same effect as assigning tsdf_options)
    assert gt.tsdf_options == (3.0, 0.3, 0.0, 1.0) and gt.maxWeight() == 3.0 and gt.dropoffEpsilon() == 0.0
    with pytest.raises(Exception):
        gt.integrateRays(rays[40000:])
    gt.tsdf_options = (3.0, 0.2, 0.0, 1.0)
    gt.integrateRays(rays[40000:])
    ot.integrate_tsdf(rays[40000:])
    gt.syncVoxels()
    assert_parity(compare_maps(ot.chunks(), map_t.chunks, ["tsdf"], exact_float=True))


def test_gpumap_value_pass_throughs_and_grouped_rays(gpu):
    """GpuMap::setHitValue / setMissValue pass through to the map (ohmgpu/GpuMap.h:234-244) and apply from the next batch;
    setGroupedRays is accepted and changes nothing (every batch is binned per region on the device)."""
    from oracle.oracle import lib as _olib
    map_ = OccupancyMap(0.1, layers=("occupancy",))
    gm = GpuMap(map_)
    om = make_oracle(map_)
    rays = synth.rays_c1(n=16000, max_range=9.0, seed=8)
    gm.integrateRays(rays[:16000])
    om.integrate_occupancy(rays[:16000])
    gm.setHitValue(1.25)
    gm.setMissValue(-0.75)
    assert gm.hitValue() == map_.hitValue() == 1.25 and gm.missValue() == -0.75
    assert gm.groupedRays() is False
    gm.setGroupedRays(True)
    assert gm.groupedRays() is True
    _olib.oracle_map_set_hit_value(om.handle, 1.25)
    _olib.oracle_map_set_miss_value(om.handle, -0.75)
    gm.integrateRays(rays[16000:])
    om.integrate_occupancy(rays[16000:])
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))
