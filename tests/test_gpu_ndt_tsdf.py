"""-m gpu: HIP GpuNdtMap / GpuTsdfMap vs the CPU oracle (RayMapperNdt / RayMapperTsdf restatements) on identical
rays.  Integer fields (mean coord/count, hit/miss counts) must be bit exact; float fields within 1e-5 relative
(north_star bar; NDT uses exp/log whose device and host libm may differ in the last bit)."""
import numpy as np
import pytest

import ohm_amd
from ohm_amd import GpuNdtMap, GpuTsdfMap, NdtMode, OccupancyMap, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def run_ndt(rays, resolution=0.2, batch=None, mode=NdtMode.kOccupancy, intensities=None, flags=0):
    map_ = OccupancyMap(resolution, (32, 32, 32), layers=("occupancy",))
    gm = GpuNdtMap(map_, ndt_mode=mode)
    om = make_oracle(map_)
    om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold,
               adaptation_rate=gm.adaptation_rate, reinit_threshold=gm.reinitialise_covariance_threshold,
               reinit_count=gm.reinitialise_covariance_point_count, ndt_tm=(mode == NdtMode.kTraversability))
    n_points = rays.shape[0]
    step = n_points if batch is None else 2 * batch
    for i in range(0, n_points, step):
        chunk = rays[i:i + step]
        ints = None if intensities is None else intensities[i // 2:(i + step) // 2]
        assert gm.integrateRays(chunk, intensities=ints, ray_update_flags=flags) == chunk.shape[0]
        om.integrate_ndt(chunk, intensities=ints, flags=int(flags))
    gm.syncVoxels()
    stats = compare_maps(om.chunks(), map_.chunks, list(map_.layers), rel=1e-5)
    return stats, gm, om


def test_ndt_room_single_batch(gpu):
    rays = synth.rays_c2(n=40000)
    stats, gm, om = run_ndt(rays)
    assert_parity(stats)
    assert gm.stats()["voxel_visits"] == om.visit_count()


def test_ndt_room_batched_many_samples_per_voxel(gpu):
    # several revolutions over the same walls in batches: covariances mature (count >> threshold), NDT misses active
    rays = np.concatenate([synth.rays_c2(n=20000, seed=100 + k) for k in range(4)])
    stats, gm, om = run_ndt(rays, batch=10000)
    assert_parity(stats)


def test_ndt_random_rays_small_voxels(gpu):
    rays = synth.random_rays(6000, extent=4.0, seed=21, origin_spread=1.0)
    stats, gm, om = run_ndt(rays, resolution=0.5, batch=2048)
    assert_parity(stats)


def test_ndt_unaddressable_points(gpu):
    # ohm/RayMapperNdt.cpp:277-286: no voxel is walked when either key is null, the sample is still applied (to the end
    # voxel when it is addressable, to Key::kNull's voxel otherwise) -- same rule as the occupancy mapper.
    good = synth.random_rays(300, extent=6.0, seed=41)
    res = 0.2
    edge = 32768 * 32 * res  # first coordinate beyond the int16 region range
    bad = np.array([[0, 0, 0], [2 * edge, 0, 0], [-3 * edge, 1, 1], [1, 1, 1], [edge - 3.0, 0, 0], [edge + 1.0, 0, 0],
                    [-(edge - 3.0), 0, 0], [-(edge - 0.3), 0, 0]], dtype=np.float64)
    rays = np.concatenate([good[:300], bad, good[300:]])
    stats, gm, om = run_ndt(rays, resolution=res)
    assert_parity(stats)


def test_ndt_ray_flags_and_clip_filter(gpu):
    # RayMapperNdt honours kRfEndPointAsFree / kRfExcludeOrigin / kRfExcludeRay (ohm/RayMapperNdt.cpp:238-262); the
    # clip filter shortens long rays and turns their end voxel into a walked voxel.
    rays = np.concatenate([synth.rays_c2(n=8000, seed=300 + k) for k in range(2)])
    for flags in (ohm_amd.RayFlag.kRfEndPointAsFree, ohm_amd.RayFlag.kRfExcludeOrigin, ohm_amd.RayFlag.kRfExcludeRay,
                  ohm_amd.RayFlag.kRfEndPointAsFree | ohm_amd.RayFlag.kRfExcludeOrigin):
        stats, gm, om = run_ndt(rays, batch=4000, flags=flags)
        assert_parity(stats)
    map_ = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
    map_.ray_filter = ("clip", 12.0)
    gm = GpuNdtMap(map_)
    om = make_oracle(map_)
    om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold, adaptation_rate=gm.adaptation_rate,
               reinit_threshold=gm.reinitialise_covariance_threshold,
               reinit_count=gm.reinitialise_covariance_point_count)
    for i in range(0, rays.shape[0], 8000):
        gm.integrateRays(rays[i:i + 8000])
        om.integrate_ndt(rays[i:i + 8000])
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(map_.layers), rel=1e-5))


def test_ndt_tm(gpu):
    rays = np.concatenate([synth.rays_c2(n=15000, seed=200 + k) for k in range(3)])
    ints = (synth.uniform01(5, np.arange(rays.shape[0] // 2, dtype=np.uint64), 0) * 100).astype(np.float32)
    stats, gm, om = run_ndt(rays, batch=15000, mode=NdtMode.kTraversability, intensities=ints)
    assert_parity(stats)


def run_tsdf(rays, resolution=0.1, batch=None, trunc=0.1, **opts):
    map_ = OccupancyMap(resolution, (32, 32, 32), layers=("tsdf",))
    gm = GpuTsdfMap(map_, default_truncation_distance=trunc, **opts)
    om = make_oracle(map_)
    om.set_tsdf(max_weight=gm.tsdf_options[0], trunc=gm.tsdf_options[1], dropoff=gm.tsdf_options[2],
                sparsity=gm.tsdf_options[3])
    n_points = rays.shape[0]
    step = n_points if batch is None else 2 * batch
    for i in range(0, n_points, step):
        chunk = rays[i:i + step]
        assert gm.integrateRays(chunk) == chunk.shape[0]
        om.integrate_tsdf(chunk)
    gm.syncVoxels()
    stats = compare_maps(om.chunks(), map_.chunks, ["tsdf"], exact_float=True)
    return stats, gm, om


def test_tsdf_room(gpu):
    rays = synth.rays_c2(n=30000)
    stats, gm, om = run_tsdf(rays, resolution=0.1)
    assert_parity(stats)
    assert gm.stats()["voxel_visits"] == om.visit_count()


def test_tsdf_batched_small_voxels(gpu):
    rays = np.concatenate([synth.rays_c2(n=8000, seed=300 + k) for k in range(3)])
    stats, gm, om = run_tsdf(rays, resolution=0.05, batch=8000)
    assert_parity(stats)


def test_tsdf_reference_test_rays(gpu):
    # tests/ohmtest/TsdfTests.cpp:18-137 ray set, large and small truncation distance, integrated twice
    dirs = [(1, 0, 0), (-1, 0, 0), (1, 1, 0), (-1, 1, 0), (1, 0, 0), (1, -1, 0), (-1, 0, 1), (1, 1, 1), (-1, 1, 1),
            (1, 0, 1), (1, -1, 1), (-1, 0, 1), (1, 1, -1), (-1, 1, -1), (1, 0, -1), (1, -1, -1)]
    rays = np.zeros((2 * len(dirs), 3))
    rays[1::2] = dirs
    for trunc in (10.0, 0.1):
        stats, gm, om = run_tsdf(np.concatenate([rays, rays]), trunc=trunc, batch=len(dirs))
        assert_parity(stats)


def test_tsdf_sparsity_and_weight_cap(gpu):
    rays = np.concatenate([synth.rays_c2(n=4000, seed=400)] * 6)
    stats, gm, om = run_tsdf(rays, batch=4000, max_weight=4.0, sparsity_compensation_factor=2.5)
    assert_parity(stats)


def test_tsdf_weight_dropoff(gpu):
    # ohm/VoxelTsdfCompute.h:103-107: with dropoff_epsilon > 0 a free-space visit adds a weight that depends on the
    # voxel's sdf, so nothing can be counted: every visit of the batch goes through the ordered replay.  Bit exact.
    rays = np.concatenate([synth.rays_c2(n=3000, seed=450 + k) for k in range(3)])
    stats, gm, om = run_tsdf(rays, batch=3000, dropoff_epsilon=0.05)
    assert_parity(stats)
    stats, gm, om = run_tsdf(rays[:6000], resolution=0.2, trunc=0.5, batch=1500, dropoff_epsilon=0.2,
                             sparsity_compensation_factor=1.5)
    assert_parity(stats)


def test_ndt_and_tsdf_soak_mixed_batch_sizes(gpu):
    """Many calls of very different sizes (1 ray ... 40 000 rays): the event-list sizing follows the previous batch's
    demand, the sort scratch grows and shrinks, single-ray batches take the same kernels as large ones."""
    rng = np.random.default_rng(77)
    sizes = [1, 3, 50, 3000, 40000, 700, 12000]
    calls = []
    first = 0
    for call in range(24):
        n = sizes[int(rng.integers(len(sizes)))]
        calls.append(synth.rays_c2(n=n, seed=900 + call, first=first))
        first += n
    # NDT
    map_n = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
    gn = GpuNdtMap(map_n)
    on = make_oracle(map_n)
    on.set_ndt(sensor_noise=gn.sensor_noise, sample_threshold=gn.sample_threshold, adaptation_rate=gn.adaptation_rate,
               reinit_threshold=gn.reinitialise_covariance_threshold,
               reinit_count=gn.reinitialise_covariance_point_count, ndt_tm=False)
    # TSDF
    map_t = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
    gt = GpuTsdfMap(map_t, default_truncation_distance=0.2)
    ot = make_oracle(map_t)
    opts = gt.tsdf_options
    ot.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    for rays in calls:
        assert gn.integrateRays(rays) == rays.shape[0]
        assert gt.integrateRays(rays) == rays.shape[0]
        on.integrate_ndt(rays)
        ot.integrate_tsdf(rays)
    gn.syncVoxels()
    gt.syncVoxels()
    assert_parity(compare_maps(on.chunks(), map_n.chunks, list(map_n.layers), rel=1e-5))
    assert_parity(compare_maps(ot.chunks(), map_t.chunks, ["tsdf"], exact_float=True))


def test_ndt_and_tsdf_survive_pool_growth(gpu):
    """A small region pool that grows while the map is being built.  The per-voxel "ordered replay" mask is persistent
    state for NDT / TSDF and has to move with the regions when the pool is re-allocated: a wall is sampled densely, the
    pool is then grown by rays elsewhere, and finally rays are shot THROUGH the wall voxels (NDT misses against their
    Gaussians, TSDF free-space updates of near-surface voxels)."""
    origin = np.array([0.05, 0.05, 0.05])
    g = np.arange(-2.0, 2.0, 0.07)
    yy, zz = np.meshgrid(g, g, indexing="ij")

    def rays_to(points):
        out = np.empty((2 * len(points), 3))
        out[0::2] = origin
        out[1::2] = points
        return out

    wall = np.stack([np.full(yy.size, 5.0), yy.ravel(), zz.ravel()], axis=1)
    phase1 = [rays_to(wall + np.array([0.013 * k, 0.011 * k, -0.007 * k])) for k in range(3)]
    elsewhere = [rays_to(np.stack([np.full(yy.size, x), 3.0 * yy.ravel(), 3.0 * zz.ravel()], axis=1)) for x in (-9.0, -14.0)]
    through = [rays_to(origin + 1.8 * (wall - origin))]
    calls = phase1 + elsewhere + through + phase1[:1]

    map_n = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
    gn = GpuNdtMap(map_n, region_capacity=4)
    on = make_oracle(map_n)
    on.set_ndt(sensor_noise=gn.sensor_noise, sample_threshold=gn.sample_threshold, adaptation_rate=gn.adaptation_rate,
               reinit_threshold=gn.reinitialise_covariance_threshold,
               reinit_count=gn.reinitialise_covariance_point_count, ndt_tm=False)
    map_t = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
    gt = GpuTsdfMap(map_t, default_truncation_distance=0.2, region_capacity=4)
    ot = make_oracle(map_t)
    opts = gt.tsdf_options
    ot.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    resident = []
    for chunk in calls:
        assert gn.integrateRays(chunk) == chunk.shape[0]
        assert gt.integrateRays(chunk) == chunk.shape[0]
        on.integrate_ndt(chunk)
        ot.integrate_tsdf(chunk)
        resident.append((gn.stats()["regions_resident"], gt.stats()["regions_resident"]))
    assert resident[2][0] < resident[4][0] and resident[2][1] < resident[4][1], resident  # the pools did grow
    assert resident[4][0] > 4 and resident[4][1] > 4
    gn.syncVoxels()
    gt.syncVoxels()
    assert_parity(compare_maps(on.chunks(), map_n.chunks, list(map_n.layers), rel=1e-5))
    assert_parity(compare_maps(ot.chunks(), map_t.chunks, ["tsdf"], exact_float=True))


def test_event_list_overflow_rewalk_applies_nothing_twice(gpu, monkeypatch):
    # OHMHIP_EVENT_LIMIT (read at map creation) caps the first sizing of the NDT / TSDF event list, so every batch
    # overflows it and the walk is repeated with a list of the right size.  Single-chunk regions are applied straight
    # from LDS by the first launch: the repeat must neither apply their counts again nor leave them behind.
    monkeypatch.setenv("OHMHIP_EVENT_LIMIT", "64")
    rays = np.concatenate([synth.rays_c2(n=12000, seed=500 + k) for k in range(3)])
    stats, gm, om = run_ndt(rays, batch=12000)
    assert_parity(stats)
    stats, gm, om = run_tsdf(rays[:24000], batch=6000)
    assert_parity(stats)


@pytest.mark.parametrize("kind", ["ndt", "tsdf"])
def test_event_sort_speculation_miss_repeats_with_the_exact_size(gpu, kind):
    """Round 6: from the second batch on, the event sort and the replay are launched on the PREVIOUS batch's event count
    plus head room (k_pad_events; the kernels that consume the sorted list do nothing when the true count exceeded the
    speculation, and the host repeats sort and replay with the exact size).  Batches growing tenfold force the miss, batches
    shrinking tenfold the padding; results must be those of the oracle either way (NDT 1e-5, TSDF bit exact)."""
    sizes = [1500, 40000, 3000, 60000, 60000, 500]
    rays = [synth.rays_c2(n=n, seed=300 + k) for k, n in enumerate(sizes)]
    if kind == "ndt":
        map_ = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
        gm = GpuNdtMap(map_)
        om = make_oracle(map_)
        om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold,
                   adaptation_rate=gm.adaptation_rate, reinit_threshold=gm.reinitialise_covariance_threshold,
                   reinit_count=gm.reinitialise_covariance_point_count)
    else:
        map_ = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
        gm = GpuTsdfMap(map_)
        om = make_oracle(map_)
        om.set_tsdf(max_weight=gm.tsdf_options[0], trunc=gm.tsdf_options[1], dropoff=gm.tsdf_options[2],
                    sparsity=gm.tsdf_options[3])
    gm.setBatchCoalescing(0)
    for chunk in rays:
        assert gm.integrateRays(chunk) == chunk.shape[0]
        if kind == "ndt":
            om.integrate_ndt(chunk)
        else:
            om.integrate_tsdf(chunk)
    gm.syncVoxels()
    stats = compare_maps(om.chunks(), map_.chunks, list(map_.layers), rel=1e-5, exact_float=(kind == "tsdf"))
    assert_parity(stats)
