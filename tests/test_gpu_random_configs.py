"""-m gpu: randomised map geometries against the CPU oracle -- region shapes that are not 32^3 (odd voxel counts, long
thin regions, tiny regions), off-grid origins, mixed ray flags and multi-batch integration.  Same bar as the other
occupancy tests: identical region sets, integer fields and log-odds bit exact."""
import numpy as np
import pytest

from ohm_amd import GpuMap, OccupancyMap, RayFlag, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu

CASES = [
    # resolution, region dims, origin, extent, rays, batches, flags, layers
    (0.1, (5, 7, 9), (0.013, -0.4, 0.27), 4.0, 6000, 3, 0, ("occupancy",)),
    (0.2, (64, 16, 8), (0.0, 0.0, 0.0), 12.0, 8000, 2, 0, ("occupancy", "mean")),
    (0.05, (3, 3, 3), (0.02, 0.02, 0.02), 1.5, 4000, 2, int(RayFlag.kRfEndPointAsFree), ("occupancy",)),
    (0.15, (31, 33, 32), (-1.0, 2.0, 0.5), 9.0, 10000, 4, int(RayFlag.kRfExcludeOrigin), ("occupancy", "mean")),
    (0.1, (32, 32, 32), (0.05, 0.05, 0.05), 8.0, 20000, 5, 0, ("occupancy", "mean")),
    (0.3, (128, 16, 16), (0.0, 0.1, 0.0), 40.0, 8000, 1, 0, ("occupancy",)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_geometry_parity(gpu, case):
    res, dims, origin, extent, n_rays, batches, flags, layers = CASES[case]
    rays = synth.random_rays(n_rays, extent=extent, seed=100 + case, origin_spread=0.3 * extent)
    map_ = OccupancyMap(res, dims, layers=layers)
    map_.setOrigin(origin)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    step = 2 * ((n_rays + batches - 1) // batches)
    total = 0
    for i in range(0, rays.shape[0], step):
        total += gm.integrateRays(rays[i:i + step], ray_update_flags=flags)
        om.integrate_occupancy(rays[i:i + step], flags=flags)
    gm.syncVoxels()
    assert total == rays.shape[0]
    stats = compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True)
    assert_parity(stats)
    assert gm.stats()["voxel_visits"] > 0


NDT_TSDF_CASES = [
    # kind, resolution, region dims, origin, extent, rays, batches
    ("ndt", 0.2, (5, 7, 9), (0.013, -0.4, 0.27), 6.0, 6000, 2),
    ("ndt", 0.25, (31, 33, 32), (-1.0, 2.0, 0.5), 14.0, 9000, 3),
    ("tsdf", 0.1, (6, 10, 4), (0.02, 0.02, 0.02), 4.0, 5000, 2),
    ("tsdf", 0.15, (64, 16, 8), (0.0, 0.0, 0.0), 10.0, 8000, 2),
]


@pytest.mark.parametrize("case", range(len(NDT_TSDF_CASES)))
def test_random_geometry_ndt_tsdf(gpu, case):
    """Region shapes other than 32^3 through the NDT and TSDF pipelines (event sort keys, LDS tile sizes and the
    replay kernels all depend on the region volume)."""
    from ohm_amd import GpuNdtMap, GpuTsdfMap
    kind, res, dims, origin, extent, n_rays, batches = NDT_TSDF_CASES[case]
    rays = synth.random_rays(n_rays, extent=extent, seed=500 + case, origin_spread=0.2 * extent)
    # cluster the samples so NDT voxels collect several samples each
    rays[1::2] = np.round(rays[1::2] / (2.5 * res)) * (2.5 * res) + 0.013 * np.sin(np.arange(n_rays))[:, None]
    layers = ("occupancy",) if kind == "ndt" else ("tsdf",)
    map_ = OccupancyMap(res, dims, layers=layers)
    map_.setOrigin(origin)
    om = None
    if kind == "ndt":
        gm = GpuNdtMap(map_)
        om = make_oracle(map_)
        om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold,
                   adaptation_rate=gm.adaptation_rate, reinit_threshold=gm.reinitialise_covariance_threshold,
                   reinit_count=gm.reinitialise_covariance_point_count, ndt_tm=False)
    else:
        gm = GpuTsdfMap(map_, default_truncation_distance=2.0 * res)
        om = make_oracle(map_)
        opts = gm.tsdf_options
        om.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    step = 2 * ((n_rays + batches - 1) // batches)
    for i in range(0, rays.shape[0], step):
        assert gm.integrateRays(rays[i:i + step]) == rays[i:i + step].shape[0]
        if kind == "ndt":
            om.integrate_ndt(rays[i:i + step])
        else:
            om.integrate_tsdf(rays[i:i + step])
    gm.syncVoxels()
    if kind == "ndt":
        assert_parity(compare_maps(om.chunks(), map_.chunks, list(map_.layers), rel=1e-5))
    else:
        assert_parity(compare_maps(om.chunks(), map_.chunks, ["tsdf"], exact_float=True))


@pytest.mark.parametrize("case", [0, 2, 3, 5])
def test_random_geometry_traversal(gpu, case):
    """The traversal pass (k_region_traversal: a 32-bit LDS tile sized by the region volume) on the same geometries:
    odd voxel counts, tiny and long regions, off-grid origins, ray flags.  Occupancy bit exact, traversal 1e-5."""
    res, dims, origin, extent, n_rays, batches, flags, _ = CASES[case]
    rays = synth.random_rays(n_rays, extent=extent, seed=900 + case, origin_spread=0.3 * extent)
    map_ = OccupancyMap(res, dims, layers=("occupancy", "traversal"))
    map_.setOrigin(origin)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    step = 2 * ((n_rays + batches - 1) // batches)
    for i in range(0, rays.shape[0], step):
        assert gm.integrateRays(rays[i:i + step], ray_update_flags=flags) == rays[i:i + step].shape[0]
        om.integrate_occupancy(rays[i:i + step], flags=flags)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))
    worst = 0.0
    for key, cpu in om.chunks().items():
        g, c = map_.chunks[key]["traversal"], cpu["traversal"]
        assert np.array_equal(c != 0, g != 0)
        nz = c != 0
        if nz.any():
            worst = max(worst, float(np.max(np.abs(g[nz] - c[nz]) / np.maximum(np.abs(c[nz]), 1e-3))))
    assert worst < 1e-5, worst
