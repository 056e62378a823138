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
