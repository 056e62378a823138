"""-m gpu: GpuMap::setRayFilter with arbitrary host filters (ohmgpu/GpuMap.cpp:348-369, 736-746) -- the reference's
GpuMap.ClipBox test (tests/ohmtestgpu/GpuMapTest.cpp:633-760) and parity with the CPU oracle fed the same filtered rays:
occupancy / mean / TSDF bit exact, NDT within 1e-5."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, GpuTsdfMap, OccupancyMap, RayFlag, synth
from ohm_amd import rayfilter as RF

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def _filtered(filt, rays):
    keep, starts, ends, flags = filt(rays[0::2].copy(), rays[1::2].copy())
    out = np.empty((2 * int(keep.sum()), 3))
    out[0::2] = starts[keep]
    out[1::2] = ends[keep]
    return out, flags[keep], keep


def test_clip_box_like_the_reference(gpu):
    resolution = 0.2
    map_ = OccupancyMap(resolution, (32, 32, 32))
    gm = GpuMap(map_, expected_element_count=4096)
    box = RF.Aabb((-1.0, -1.0, -1.0), (2.0, 2.0, 2.0))
    gm.setRayFilter(RF.clip_bounded(box))
    assert gm.rayFilter() is not None and gm.effectiveRayFilter() is gm.rayFilter()
    through = np.array([(-2, 0, 0), (3, 0, 0), (0, -2, 0), (0, 3, 0), (0, 0, 3), (0, 0, -2)], dtype=np.float64)
    assert gm.integrateRays(through) == through.shape[0]
    gm.syncVoxels()
    touched = 0
    for key, layers in map_.chunks.items():
        occ = layers["occupancy"].reshape(32, 32, 32)
        seen = np.isfinite(occ)
        touched += int(seen.sum())
        assert np.all(occ[seen] < map_.occupancy_threshold_value)  # only free and unknown
        zs, ys, xs = np.nonzero(seen)
        centre = (np.stack([xs, ys, zs], axis=1) + 0.5) * resolution + (np.array(key) * 32 - 16) * resolution
        assert np.all(centre + 0.5 * resolution >= box.min - 1e-9) and np.all(centre - 0.5 * resolution <= box.max + 1e-9)
    assert touched > 0
    # "Reset the map.  This also tests that resetting a GPU map works."
    map_.chunks.clear()
    gm.clear()
    into = np.array([(-2, 0, 0), (0, 0, 0), (0, -2, 0), (0, 0, 0), (0, 0, 3), (0, 0, 0)], dtype=np.float64)
    gm.integrateRays(into)
    gm.syncVoxels()
    occupied = sum(int(np.count_nonzero(np.isfinite(l["occupancy"]) & (l["occupancy"] > map_.occupancy_threshold_value)))
                   for l in map_.chunks.values())
    assert occupied == 1  # the voxel holding the shared sample; everything else is free or unknown
    gm.clearRayFilter()
    assert gm.rayFilter() is None


@pytest.mark.parametrize("which", ["clip_bounded", "clip_to_bounds", "clip_ray"])
def test_host_filters_match_oracle_occupancy(gpu, which):
    box = RF.Aabb((-3.0, -2.5, -1.0), (3.5, 2.0, 1.5))
    filt = {"clip_bounded": RF.clip_bounded(box), "clip_to_bounds": RF.clip_to_bounds(box),
            "clip_ray": RF.clip_ray_filter(4.0)}[which]
    layers = ("occupancy", "mean")
    rays = np.concatenate([synth.random_rays(9000, extent=6.0, seed=31, origin_spread=3.0),
                           synth.rays_c1(n=9000, max_range=9.0, seed=32)])
    rays[7] = np.nan  # clip_ray_filter rejects it; the box filters have no validity test of their own
    map_ = OccupancyMap(0.1, layers=layers)
    if which != "clip_ray":
        rays = rays[np.repeat(np.all(np.isfinite(rays.reshape(-1, 6)), axis=1), 2)]
    gm = GpuMap(map_)
    gm.setRayFilter(filt)
    om = make_oracle(map_)
    for flags in (0, int(RayFlag.kRfExcludeOrigin)):
        for i in range(0, rays.shape[0], 2 * 7000):
            chunk = rays[i:i + 2 * 7000]
            kept, fflags, keep = _filtered(filt, chunk)
            assert gm.integrateRays(chunk, ray_update_flags=flags) == kept.shape[0]
            om.integrate_occupancy(kept, flags=flags, filter_flags=fflags)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))


def test_host_filter_ndt_and_tsdf(gpu):
    box = RF.Aabb((-12.0, -12.0, -2.0), (15.0, 12.0, 3.0))
    filt = RF.clip_bounded(box)
    rays = synth.rays_c2(n=25000)
    # NDT
    map_n = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
    gn = GpuNdtMap(map_n)
    gn.setRayFilter(filt)
    on = make_oracle(map_n)
    on.set_ndt(sensor_noise=gn.sensor_noise, sample_threshold=gn.sample_threshold, adaptation_rate=gn.adaptation_rate,
               reinit_threshold=gn.reinitialise_covariance_threshold,
               reinit_count=gn.reinitialise_covariance_point_count, ndt_tm=False)
    # TSDF
    map_t = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
    gt = GpuTsdfMap(map_t, default_truncation_distance=0.2)
    gt.setRayFilter(filt)
    ot = make_oracle(map_t)
    opts = gt.tsdf_options
    ot.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    for i in range(0, rays.shape[0], 2 * 10000):
        chunk = rays[i:i + 2 * 10000]
        kept, fflags, _ = _filtered(filt, chunk)
        assert gn.integrateRays(chunk) == kept.shape[0]
        assert gt.integrateRays(chunk) == kept.shape[0]
        on.integrate_ndt(kept, filter_flags=fflags)
        ot.integrate_tsdf(kept, filter_flags=fflags)
    gn.syncVoxels()
    gt.syncVoxels()
    assert_parity(compare_maps(on.chunks(), map_n.chunks, list(map_n.layers), rel=1e-5))
    assert_parity(compare_maps(ot.chunks(), map_t.chunks, ["tsdf"], exact_float=True))
