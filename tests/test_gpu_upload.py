"""-m gpu: a GpuMap created over a host map that already holds data (gpumap::enableGpu + GpuLayerCache::upload,
ohmgpu/GpuMap.cpp:106-122, GpuLayerCache.cpp:172-182): the CPU-built regions are uploaded, integration continues on the
device and the result must be what the CPU mapper gives for the whole ray sequence.  For NDT / TSDF the upload also has
to rebuild the per-voxel ordered-replay mask from the uploaded layers."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, GpuTsdfMap, OccupancyMap, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def _wall_scene():
    origin = np.array([0.05, 0.05, 0.05])
    g = np.arange(-2.0, 2.0, 0.07)
    yy, zz = np.meshgrid(g, g, indexing="ij")
    wall = np.stack([np.full(yy.size, 5.0), yy.ravel(), zz.ravel()], axis=1)

    def rays_to(points):
        out = np.empty((2 * len(points), 3))
        out[0::2] = origin
        out[1::2] = points
        return out

    first = np.concatenate([rays_to(wall + 0.011 * k) for k in range(3)])      # dense samples on the wall
    second = np.concatenate([rays_to(origin + 1.7 * (wall - origin)), rays_to(wall - 0.02)])  # through it, and onto it
    return first, second


def _adopt(map_, oracle_chunks):
    for key, layers in oracle_chunks.items():
        map_.chunks[key] = {name: np.array(block, copy=True) for name, block in layers.items() if name in map_.layers}


def test_occupancy_continues_from_a_cpu_built_map(gpu):
    layers = ("occupancy", "mean")
    a = synth.rays_c1(n=15000, max_range=10.0, seed=61)
    b = synth.rays_c1(n=15000, max_range=10.0, seed=62, first=4000)
    map_ = OccupancyMap(0.1, layers=layers)
    om = make_oracle(map_)
    om.integrate_occupancy(a)
    _adopt(map_, om.chunks())
    gm = GpuMap(map_)  # uploads what the host map holds
    assert gm.integrateRays(b) == b.shape[0]
    gm.syncVoxels()
    om.integrate_occupancy(b)
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))


def test_ndt_and_tsdf_continue_from_a_cpu_built_map(gpu):
    first, second = _wall_scene()
    # NDT: the host map must already have the NDT layers for its data to be adopted
    map_n = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy", "mean", "covariance"))
    probe = GpuNdtMap(OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",)))  # only to read the default parameters
    on = make_oracle(map_n)
    on.set_ndt(sensor_noise=probe.sensor_noise, sample_threshold=probe.sample_threshold,
               adaptation_rate=probe.adaptation_rate, reinit_threshold=probe.reinitialise_covariance_threshold,
               reinit_count=probe.reinitialise_covariance_point_count, ndt_tm=False)
    on.integrate_ndt(first)
    _adopt(map_n, on.chunks())
    gn = GpuNdtMap(map_n)
    assert gn.integrateRays(second) == second.shape[0]
    gn.syncVoxels()
    on.integrate_ndt(second)
    assert_parity(compare_maps(on.chunks(), map_n.chunks, list(map_n.layers), rel=1e-5))

    map_t = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
    ot = make_oracle(map_t)
    ot.set_tsdf(max_weight=1e4, trunc=0.2, dropoff=0.0, sparsity=1.0)
    ot.integrate_tsdf(first)
    _adopt(map_t, ot.chunks())
    gt = GpuTsdfMap(map_t, default_truncation_distance=0.2)
    assert gt.integrateRays(second) == second.shape[0]
    gt.syncVoxels()
    ot.integrate_tsdf(second)
    assert_parity(compare_maps(ot.chunks(), map_t.chunks, ["tsdf"], exact_float=True))


def test_cpu_side_integration_between_device_batches(gpu):
    """Mixed use: the device integrates A, the host map is synced and the CPU integrates X into it, the touched regions
    are pushed back (GpuLayerCache::upload of CPU-newer regions) and the device integrates B: CPU result of A + X + B."""
    layers = ("occupancy", "mean")
    a = synth.rays_c1(n=12000, max_range=9.0, seed=71)
    x = synth.random_rays(5000, extent=5.0, seed=72, origin_spread=2.0)
    b = synth.rays_c1(n=12000, max_range=9.0, seed=73, first=3000)
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    gm.integrateRays(a)
    gm.syncVoxels()
    om = make_oracle(map_)
    om.integrate_occupancy(a)
    before = {k: {n: v.copy() for n, v in c.items()} for k, c in om.chunks().items()}
    om.integrate_occupancy(x)  # the CPU mapper working on the host map
    after = om.chunks()
    edited = [k for k, c in after.items()
              if k not in before or any(not np.array_equal(c[n].view(np.uint32), before[k][n].view(np.uint32)) for n in layers)]
    assert edited
    for k in edited:
        map_.chunks[k] = {n: after[k][n].copy() for n in layers}
    assert gm.uploadRegions(edited) == len(edited)
    gm.integrateRays(b)
    gm.syncVoxels()
    om.integrate_occupancy(b)
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))
