"""-m gpu: spill to host (include/ohmhip.h "SPILL TO HOST") -- a map bounded to a fraction of the regions a sensor
track covers keeps integrating: cold regions move to the library's host store, come back when rays reach them again,
and the result equals the CPU oracle's (the reference bounds its GPU cache and reuses the least recently used slot,
ohmgpu/GpuLayerCache.cpp:530-584)."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, OccupancyMap, _lib as L, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def sensor_rays(origin, n, seed, min_range=1.5, max_range=4.0):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    length = rng.uniform(min_range, max_range, n)
    rays = np.empty((2 * n, 3), dtype=np.float64)
    rays[0::2] = np.asarray(origin, dtype=np.float64) + 0.013
    rays[1::2] = rays[0::2] + d * length[:, None]
    return rays


def track(n_stops, spacing=9.0):
    """A sensor moving out along x and back again: the return leg reaches regions that were evicted on the way out."""
    out = [(spacing * i, 0.3 * i, 0.0) for i in range(n_stops)]
    return out + out[-2::-1]


def limited_map(map_, regions, cls=GpuMap, **kwargs):
    gm = cls(map_, region_capacity=64, **kwargs)
    per_region = gm.cacheStats()["bytes_per_region"]
    gm.setMemoryLimit(regions * per_region)
    return gm


@pytest.mark.parametrize("layers,writeback", [(("occupancy",), False), (("occupancy", "mean"), False),
                                              (("occupancy", "mean"), True)])
def test_track_under_a_memory_limit_equals_the_unbounded_result(gpu, layers, writeback):
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=layers)
    gm = limited_map(map_, 100)
    gm.setSpillToHost(True)
    gm.setSpillWriteback(writeback)  # (background copies of the next victims: same results, fewer copy-outs on the path)
    om = make_oracle(map_)
    for k, origin in enumerate(track(6)):
        rays = sensor_rays(origin, 6000, seed=700 + k)
        assert gm.integrateRays(rays) == rays.shape[0]
        om.integrate_occupancy(rays)
    st = gm.cacheStats()
    assert st["evictions"] > 0 and st["readmissions"] > 0 and st["regions_spilled"] > 0 and st["spill_enabled"] == 1
    assert st["regions_resident"] <= 100
    assert (st["writebacks"] > 0 and st["writeback_hits"] > 0) if writeback else st["writebacks"] == 0
    n_regions = len(om.chunks())
    assert len(gm.regionKeys()) == n_regions == st["regions_resident"] + st["regions_spilled"]
    assert len(gm.regionKeys(dirty_only=True)) == n_regions  # nothing synced yet: stored regions count as well
    gm.syncVoxels()
    assert len(gm.regionKeys(dirty_only=True)) == 0
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))
    # a second leg after the sync: only what it touches is dirty again, and the map still matches
    rays = sensor_rays((0.0, 0.0, 0.0), 6000, seed=999)
    assert gm.integrateRays(rays) == rays.shape[0]
    om.integrate_occupancy(rays)
    assert 0 < len(gm.regionKeys(dirty_only=True)) < n_regions
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))


def test_ndt_regions_keep_their_replay_mask_across_a_spill(gpu):
    map_ = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
    gm = limited_map(map_, 70, cls=GpuNdtMap)
    gm.setSpillToHost(True)
    om = make_oracle(map_)
    om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold,
               adaptation_rate=gm.adaptation_rate, reinit_threshold=gm.reinitialise_covariance_threshold,
               reinit_count=gm.reinitialise_covariance_point_count, ndt_tm=False)
    for k, origin in enumerate(track(4, spacing=16.0)):
        rays = sensor_rays(origin, 5000, seed=800 + k, min_range=2.0, max_range=7.0)
        # cluster the samples so voxels collect several of them (NDT state beyond the first sample)
        rays[1::2] = np.round(rays[1::2] / 0.5) * 0.5 + 0.017 * np.sin(np.arange(5000))[:, None]
        assert gm.integrateRays(rays) == rays.shape[0]
        om.integrate_ndt(rays)
    st = gm.cacheStats()
    assert st["evictions"] > 0 and st["readmissions"] > 0
    gm.syncVoxels()
    stats = compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean", "covariance"], rel=1e-5)
    assert_parity(stats)


@pytest.mark.parametrize("writeback", [False, True])
def test_tsdf_regions_across_a_spill(gpu, writeback):
    from ohm_amd import GpuTsdfMap
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",))
    gm = limited_map(map_, 100, cls=GpuTsdfMap, default_truncation_distance=0.3)
    gm.setSpillToHost(True)
    gm.setSpillWriteback(writeback)  # (TSDF: the replay mask row travels with the pre-cleaned copy)
    om = make_oracle(map_)
    opts = gm.tsdf_options
    om.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    for k, origin in enumerate(track(5)):
        rays = sensor_rays(origin, 4000, seed=850 + k)
        assert gm.integrateRays(rays) == rays.shape[0]
        om.integrate_tsdf(rays)
    st = gm.cacheStats()
    assert st["evictions"] > 0 and st["readmissions"] > 0
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["tsdf"], exact_float=True))


@pytest.mark.parametrize("writeback", [False, True])
def test_c3_sweep_under_the_reference_cache_budget(gpu, writeback):
    """SURVEY 8d's cache-stress variant of C3 at test size: the lidar sweep presented as 45-degree sectors to a TSDF map
    whose pool holds a third of the regions a revolution touches; a quarter of a second revolution brings the first
    sectors back from the host store.  Bit exact against the oracle -- also with the background write-back on, whose
    copies race the batches by design (a copy of a region that is touched afterwards must be discarded)."""
    from ohm_amd import GpuTsdfMap
    map_ = OccupancyMap(0.05, (32, 32, 32), layers=("tsdf",))
    gm = limited_map(map_, 1500, cls=GpuTsdfMap)
    gm.setSpillToHost(True)
    gm.setSpillWriteback(writeback)
    om = make_oracle(map_)
    opts = gm.tsdf_options
    om.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    rays = synth.rays_c3(n=1_250_000)
    per = 125_000
    for k in range(10):
        part = rays[2 * k * per:2 * (k + 1) * per]
        assert gm.integrateRays(part) == part.shape[0]
    om.integrate_tsdf(rays)
    st = gm.cacheStats()
    assert st["evictions"] > 1000 and st["readmissions"] > 100 and st["regions_resident"] <= 1500
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["tsdf"], exact_float=True))


def test_stored_regions_answer_the_region_calls(gpu):
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    gm = limited_map(map_, 80)
    gm.setSpillToHost(True)
    # the GpuCache view reports the budget and the layers (ohmgpu/GpuCache.h:139-143)
    assert gm.gpuCache().targetGpuAllocSize() == 80 * gm.cacheStats()["bytes_per_region"]
    assert gm.gpuCache().layerCount() == 1
    om = make_oracle(map_)
    for k, origin in enumerate([(0.0, 0.0, 0.0), (12.0, 0.0, 0.0), (24.0, 0.0, 0.0)]):
        rays = sensor_rays(origin, 5000, seed=900 + k)
        gm.integrateRays(rays)
        om.integrate_occupancy(rays)
    st = gm.cacheStats()
    assert st["regions_spilled"] > 0
    resident = {tuple(int(v) for v in k) for k in gm.regionKeys()}
    assert resident == set(om.chunks().keys())
    # the first stop's own region is cold by now: stored, so it has no slot ...
    import ctypes as C
    first_key = np.array([[0, 0, 0]], dtype=np.int16)
    slot = C.c_uint32(0)
    assert L.lib.ohmhip_map_region_slot(gm._handle, first_key.ctypes.data, C.byref(slot)) == L.ERR_NOT_FOUND
    # ... an upload of the host copy brings it back as an ordinary resident region ...
    gm.syncVoxels()
    gm.uploadRegions(first_key)
    assert L.lib.ohmhip_map_region_slot(gm._handle, first_key.ctypes.data, C.byref(slot)) == L.OK
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))
    # ... removing a stored region forgets it, clear() forgets everything
    stored = [k for k in gm.regionKeys() if L.lib.ohmhip_map_region_slot(
        gm._handle, np.ascontiguousarray(k).ctypes.data, C.byref(slot)) == L.ERR_NOT_FOUND]
    assert stored
    before = len(gm.regionKeys())
    assert gm.removeRegions([stored[0]]) == 1
    assert len(gm.regionKeys()) == before - 1
    # replica merge does not combine with spilling
    assert L.lib.ohmhip_map_enable_merge(gm._handle) == L.ERR_UNSUPPORTED
    gm.clear()
    assert len(gm.regionKeys()) == 0 and gm.cacheStats()["regions_spilled"] == 0


def test_a_batch_larger_than_the_limit_is_integrated_in_pieces(gpu):
    """A batch that alone touches more regions than the limit leaves room for: without spilling it fails and changes
    nothing; with spilling it is integrated as halves in ray order (the reference finalises what it has enqueued when its
    cache fills mid-batch and carries on, ohmgpu/GpuMap.cpp:900-996) and the map is the CPU mapper's for the whole batch."""
    small = sensor_rays((0.0, 0.0, 0.0), 2000, seed=11, max_range=2.5)
    big = sensor_rays((0.0, 0.0, 0.0), 4000, seed=12, min_range=5.0, max_range=9.0)  # > 20 regions on its own
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    gm = limited_map(map_, 20)
    gm.setBatchCoalescing(0)                   # (a collected batch would report its failure at the flush, not here)
    om = make_oracle(map_)
    assert gm.integrateRays(small) == small.shape[0]
    om.integrate_occupancy(small)
    assert gm.integrateRays(big) == 0          # no spilling: clean failure
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))
    gm.setSpillToHost(True)
    assert gm.integrateRays(big) == big.shape[0]
    om.integrate_occupancy(big)
    assert gm.integrateRays(small) == small.shape[0]
    om.integrate_occupancy(small)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))
    assert gm.cacheStats()["evictions"] > 0
    # a traversal layer carries its exit range within a call: such a map keeps failing cleanly
    map_t = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy", "traversal"))
    gt = limited_map(map_t, 20)
    gt.setSpillToHost(True)
    assert gt.integrateRays(small) == small.shape[0]
    assert gt.integrateRays(big) == 0


def test_regions_created_by_name_obey_the_limit(gpu):
    """uploads (write_regions) and ensure_regions used to grow the pool past the memory limit.  Now they make
    room first -- the least recently used OTHER regions go to the store -- or fail and change nothing without spilling."""
    import ctypes as C
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    gm = limited_map(map_, 40)
    gm.setSpillToHost(True)
    om = make_oracle(map_)
    rays = sensor_rays((0.0, 0.0, 0.0), 6000, seed=21, max_range=3.5)
    assert gm.integrateRays(rays) == rays.shape[0]
    om.integrate_occupancy(rays)
    gm.syncVoxels()
    before = gm.cacheStats()
    assert 20 < before["regions_resident"] <= 40
    # upload 30 regions far away (new keys): the pool would need 30 more slots than the limit allows
    far = np.array([[40 + i, 0, 0] for i in range(30)], dtype=np.int16)
    block = np.full(32 ** 3, np.float32(0.25), dtype=np.float32)
    for k in far:
        map_.chunks[tuple(int(v) for v in k)] = {"occupancy": block.copy()}
    assert gm.uploadRegions(far) == 30
    st = gm.cacheStats()
    assert st["regions_resident"] <= 40 and st["regions_spilled"] >= before["regions_resident"] + 30 - 40
    assert st["evictions"] >= st["regions_spilled"]
    # everything is still there and correct: the rays' regions (some from the store) and the uploaded blocks
    keys = {tuple(int(v) for v in k) for k in gm.regionKeys()}
    assert keys == set(om.chunks().keys()) | {tuple(int(v) for v in k) for k in far}
    ks = np.array(sorted(keys), dtype=np.int16)
    out = np.zeros((len(ks), 32 ** 3), dtype=np.float32)
    ptrs = (C.c_void_p * len(ks))(*[out[i].ctypes.data for i in range(len(ks))])
    L.check(L.lib.ohmhip_map_read_regions(gm._handle, L.LID_OCCUPANCY, ks.ctypes.data, len(ks), ptrs))
    got = {tuple(int(v) for v in k): out[i] for i, k in enumerate(ks)}
    for k, layers in om.chunks().items():
        assert np.array_equal(got[k].view(np.uint32), layers["occupancy"].reshape(-1).view(np.uint32)), k
    for k in far:
        assert np.all(got[tuple(int(v) for v in k)] == np.float32(0.25))
    # without spilling the same request fails cleanly
    map2 = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    gm2 = limited_map(map2, 40)
    assert gm2.integrateRays(rays) == rays.shape[0]
    gm2.syncVoxels()
    n_before = len(gm2.regionKeys())
    slots = np.zeros(30, dtype=np.uint32)
    assert L.lib.ohmhip_map_ensure_regions(gm2._handle, far.ctypes.data, 30, slots.ctypes.data) == L.ERR_CAPACITY
    assert len(gm2.regionKeys()) == n_before and gm2.cacheStats()["regions_resident"] == n_before


def test_a_rejected_batch_leaves_no_modified_flags(gpu):
    """a batch that fails with OHMHIP_ERR_CAPACITY must leave the map exactly as it was -- also the
    'modified since the last sync' flags of the resident regions it would have touched."""
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    gm = limited_map(map_, 30)  # no spilling: the batch below cannot run
    gm.setBatchCoalescing(0)    # every call launches (and answers for) its own device batch
    small = sensor_rays((0.0, 0.0, 0.0), 3000, seed=31, max_range=2.5)
    assert gm.integrateRays(small) == small.shape[0]
    gm.syncVoxels()
    assert len(gm.regionKeys(dirty_only=True)) == 0
    n_regions = len(gm.regionKeys())
    big = sensor_rays((0.0, 0.0, 0.0), 5000, seed=32, min_range=5.0, max_range=9.0)  # crosses the resident regions too
    assert gm.integrateRays(big) == 0
    assert len(gm.regionKeys(dirty_only=True)) == 0, "the rejected batch marked regions as modified"
    assert len(gm.regionKeys()) == n_regions
    # and a batch that fits afterwards is tracked normally
    assert gm.integrateRays(small) == small.shape[0]
    assert 0 < len(gm.regionKeys(dirty_only=True)) <= n_regions


def test_repeated_sweep_keeps_part_of_the_map_resident(gpu):
    """A sensor sweeping a map larger than the pool again and again is the access pattern plain LRU is worst at: it would
    evict exactly what the next calls need, every region leaving and returning once per revolution.  The eviction ranks
    regions by their predicted next use (a region's own return period, or the median of recent re-admissions), so from
    the second revolution on part of the map stays resident: fewer re-admissions than regions x revolutions -- and the
    map is the oracle's, bit exact, whatever the policy evicts."""
    sectors, revolutions, per_sector = 8, 5, 5000
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy", "mean"))
    om = make_oracle(map_)
    rng = np.random.default_rng(2026)

    def sector_rays(j, seed):
        r = np.random.default_rng(seed)
        az = (j + r.uniform(0.0, 1.0, per_sector)) * (2.0 * np.pi / sectors)
        el = r.uniform(-0.2, 0.2, per_sector)
        length = r.uniform(9.0, 14.0, per_sector)
        d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
        rays = np.empty((2 * per_sector, 3), dtype=np.float64)
        rays[0::2] = np.array([0.013, 0.021, 0.017])
        rays[1::2] = rays[0::2] + d * length[:, None]
        return rays

    # how many regions the whole sweep and its largest sector touch (oracle only)
    probe = make_oracle(OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",)))
    per_sector_regions = []
    for j in range(sectors):
        before = len(probe.chunks())
        probe.integrate_occupancy(sector_rays(j, 100 + j))
        per_sector_regions.append(len(probe.chunks()) - before)
    total = len(probe.chunks())
    limit = int(0.6 * total)
    assert limit > 2 * max(per_sector_regions)
    gm = limited_map(map_, limit)
    gm.setSpillToHost(True)
    gm.setBatchCoalescing(0)
    for rev in range(revolutions):
        for j in range(sectors):
            rays = sector_rays(j, 100 + j + 1000 * rev + int(rng.integers(0, 1)))
            assert gm.integrateRays(rays) == rays.shape[0]
            om.integrate_occupancy(rays)
    st = gm.cacheStats()
    assert st["evictions"] > 0 and st["regions_resident"] <= limit
    # plain LRU on this cyclic pattern re-admits (nearly) every region once per revolution after the first
    lru_readmissions = (revolutions - 1) * total
    assert st["readmissions"] < 0.8 * lru_readmissions, (st["readmissions"], lru_readmissions, total, limit)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean"], exact_float=True))
