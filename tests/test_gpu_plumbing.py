"""-m gpu: device plumbing of the C ABI (the gputil replacement), after tests/gputiltest/GpuBufferTest.cpp:22-590
(ReadWriteCopy, Pinned, Allocation) and GpuDevice.Enumerate."""
import ctypes as C

import numpy as np
import pytest

from ohm_amd import _lib as L

pytestmark = pytest.mark.gpu


def test_device_enumerate(gpu):
    assert "gfx950" in gpu["arch"]
    assert gpu["compute_units"] == 256
    assert gpu["lds_bytes_per_block"] >= 160 * 1024


def test_buffer_read_write_and_events(gpu):
    n = 1 << 20
    src = ((np.arange(n, dtype=np.uint64) * 2654435761) & 0xFFFFFFFF).astype(np.uint32)
    buf = L._vp()
    L.check(L.lib.ohmhip_buffer_create(C.byref(buf), src.nbytes, 3))
    stream, ev0, ev1 = L._vp(), L._vp(), L._vp()
    L.check(L.lib.ohmhip_stream_create(C.byref(stream)))
    L.check(L.lib.ohmhip_event_create(C.byref(ev0)))
    L.check(L.lib.ohmhip_event_create(C.byref(ev1)))
    L.check(L.lib.ohmhip_event_record(ev0, stream))
    # async write with completion event, then async read blocked on it
    L.check(L.lib.ohmhip_buffer_write(buf, src.ctypes.data, src.nbytes, 0, stream, None, ev1))
    dst = np.zeros_like(src)
    L.check(L.lib.ohmhip_buffer_read(buf, dst.ctypes.data, dst.nbytes, 0, stream, ev1, None))
    L.check(L.lib.ohmhip_stream_finish(stream))
    assert np.array_equal(src, dst)
    done = C.c_int(0)
    L.check(L.lib.ohmhip_event_is_complete(ev1, C.byref(done)))
    assert done.value == 1
    ms = C.c_float(-1)
    L.check(L.lib.ohmhip_event_elapsed_ms(ev0, ev1, C.byref(ms)))
    assert ms.value >= 0
    # partial write at an offset + fill
    L.check(L.lib.ohmhip_buffer_fill(buf, 0, 4096, 0, None))
    patch = np.full(16, 0xDEADBEEF, dtype=np.uint32)
    L.check(L.lib.ohmhip_buffer_write(buf, patch.ctypes.data, patch.nbytes, 64, None, None, None))
    L.check(L.lib.ohmhip_buffer_read(buf, dst.ctypes.data, 4096, 0, None, None, None))
    assert np.all(dst[:16] == 0) and np.all(dst[16:32] == 0xDEADBEEF) and np.all(dst[32:1024] == 0)
    # grow-only resize
    actual = C.c_size_t(0)
    L.check(L.lib.ohmhip_buffer_resize(buf, 1024, C.byref(actual)))
    assert actual.value == src.nbytes
    L.check(L.lib.ohmhip_buffer_resize(buf, 2 * src.nbytes, C.byref(actual)))
    assert actual.value == 2 * src.nbytes
    # out-of-range access is refused, not clamped
    assert L.lib.ohmhip_buffer_read(buf, dst.ctypes.data, 16, 2 * src.nbytes, None, None, None) == L.ERR_INVALID_ARG
    for h, fn in ((ev0, L.lib.ohmhip_event_destroy), (ev1, L.lib.ohmhip_event_destroy),
                  (stream, L.lib.ohmhip_stream_destroy), (buf, L.lib.ohmhip_buffer_destroy)):
        L.check(fn(h))


def test_pinned_host_memory(gpu):
    ptr = L._vp()
    L.check(L.lib.ohmhip_host_alloc(C.byref(ptr), 1 << 16))
    arr = np.ctypeslib.as_array((C.c_uint8 * (1 << 16)).from_address(ptr.value))
    arr[:] = 7
    buf = L._vp()
    L.check(L.lib.ohmhip_buffer_create(C.byref(buf), 1 << 16, 3))
    L.check(L.lib.ohmhip_buffer_write(buf, ptr, 1 << 16, 0, None, None, None))
    out = np.zeros(1 << 16, dtype=np.uint8)
    L.check(L.lib.ohmhip_buffer_read(buf, out.ctypes.data, 1 << 16, 0, None, None, None))
    assert np.all(out == 7)
    L.check(L.lib.ohmhip_buffer_destroy(buf))
    L.check(L.lib.ohmhip_host_free(ptr))


def test_sync_voxels_layer_subset_keeps_regions_marked(gpu):
    """GpuMap::syncVoxels(layer_indices) (ohmgpu/GpuMap.cpp:327-345): only the listed layers come back; the others must
    still arrive with the next full syncVoxels()."""
    import numpy as np
    from ohm_amd import GpuMap, OccupancyMap, synth
    from oracle.oracle import OracleMap
    layers = ("occupancy", "mean")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    rays = synth.rays_c0(n=3000, length=3.0)
    gm.integrateRays(rays)
    gm.syncVoxels(layer_names=["occupancy"])
    assert map_.chunks and all("occupancy" in c and "mean" not in c for c in map_.chunks.values())
    om = OracleMap(0.1, layers=layers)
    om.integrate_occupancy(rays)
    expect = om.chunks()
    for key, c in map_.chunks.items():
        assert np.array_equal(c["occupancy"].view(np.uint32), expect[key]["occupancy"].view(np.uint32))
    gm.syncVoxels()
    for key, c in map_.chunks.items():
        assert np.array_equal(c["mean"].view(np.uint32), expect[key]["mean"].view(np.uint32))
    assert len(gm.regionKeys(dirty_only=True)) == 0


def test_remove_regions_restarts_them_and_keeps_the_rest(gpu):
    """MapRegionCache::remove (OccupancyMap::cullRegions -> gpu_cache->remove, ohm/OccupancyMap.cpp:1202-1234): removed
    regions leave the device map and start from scratch when rays reach them again; the others are untouched although
    their slots may have moved.  Occupancy and NDT (whose per-voxel replay mask has to move with the slots)."""
    import numpy as np
    from ohm_amd import GpuMap, GpuNdtMap, OccupancyMap, synth
    from parity import assert_parity, compare_maps, make_oracle
    a = synth.rays_c1(n=20000, max_range=12.0, seed=21)
    b = synth.rays_c1(n=20000, max_range=12.0, seed=22, first=7000)
    for cls, layers, res in ((GpuMap, ("occupancy", "mean"), 0.1), (GpuNdtMap, ("occupancy",), 0.2)):
        map_ = OccupancyMap(res, (32, 32, 32), layers=layers)
        gm = cls(map_)
        ndt = cls is GpuNdtMap

        def oracle():
            om = make_oracle(map_)
            if ndt:
                om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold,
                           adaptation_rate=gm.adaptation_rate, reinit_threshold=gm.reinitialise_covariance_threshold,
                           reinit_count=gm.reinitialise_covariance_point_count, ndt_tm=False)
            return om

        def integrate(om, rays):
            om.integrate_ndt(rays) if ndt else om.integrate_occupancy(rays)

        gm.integrateRays(a)
        gm.syncVoxels()
        keys = gm.regionKeys()
        victims = keys[(keys[:, 0] + keys[:, 1] + keys[:, 2]) % 2 == 0]  # every other region, scattered over the slots
        assert 0 < len(victims) < len(keys)
        assert gm.removeRegions(np.concatenate([victims, [[30000, 0, 0]]])) == len(victims)  # unknown key: ignored
        left = {tuple(k) for k in gm.regionKeys().tolist()}
        assert left == {tuple(k) for k in keys.tolist()} - {tuple(k) for k in victims.tolist()}
        for k in victims:
            map_.chunks.pop(tuple(int(v) for v in k), None)  # the host side of cullRegions
        gm.integrateRays(b)
        gm.syncVoxels()
        both, only_b = oracle(), oracle()
        integrate(both, a)
        integrate(both, b)
        integrate(only_b, b)
        victim_set = {tuple(int(v) for v in k) for k in victims}
        expect = {k: (only_b.chunks()[k] if k in victim_set else v) for k, v in both.chunks().items()
                  if k not in victim_set or k in only_b.chunks()}
        assert_parity(compare_maps(expect, map_.chunks, list(map_.layers), rel=1e-5, exact_float=not ndt))
