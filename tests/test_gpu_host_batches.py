"""-m gpu: the host-pointer entry point (ohmhip_map_integrate_rays) -- double-buffered staging (batch N+1 is copied
and uploaded while batch N runs; the caller's array may be reused the moment the call returns) and batch coalescing
(ohmhip_map_set_batch_coalescing: consecutive small batches run as one device batch).  Everything is compared with
the CPU oracle integrating the same calls one by one: bit exact."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, OccupancyMap, RayFlag, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu


def test_back_to_back_host_batches_reuse_the_callers_buffer(gpu):
    layers = ("occupancy", "mean")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    scratch = np.empty((2 * 30000, 3), dtype=np.float64)
    for k in range(7):
        n = 30000 - 3500 * k  # shrinking and growing batches through both staging slots
        rays = synth.rays_c1(n=n, max_range=12.0, seed=40 + k, first=1000 * k)
        scratch[:2 * n] = rays
        assert gm.integrateRays(scratch[:2 * n]) == 2 * n
        scratch[:] = np.nan  # the library must have taken its copy already
        om.integrate_occupancy(rays)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))


@pytest.mark.parametrize("min_rays", [1, 5000, 1 << 20])
def test_coalesced_small_batches_equal_separate_calls(gpu, min_rays):
    layers = ("occupancy", "mean", "touch_time")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    gm.setBatchCoalescing(min_rays)
    om = make_oracle(map_)
    rays = synth.rays_c1(n=24000, max_range=10.0, seed=9)
    ts = 50.0 + 0.002 * np.arange(rays.shape[0] // 2, dtype=np.float64)
    small = 2 * 1000
    for i in range(0, rays.shape[0], small):
        chunk, tchunk = rays[i:i + small], ts[i // 2:(i + small) // 2]
        # the flags change half way: batches on either side must not be merged
        flags = int(RayFlag.kRfEndPointAsFree) if i >= rays.shape[0] // 2 else 0
        assert gm.integrateRays(chunk, timestamps=tchunk, ray_update_flags=flags) == chunk.shape[0]
        om.integrate_occupancy(chunk, timestamps=tchunk, flags=flags)
    gm.syncVoxels()  # observing the map runs what is still pending
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))
    if min_rays == 1 << 20:
        # two device batches in total: one per flag value
        assert gm.stats()["rays_in"] == 12000


def test_coalescing_keeps_order_with_device_batches_and_optional_arrays(gpu):
    import ctypes as C
    from ohm_amd import _lib as L
    map_ = OccupancyMap(0.1, layers=("occupancy",))
    gm = GpuMap(map_)
    gm.setBatchCoalescing(1 << 20)
    om = make_oracle(map_)
    a = synth.rays_c0(n=3000, length=3.0, seed=1)
    b = synth.rays_c0(n=3000, length=3.0, seed=2)
    c = synth.rays_c0(n=3000, length=3.0, seed=3)
    ts = np.linspace(1.0, 2.0, 3000)
    gm.integrateRays(a)                  # deferred
    buf = L._vp()
    L.check(L.lib.ohmhip_buffer_create(C.byref(buf), b.nbytes, 3))
    L.check(L.lib.ohmhip_buffer_write(buf, b.ctypes.data, b.nbytes, 0, None, None, None))
    ptr = L._vp()
    L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(ptr)))
    gm.integrateRaysDevice(ptr, b.shape[0])  # must run after `a`
    gm.integrateRays(c)                  # deferred
    gm.integrateRays(a, timestamps=ts)   # different optional arrays: `c` is launched first
    for r in (a, b, c, a):
        om.integrate_occupancy(r)
    gm.syncVoxels()
    L.lib.ohmhip_buffer_destroy(buf)
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))


def test_traversal_maps_never_merge_batches(gpu):
    map_ = OccupancyMap(0.1, layers=("occupancy", "traversal"))
    gm = GpuMap(map_)
    gm.setBatchCoalescing(1 << 20)
    rays = synth.rays_c0(n=2000, length=3.0, seed=4)
    gm.integrateRays(rays[:2000])
    gm.integrateRays(rays[2000:])
    gm.wait()
    assert gm.stats()["rays_in"] == 1000  # the second call's own batch


def test_ndt_coalesced(gpu):
    rays = synth.rays_c2(n=20000)
    map_ = OccupancyMap(0.2, (32, 32, 32), layers=("occupancy",))
    gm = GpuNdtMap(map_)
    gm.setBatchCoalescing(8192)
    om = make_oracle(map_)
    om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold, adaptation_rate=gm.adaptation_rate,
               reinit_threshold=gm.reinitialise_covariance_threshold,
               reinit_count=gm.reinitialise_covariance_point_count, ndt_tm=False)
    for i in range(0, rays.shape[0], 2 * 2048):
        gm.integrateRays(rays[i:i + 2 * 2048])
    om.integrate_ndt(rays)  # one CPU call == the sequence of calls (ray by ray either way)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(map_.layers), rel=1e-5))


def test_soak_mixed_batch_sizes_flags_and_coalescing(gpu):
    """Sixty calls of wildly different sizes (1 ray ... 60 000 rays), changing flags and a coalescing threshold that is
    switched on and off along the way: exercises the alternating batch-summary blocks, the speculative binning (which
    keys on the previous batch), both staging slots and pool growth, against the CPU oracle fed the same calls."""
    layers = ("occupancy", "mean")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_, region_capacity=64)  # small pool: grows several times
    om = make_oracle(map_)
    rng = np.random.default_rng(2026)
    sizes = [1, 7, 100, 5000, 60000, 300, 2, 20000]
    flag_choices = [0, int(RayFlag.kRfEndPointAsFree), int(RayFlag.kRfExcludeOrigin), int(RayFlag.kRfExcludeSample)]
    first = 0
    for call in range(60):
        n = sizes[int(rng.integers(len(sizes)))]
        flags = flag_choices[int(rng.integers(len(flag_choices)))]
        if call % 9 == 0:
            gm.setBatchCoalescing([0, 4096, 1 << 20][int(rng.integers(3))])
        if call % 2:
            rays = synth.rays_c1(n=n, max_range=9.0, seed=7000 + call, first=first)
        else:
            rays = synth.random_rays(n, extent=6.0, seed=8000 + call, origin_spread=2.0)
        first += n
        assert gm.integrateRays(rays, ray_update_flags=flags) == rays.shape[0]
        om.integrate_occupancy(rays, flags=flags)
        if call % 13 == 5:
            assert gm.stats()["rays_in"] > 0  # observing the map mid-stream flushes what is pending
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))


def test_deferred_calls_report_their_own_filter_count(gpu):
    """Small host batches are collected by default; every call still returns the number of points ITS rays contribute
    (the ray filter's verdict, evaluated on the host with the device's arithmetic)."""
    rays = synth.random_rays(3000, extent=6.0, seed=61)
    rays[2 * 10 + 1] = np.nan              # rejected: not finite
    rays[2 * 2500 + 1, 0] = np.inf         # rejected
    rays[2 * 1500 + 1] = rays[2 * 1500] + np.array([3e10, 0, 0])  # rejected: longer than the default 1e10 filter
    map_ = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy", "mean"))
    gm = GpuMap(map_)
    om = make_oracle(map_)
    total = 0
    for first in range(0, 3000, 1000):
        part = rays[2 * first:2 * (first + 1000)]
        got = gm.integrateRays(part)
        assert got == part.shape[0] - 2  # one bad ray per call
        total += got
        om.integrate_occupancy(part)
    assert gm.stats()["rays_integrated"] * 2 == total  # the three calls ran as one device batch
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean"], exact_float=True))


def test_small_device_pointer_batches_are_collected_and_report_their_own_counts(gpu):
    """Device-pointer calls of 4096 rays (the f4 pipeline: GpuTransformSamples output presented as the reference tools
    present host rays) are collected on the device like small host batches: copied behind the rays already waiting and run
    as one device batch per 65 536 rays.  Every call still returns ITS count (one-workgroup filter pass over its staged
    rays), the caller may overwrite its array once the call has returned a count, host and device calls keep their order,
    and the result is the CPU oracle's for the same sequence of calls."""
    import ctypes as C
    from ohm_amd import _lib as L
    layers = ("occupancy", "mean")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    rays = synth.rays_c1(n=40 * 4096, max_range=11.0, seed=77)
    bad = [5, 4096 + 17, 9 * 4096 + 4000]
    for b in bad:
        rays[2 * b + 1] = np.nan  # the default filter rejects these: one per affected call
    scratch = L._vp()
    L.check(L.lib.ohmhip_buffer_create(C.byref(scratch), 2 * 4096 * 24, 3))
    ptr = L._vp()
    L.check(L.lib.ohmhip_buffer_ptr(scratch, C.byref(ptr)))
    device_batches = 0
    last_seen = None
    for call in range(40):
        part = np.ascontiguousarray(rays[2 * 4096 * call:2 * 4096 * (call + 1)])
        if call == 20:
            # a host call in the middle: what waits on the device runs first
            expect = part.shape[0] - 2 * sum(1 for b in bad if b // 4096 == call)
            assert gm.integrateRays(part) == expect
        else:
            # the SAME device array is overwritten for every call: the library has taken its copy when the call returns
            L.check(L.lib.ohmhip_buffer_write(scratch, part.ctypes.data, part.nbytes, 0, None, None, None))
            expect = part.shape[0] - 2 * sum(1 for b in bad if b // 4096 == call)
            assert gm.integrateRaysDevice(ptr, part.shape[0]) == expect, call
        om.integrate_occupancy(part)
    st = gm.stats()  # observing the map flushes what is pending
    assert st["rays_in"] in (16 * 4096, 3 * 4096), st["rays_in"]  # 20 calls -> 16 + 4 (flushed by the host call); 19 -> 16 + 3
    gm.syncVoxels()
    L.lib.ohmhip_buffer_destroy(scratch)
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))
    # one device batch per call when coalescing is off
    gm2 = GpuMap(OccupancyMap(0.1, layers=("occupancy",)))
    gm2.setBatchCoalescing(0)
    buf = L._vp()
    L.check(L.lib.ohmhip_buffer_create(C.byref(buf), 2 * 4096 * 24, 3))
    L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, 2 * 4096 * 24, 0, None, None, None))
    p2 = L._vp()
    L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(p2)))
    assert gm2.integrateRaysDevice(p2, 2 * 4096) == 2 * 4096 - 2
    assert gm2.stats()["rays_in"] == 4096
    L.lib.ohmhip_buffer_destroy(buf)


@pytest.mark.parametrize("n", [4 * 32768, 4 * 32768 + 1, 200_001, 9 * 32768 - 1])
def test_large_host_batches_are_uploaded_piece_by_piece(gpu, n):
    """A host call that is a device batch on its own is staged by the map's pool threads in 32 768-ray pieces, each
    piece's host-to-device copy queued as soon as it is staged (stageRaysAndUpload).  Ragged sizes around the piece
    boundaries, rejected rays in the first, a middle and the last piece, optional arrays, and the caller's buffer
    reused at once: bit exact against the oracle, counts reported by the call."""
    layers = ("occupancy", "mean", "touch_time")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    scratch = np.empty((2 * n, 3), dtype=np.float64)
    for k in range(3):
        rays = synth.rays_c1(n=n, max_range=9.0, seed=700 + k, first=17 * k)
        ts = 50.0 + k + 1e-6 * np.arange(n, dtype=np.float64)
        bad = [5, 32768 + 9, n - 1] if k != 1 else []
        for i in bad:
            rays[2 * i + 1, k % 3] = np.nan
        scratch[:] = rays
        assert gm.integrateRays(scratch, timestamps=ts) == 2 * (n - len(bad))
        scratch[:] = np.nan  # the library must have taken its copy already
        om.integrate_occupancy(rays, timestamps=ts)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))


def test_async_launch_gives_the_same_map_and_counts(gpu):
    """ohmhip_map_set_async_launch: large host calls return once staged, the launch sequence runs on the map's thread.
    Mixed with small (collected) calls, device-pointer-free optional arrays, rejected rays and observers in between:
    bit exact against the oracle, every call reports its own count."""
    layers = ("occupancy", "mean", "touch_time")
    map_ = OccupancyMap(0.1, layers=layers)
    gm = GpuMap(map_)
    gm.setAsyncLaunch(True)
    om = make_oracle(map_)
    sizes = [150_000, 140_000, 3000, 2000, 135_000, 70_000, 200_000, 2500, 1500]  # (ends with calls still collected)
    for k, n in enumerate(sizes):
        rays = synth.rays_c1(n=n, max_range=9.0, seed=900 + k, first=31 * k)
        ts = 10.0 * k + 1e-6 * np.arange(n, dtype=np.float64)
        bad = [1, n // 2, n - 2] if k % 2 == 0 else []
        for i in bad:
            rays[2 * i + 1, 1] = np.inf
        assert gm.integrateRays(rays, timestamps=ts) == 2 * (n - len(bad))
        rays[:] = np.nan  # the library must have taken its copy already
        if k == 4:
            assert len(gm.regionKeys()) > 0  # an observer in the middle settles the launch thread first
    rng_check = [(k, n) for k, n in enumerate(sizes)]
    for k, n in rng_check:
        rays = synth.rays_c1(n=n, max_range=9.0, seed=900 + k, first=31 * k)
        ts = 10.0 * k + 1e-6 * np.arange(n, dtype=np.float64)
        if k % 2 == 0:
            for i in (1, n // 2, n - 2):
                rays[2 * i + 1, 1] = np.inf
        om.integrate_occupancy(rays, timestamps=ts)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True))


def test_async_launch_reports_a_failed_batch_at_the_next_call(gpu):
    """The one change of contract: the batch's error (here: the region pool is full) comes back from the next call that
    settles the launch, not from the call that presented the rays; the map stays usable and unchanged by the batch."""
    from ohm_amd import _lib as L
    map_ = OccupancyMap(0.1, layers=("occupancy",))
    gm = GpuMap(map_, region_capacity=64)
    gm.setMemoryLimit(40 * gm.cacheStats()["bytes_per_region"])
    gm.setAsyncLaunch(True)
    small = synth.rays_c1(n=140_000, max_range=3.0, seed=5)   # a few regions around the sensor: fits
    assert gm.integrateRays(small) == small.shape[0]
    gm.wait()
    before = len(gm.regionKeys())
    assert 0 < before <= 40
    big = synth.rays_c1(n=140_000, max_range=25.0, seed=6)    # hundreds of regions: cannot fit 40
    assert gm.integrateRays(big) == big.shape[0]              # accepted: staged and handed to the launch thread
    with pytest.raises(L.OhmHipError) as info:
        gm.wait()
    assert info.value.status == L.ERR_CAPACITY
    assert len(gm.regionKeys()) == before                     # the failed batch left nothing behind
    assert gm.integrateRays(small) == small.shape[0]          # and the map goes on
    gm.wait()
    om = make_oracle(map_)
    om.integrate_occupancy(small)
    om.integrate_occupancy(small)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))


def test_async_launch_failure_keeps_the_following_calls_intact(gpu):
    """batch A fails on the launch thread; call B (a device batch of its own, staged AND uploaded piece by
    piece) is the one that learns of it and returns A's error with B's rays still queued; call C is appended to the
    same slot.  The flush that runs B + C must send the whole block again -- a stale 'uploaded' flag made it read
    n_B + n_C rays from a device copy holding only B's."""
    from ohm_amd import _lib as L
    map_ = OccupancyMap(0.1, layers=("occupancy",))
    gm = GpuMap(map_, region_capacity=64)
    gm.setMemoryLimit(40 * gm.cacheStats()["bytes_per_region"])
    gm.setAsyncLaunch(True)
    small = synth.rays_c1(n=140_000, max_range=3.0, seed=5)
    assert gm.integrateRays(small) == small.shape[0]
    gm.wait()
    big = synth.rays_c1(n=140_000, max_range=25.0, seed=6)     # A: cannot fit 40 regions
    assert gm.integrateRays(big) == big.shape[0]
    rays_b = synth.rays_c1(n=140_000, max_range=3.0, seed=7)   # B: learns of A's failure
    assert gm.integrateRays(rays_b) == 0 and gm._last_error == L.ERR_CAPACITY  # (0 on failure, like the reference)
    rays_c = synth.rays_c1(n=9_000, max_range=3.0, seed=8)     # C: small, appended behind B in the same slot
    assert gm.integrateRays(rays_c) == rays_c.shape[0]
    gm.wait()
    gm.syncVoxels()
    om = make_oracle(map_)
    for r in (small, rays_b, rays_c):                           # A left nothing; B ran with C
        om.integrate_occupancy(r)
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy"], exact_float=True))


def test_batch_timings_do_not_count_host_idle_time(gpu):
    """with phase timing off ms_total of a batch was replaced by the gap between the previous batch's end and
    this batch's end whenever a previous batch sat in the ring -- a batch presented after a host pause (one per sensor
    frame) then reported the pause.  The period is only used for batches that really overlapped."""
    import time
    map_ = OccupancyMap(0.1, layers=("occupancy",))
    gm = GpuMap(map_)
    gm.setBatchCoalescing(0)
    rays = synth.rays_c1(n=20000, max_range=10.0, seed=3)
    for _ in range(3):
        assert gm.integrateRays(rays) == rays.shape[0]
    gm.wait()
    time.sleep(0.25)
    assert gm.integrateRays(rays) == rays.shape[0]
    gm.wait()
    ms = gm.batchTimings(0)["ms_total"]
    assert 0.0 < ms < 50.0, ms  # (the pause was 250 ms; the batch itself takes a fraction of a millisecond)
