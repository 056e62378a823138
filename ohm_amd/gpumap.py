"""Host-side mirror (Python) of the reference interface for the ray-integration path.

Names, argument meaning and error behaviour follow the reference so the parity tests read like the reference's own
(tests/ohmtestgpu/GpuMapTest.cpp):

  ohm::OccupancyMap  (ohm/OccupancyMap.h:291)  -> OccupancyMap   : parameters + MapChunk/VoxelBlock-layout host blocks
  ohm::RayMapper     (ohm/RayMapper.h:22-65)   -> RayMapper
  ohm::GpuMap        (ohmgpu/GpuMap.h:143-384) -> GpuMap
  ohm::GpuNdtMap     (ohmgpu/GpuNdtMap.h:63)   -> GpuNdtMap
  ohm::GpuTsdfMap    (ohmgpu/GpuTsdfMap.h:37)  -> GpuTsdfMap

All compute goes through libohmhip.so (include/ohmhip.h).  There is no CPU implementation here.
The C++14 mirror of the same classes lives in ohm_amd/host/ (header-only, over the same C ABI).
"""
import ctypes as C
import enum
import math

import numpy as np

from . import _lib as L


class RayFlag(enum.IntFlag):
    """ohm/RayFlag.h:16-60"""
    kRfDefault = 0
    kRfEndPointAsFree = 1 << 0
    kRfStopOnFirstOccupied = 1 << 1
    kRfExcludeOrigin = 1 << 2
    kRfExcludeSample = 1 << 3
    kRfExcludeRay = 1 << 4
    kRfExcludeUnobserved = 1 << 5
    kRfExcludeFree = 1 << 6
    kRfExcludeOccupied = 1 << 7
    kRfReverseWalk = 1 << 8


class NdtMode(enum.IntEnum):
    """ohm/NdtMode.h"""
    kNone = 0
    kOccupancy = 1
    kTraversability = 2


# Layer name -> (layer id, numpy dtype, components).  ohm/DefaultLayer.cpp:76-311
LAYERS = {
    "occupancy": (L.LID_OCCUPANCY, np.float32, 1),
    "mean": (L.LID_MEAN, np.uint32, 2),
    "covariance": (L.LID_COVARIANCE, np.float32, 6),
    "traversal": (L.LID_TRAVERSAL, np.float32, 1),
    "touch_time": (L.LID_TOUCH_TIME, np.uint32, 1),
    "incident_normal": (L.LID_INCIDENT, np.uint32, 1),
    "intensity": (L.LID_INTENSITY, np.float32, 2),
    "hit_miss_count": (L.LID_HIT_MISS, np.uint32, 2),
    "tsdf": (L.LID_TSDF, np.float32, 2),
}


_libm = C.CDLL("libm.so.6")
_libm.logf.restype = C.c_float
_libm.logf.argtypes = [C.c_float]
_libm.expf.restype = C.c_float
_libm.expf.argtypes = [C.c_float]


def probability_to_value(p):
    """ohm/MapProbability.h:33-36, float instantiation: std::log(float) == libm logf (numpy's float32 log is not
    bit-identical to libm, and these constants must match the C++ host / CPU mapper exactly)."""
    p = np.float32(p)
    return np.float32(_libm.logf(float(p / (np.float32(1.0) - p))))


def value_to_probability(v):
    """ohm/MapProbability.h:20-27"""
    v = np.float32(v)
    if v == -np.inf:
        return np.float32(0)
    return np.float32(1) - (np.float32(1) / (np.float32(1) + np.float32(_libm.expf(float(v)))))


class OccupancyMap:
    """Host-side map description + chunk storage in the reference's MapChunk layout.

    chunks[(rx, ry, rz)][layer_name] is a flat numpy array indexed x + y*dx + z*dx*dy (ohm/MapChunk.h:33-50), i.e. the
    bytes of VoxelBlock::voxelBytes() for that layer (ohm/VoxelBlock.h:270-278).
    """

    def __init__(self, resolution=0.1, region_voxel_dimensions=(32, 32, 32), layers=("occupancy",)):
        self.resolution = float(resolution)
        self.region_voxel_dimensions = tuple(int(d) if d > 0 else 32 for d in region_voxel_dimensions)
        self.origin = (0.0, 0.0, 0.0)
        self.layers = list(layers)
        # ohm/OccupancyMap.cpp:205-213
        self.min_voxel_value = np.float32(-2.0)
        self.max_voxel_value = np.float32(3.511)
        self.hit_value = probability_to_value(0.9)
        self.miss_value = probability_to_value(0.45)
        self.occupancy_threshold_value = probability_to_value(0.5)
        self.saturate_at_min_value = False
        self.saturate_at_max_value = False
        self.ray_filter = ("good", 1e10)  # ohm/OccupancyMap.cpp:215-218
        self.chunks = {}

    # -- ohm::OccupancyMap setters used by the tests ---------------------------------------------------------------
    def setOrigin(self, origin):
        self.origin = tuple(float(v) for v in origin)

    def setHitProbability(self, p):
        self.hit_value = probability_to_value(p)

    def setMissProbability(self, p):
        self.miss_value = probability_to_value(p)

    def setOccupancyThresholdProbability(self, p):
        self.occupancy_threshold_value = probability_to_value(p)

    def setHitValue(self, value):
        """ohm/OccupancyMap.h:623: the log-odds adjustment of a hit, set directly."""
        self.hit_value = float(np.float32(value))

    def setMissValue(self, value):
        self.miss_value = float(np.float32(value))

    def hitValue(self):
        return self.hit_value

    def missValue(self):
        return self.miss_value

    def missProbability(self):
        return value_to_probability(self.miss_value)

    def regionVoxelVolume(self):
        d = self.region_voxel_dimensions
        return d[0] * d[1] * d[2]

    def addLayer(self, name):
        if name not in LAYERS:
            raise KeyError(name)
        if name not in self.layers:
            self.layers.append(name)

    def regionCount(self):
        return len(self.chunks)


class RayMapper:
    """ohm/RayMapper.h:22-65"""

    def valid(self):
        raise NotImplementedError

    def integrateRays(self, rays, intensities=None, timestamps=None, ray_update_flags=RayFlag.kRfDefault):
        raise NotImplementedError


class GpuMap(RayMapper):
    """ohm::GpuMap (ohmgpu/GpuMap.h:143-384) over libohmhip.so.

    integrateRays() is asynchronous like the reference (returns once the batch is queued); syncVoxels() is the fence
    and copies modified regions back into map.chunks.
    """
    _mode = L.MODE_OCCUPANCY

    def __init__(self, map_, borrowed_map=True, expected_element_count=2048, gpu_mem_size=0, region_capacity=0):
        self._map = map_
        self._borrowed = borrowed_map
        self._handle = L._vp()
        self._ok = False
        self._block_addr = {}  # layer name -> {region key: (block, address)}: syncVoxels' destination pointers
        self._ray_segment_length = 0.0
        self._configure_layers()
        cfg = L.MapConfig()
        L.lib.ohmhip_map_config_default(C.byref(cfg))
        cfg.resolution = map_.resolution
        for a in range(3):
            cfg.region_dim[a] = map_.region_voxel_dimensions[a]
            cfg.origin[a] = map_.origin[a]
        layer_bits = 0
        for name in map_.layers:
            layer_bits |= 1 << LAYERS[name][0]
        cfg.layers = layer_bits
        cfg.mode = self._mode
        self._fill_map_values(cfg)
        cfg.gpu_mem_size = int(gpu_mem_size)
        cfg.region_capacity = int(region_capacity)
        self._fill_config(cfg)
        self._cfg = cfg
        status = L.lib.ohmhip_map_create(C.byref(self._handle), C.byref(cfg))
        # gputil::Exception from the ctor on allocation failure (ohmgpu/GpuMap.h:53-54,159-160)
        L.check(status, "GpuMap: ohmhip_map_create")
        self._ok = True
        if expected_element_count > 2048:
            # the reference sizes its ray / key buffers for expected_element_count points in the constructor
            # (ohmgpu/GpuMap.cpp:429-470); the default (2048) is left to the first batch
            # best effort: a reservation the device cannot hold is not an error, the batches grow their buffers on demand
            L.lib.ohmhip_map_reserve_rays(self._handle, int(expected_element_count) // 2)
        self._upload_existing()

    def _fill_map_values(self, cfg):
        """The host map's probabilities, clamps and built-in filter: re-read before every batch, like the reference,
        whose GpuMap takes them from the OccupancyMap at each launch (ohmgpu/GpuMap.cpp:1036-1191)."""
        map_ = self._map
        cfg.hit_value = float(map_.hit_value)
        cfg.miss_value = float(map_.miss_value)
        cfg.threshold_value = float(map_.occupancy_threshold_value)
        cfg.min_value = float(map_.min_voxel_value)
        cfg.max_value = float(map_.max_voxel_value)
        cfg.saturate_at_min = int(map_.saturate_at_min_value)
        cfg.saturate_at_max = int(map_.saturate_at_max_value)
        mode, rng = map_.ray_filter if map_.ray_filter else ("none", 0.0)
        cfg.ray_filter = {"none": L.FILTER_NONE, "good": L.FILTER_GOOD, "clip": L.FILTER_CLIP}[mode]
        cfg.ray_filter_range = float(rng)

    def _push_config_if_changed(self):
        now = L.MapConfig.from_buffer_copy(bytes(self._cfg))
        self._fill_map_values(now)
        self._fill_config(now)
        if bytes(now) != bytes(self._cfg):
            L.check(L.lib.ohmhip_map_update_config(self._handle, C.byref(now)), "update_config")
            self._cfg = now

    # hooks for subclasses ---------------------------------------------------------------------------------------
    def _configure_layers(self):
        if "occupancy" not in self._map.layers:
            self._map.addLayer("occupancy")

    def _fill_config(self, cfg):
        pass

    # ------------------------------------------------------------------------------------------------------------
    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if self._handle:
            L.lib.ohmhip_map_destroy(self._handle)
            self._handle = L._vp()
            self._ok = False

    def gpuOk(self):
        return self._ok

    def valid(self):
        return self._ok

    def map(self):
        return self._map

    def borrowedMap(self):
        return self._borrowed

    def hitValue(self):
        return self._map.hit_value

    def missValue(self):
        return self._map.miss_value

    def setHitValue(self, value):
        """Pass-through to OccupancyMap::setHitValue for API compatibility (ohmgpu/GpuMap.h:234-236); the device takes
        the new value with the next batch."""
        self._map.setHitValue(value)

    def setMissValue(self, value):
        """ohmgpu/GpuMap.h:242-244."""
        self._map.setMissValue(value)

    def setGroupedRays(self, group):
        """ohmgpu/GpuMap.h:271, 323: the reference can sort a batch's rays by region before upload to help its GPU
        threads.  Here every batch is binned per region on the device: the flag is stored and otherwise ignored."""
        self._grouped_rays = bool(group)

    def groupedRays(self):
        return getattr(self, "_grouped_rays", False)

    def setRaySegmentLength(self, length):
        """ohmgpu/GpuMap.h:246-262.  Segmentation exists in the reference to balance GPU threads; this backend bins
        rays per region instead, so the value is stored and otherwise ignored (results follow the CPU mapper)."""
        self._ray_segment_length = float(length)

    def raySegmentLength(self):
        return self._ray_segment_length

    # GpuMap::setRayFilter / rayFilter / effectiveRayFilter / clearRayFilter (ohmgpu/GpuMap.cpp:348-369).  A filter is a
    # callable of the vectorised form described in ohm_amd/rayfilter.py; without one the map's built-in filter
    # (OccupancyMap.ray_filter) runs on the device.
    def setRayFilter(self, ray_filter):
        self._ray_filter = ray_filter

    def rayFilter(self):
        return getattr(self, "_ray_filter", None)

    def effectiveRayFilter(self):
        return self.rayFilter()

    def clearRayFilter(self):
        self._ray_filter = None

    def integrateRays(self, rays, intensities=None, timestamps=None, ray_update_flags=RayFlag.kRfDefault):
        """rays: (2N, 3) float64 origin/sample pairs.  Returns number of POINTS integrated (2 per ray), 0 on failure
        (ohmgpu/GpuMap.cpp:416, 548-551, 874)."""
        if not self._ok:
            return 0
        self._push_config_if_changed()
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
        element_count = rays.shape[0]
        if element_count < 2:
            return 0
        ints = None if intensities is None else np.ascontiguousarray(intensities, dtype=np.float32)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, dtype=np.float64)
        done = C.c_size_t(0)
        if self.rayFilter() is not None:
            # The reference runs the RayFilterFunction per ray on the host before upload (ohmgpu/GpuMap.cpp:736-746):
            # rejected rays are dropped, the others go on with their (possibly moved) end points and filter flags.
            keep, starts, ends, fflags = self.rayFilter()(rays[0::2].copy(), rays[1::2].copy())
            keep = np.asarray(keep, dtype=bool)
            n_keep = int(keep.sum())
            if n_keep == 0:
                return 0
            kept = np.empty((2 * n_keep, 3), dtype=np.float64)
            kept[0::2] = np.asarray(starts, dtype=np.float64)[keep]
            kept[1::2] = np.asarray(ends, dtype=np.float64)[keep]
            fflags = np.ascontiguousarray(np.asarray(fflags, dtype=np.uint8)[keep])
            ints = None if ints is None else np.ascontiguousarray(ints[keep])
            ts = None if ts is None else np.ascontiguousarray(ts[keep])
            status = L.lib.ohmhip_map_integrate_rays_filtered(
                self._handle, kept.ctypes.data, kept.shape[0], None if ints is None else ints.ctypes.data,
                None if ts is None else ts.ctypes.data, int(ray_update_flags), fflags.ctypes.data, C.byref(done))
            if status == L.ERR_UNSUPPORTED:
                raise L.OhmHipError(status, "GpuMap.integrateRays")
            if status != L.OK:
                self._last_error = status
                return 0
            return int(done.value)
        status = L.lib.ohmhip_map_integrate_rays(
            self._handle, rays.ctypes.data, element_count, None if ints is None else ints.ctypes.data,
            None if ts is None else ts.ctypes.data, int(ray_update_flags), C.byref(done))
        if status == L.ERR_UNSUPPORTED:
            raise L.OhmHipError(status, "GpuMap.integrateRays")
        if status != L.OK:
            self._last_error = status
            self._last_partial = int(done.value)  # leading elements a split batch did integrate (include/ohmhip.h)
            return 0
        return int(done.value)

    def lastPartialCount(self):
        """After a failed integrateRays: how many leading elements of that call WERE integrated (non-zero only when a
        batch over the residency limit was split and a later part still did not fit); do not present those again."""
        return getattr(self, "_last_partial", 0)

    def integrateRaysDevice(self, d_rays_ptr, element_count, ray_update_flags=RayFlag.kRfDefault, d_intensities=None,
                            d_timestamps=None):
        """Rays already resident in HBM (bench path): d_rays_ptr is a raw device pointer to element_count dvec3
        (d_intensities / d_timestamps: optional device pointers to one float / double per ray).  The arrays must be
        complete when the call is made (synchronise the stream that produced them): the map reads them on streams of its
        own."""
        done = C.c_size_t(0)
        status = L.lib.ohmhip_map_integrate_rays_device(self._handle, d_rays_ptr, element_count, d_intensities,
                                                        d_timestamps, int(ray_update_flags), C.byref(done))
        if status != L.OK:
            self._last_partial = int(done.value)
            raise L.OhmHipError(status, "GpuMap.integrateRaysDevice (%d leading elements integrated)" % done.value)
        return int(done.value)

    def stats(self):
        st = L.BatchStats()
        L.check(L.lib.ohmhip_map_last_stats(self._handle, C.byref(st)), "stats")
        return {name: getattr(st, name) for name, _ in L.BatchStats._fields_}

    def batchTimings(self, batches_back=0):
        """Device phase times (ms) of one of the last 32 batches: total, setup + bin, walk kernel, order + apply."""
        ms = (C.c_float * 4)()
        L.check(L.lib.ohmhip_map_batch_timings(self._handle, batches_back, ms), "batch_timings")
        return {"ms_total": ms[0], "ms_setup": ms[1], "ms_walk": ms[2], "ms_apply": ms[3]}

    def setFirstRayTime(self, time):
        """OccupancyMap::setFirstRayTime (ohm/OccupancyMap.h:346): the base the touch-time layer is encoded against.  The
        ranks of a partitioned map share one (PartitionedIntegrator sets it from the first stamp of the whole job)."""
        L.check(L.lib.ohmhip_map_set_first_ray_time(self._handle, float(time)), "set_first_ray_time")

    def firstRayTime(self):
        t = C.c_double(-1.0)
        L.check(L.lib.ohmhip_map_first_ray_time(self._handle, C.byref(t)), "first_ray_time")
        return float(t.value)

    def setPhaseTiming(self, enable=True):
        """Record the start markers of the set-up and binning passes too (ms_setup, and ms_total of a batch on its own as
        first kernel start -> last kernel end): a few microseconds per batch, off by default (include/ohmhip.h)."""
        L.check(L.lib.ohmhip_map_set_phase_timing(self._handle, 1 if enable else 0), "set_phase_timing")

    def batchesLaunched(self):
        """Device batches launched so far (calls that only collect their rays launch none)."""
        n = C.c_uint64(0)
        L.check(L.lib.ohmhip_map_batches_launched(self._handle, C.byref(n)), "batches_launched")
        return int(n.value)

    def raysBeyondTiles(self):
        """Maps with regions above 32768 voxels: rays cut because a tile coordinate left the key range although the
        reference addresses the region (include/ohmhip.h: ohmhip_map_rays_beyond_tiles).  0 for ordinary regions."""
        n = C.c_uint64(0)
        L.check(L.lib.ohmhip_map_rays_beyond_tiles(self._handle, C.byref(n)), "rays_beyond_tiles")
        return int(n.value)

    def wait(self):
        L.check(L.lib.ohmhip_map_sync(self._handle), "sync")

    def gpuCache(self):
        """GpuMap::gpuCache(): the MapRegionCache face of the reference's GpuCache (flush / clear / remove) as a view of
        this map -- the whole map is resident, there is no separate cache object."""
        owner = self

        class _GpuCacheView:
            def flush(self):
                owner.syncVoxels()

            def clear(self):
                owner.clear()

            def remove(self, region_key):
                owner.removeRegions([region_key])

            def reinitialise(self):
                """ohmgpu/GpuCache.h:103: the device map keeps its layout for life, so this is clear()."""
                owner.clear()

            def targetGpuAllocSize(self):
                """ohmgpu/GpuCache.h:139: byte budget of the device-side voxel storage (0: the device's free memory)."""
                return int(owner.cacheStats()["memory_limit"])

            def layerCount(self):
                """ohmgpu/GpuCache.h:143"""
                return len(owner.map().layers)

        return _GpuCacheView()

    def removeRegions(self, keys):
        """MapRegionCache::remove (what OccupancyMap::cullRegions calls on the GPU cache, ohm/OccupancyMap.cpp:1202-1234):
        drop the listed regions from the device map.  Returns how many were resident."""
        keys = np.ascontiguousarray(keys, dtype=np.int16).reshape(-1, 3)
        removed = C.c_size_t(0)
        L.check(L.lib.ohmhip_map_remove_regions(self._handle, keys.ctypes.data, len(keys), C.byref(removed)),
                "removeRegions")
        return int(removed.value)

    def clear(self):
        """OccupancyMap::clear() as the GPU cache sees it (GpuCache::clear, ohmgpu/GpuCache.cpp): drop every resident
        region; the host map's chunks are the caller's to clear."""
        L.check(L.lib.ohmhip_map_clear(self._handle), "clear")

    def regionKeys(self, dirty_only=False):
        n = C.c_size_t(0)
        fn = L.lib.ohmhip_map_dirty_regions if dirty_only else L.lib.ohmhip_map_regions
        L.check(fn(self._handle, None, 0, C.byref(n)), "regions")
        keys = np.zeros((n.value, 3), dtype=np.int16)
        if n.value:
            L.check(fn(self._handle, keys.ctypes.data, n.value, C.byref(n)), "regions")
        return keys

    def syncVoxels(self, layer_names=None):
        """ohmgpu/GpuMap.cpp:308-345: fence + copy modified regions back into the host MapChunk blocks -- every layer,
        or only `layer_names` (the regions then stay marked as modified: there is one mark per region, not per layer, so
        a later full syncVoxels() still brings the other layers over)."""
        if not self._ok:
            return
        keys = self.regionKeys(dirty_only=True)
        names = layer_names if layer_names is not None else self._map.layers
        rv = self._map.regionVoxelVolume()
        key_tuples = [tuple(k) for k in keys.tolist()]
        chunks = self._map.chunks
        for name in names:
            lid, dtype, comps = LAYERS[name]
            # (block addresses are remembered per chunk: `ndarray.ctypes` costs a microsecond per block, which at C1's
            # 1243 regions is a third of the copy itself)
            cache = self._block_addr.setdefault(name, {})
            ptrs = np.empty(len(key_tuples), dtype=np.uint64)
            for i, k in enumerate(key_tuples):
                chunk = chunks.get(k)
                if chunk is None:
                    chunk = chunks[k] = {}
                block = chunk.get(name)
                if block is None:
                    block = chunk[name] = np.empty(rv * comps, dtype=dtype)  # fully overwritten by the copy below
                cached = cache.get(k)
                if cached is None or cached[0] is not block:
                    cached = cache[k] = (block, block.ctypes.data)
                ptrs[i] = cached[1]
            if not len(key_tuples):
                continue
            L.check(L.lib.ohmhip_map_read_regions(self._handle, lid, keys.ctypes.data, len(key_tuples),
                                                  ptrs.ctypes.data_as(C.POINTER(C.c_void_p))), "syncVoxels")
        if layer_names is None:
            L.check(L.lib.ohmhip_map_clear_dirty(self._handle), "clear_dirty")
        self.wait()

    def cacheStats(self, reset=False):
        """GpuLayerCache::queryStats (ohmgpu/GpuLayerCache.h:334-339) for the resident region pool: hits / misses / full
        plus the pool's size (include/ohmhip.h: ohmhip_cache_stats)."""
        st = L.CacheStats()
        L.check(L.lib.ohmhip_map_cache_stats(self._handle, C.byref(st), 1 if reset else 0), "cache_stats")
        return {name: getattr(st, name) for name, _ in L.CacheStats._fields_}

    def setMemoryLimit(self, nbytes):
        """Bound the region pool (include/ohmhip.h "RESIDENCY LIMIT"): a batch that needs more fails with
        OHMHIP_ERR_CAPACITY and leaves the map as it was.  0 removes the bound."""
        L.check(L.lib.ohmhip_map_set_memory_limit(self._handle, int(nbytes)), "set_memory_limit")

    def setSpillToHost(self, enable=True):
        """With a memory limit set: instead of failing, a batch that needs more regions than fit moves the least
        recently used resident regions to a host store inside the library and repeats; stored regions stay part of the
        map (listed, synced, brought back when touched).  include/ohmhip.h "SPILL TO HOST"; the reference's LRU reuse
        of cache slots (ohmgpu/GpuLayerCache.cpp:530-584) serves the same purpose."""
        L.check(L.lib.ohmhip_map_set_spill_to_host(self._handle, 1 if enable else 0), "set_spill_to_host")

    def setSpillWriteback(self, enable=True):
        """Opt-in background write-back of the spill path (include/ohmhip.h "WRITE-BACK"): the regions the eviction
        policy would take next are copied to the host store while batches run, so an eviction finds them clean."""
        L.check(L.lib.ohmhip_map_set_spill_writeback(self._handle, 1 if enable else 0), "set_spill_writeback")

    def setBatchCoalescing(self, min_rays):
        """Collect consecutive small host batches and run them as one device batch of >= min_rays rays
        (include/ohmhip.h: ohmhip_map_set_batch_coalescing; on by default with 65536).  0 turns it off."""
        L.check(L.lib.ohmhip_map_set_batch_coalescing(self._handle, int(min_rays)), "setBatchCoalescing")

    def setAsyncLaunch(self, enable=True):
        """Large host-pointer batches return once their rays are staged; the launch sequence runs on a thread of the map
        (include/ohmhip.h: ohmhip_map_set_async_launch).  Off by default."""
        L.check(L.lib.ohmhip_map_set_async_launch(self._handle, 1 if enable else 0), "setAsyncLaunch")

    def setRegionOwnership(self, world_size, rank, block_shift=0):
        """Owner-computes multi-GPU mode (include/ohmhip.h: ohmhip_map_set_region_ownership): integrate only what falls
        in the regions `rank` owns among `world_size` region-partitioned maps.  Call before the first integrateRays."""
        L.check(L.lib.ohmhip_map_set_region_ownership(self._handle, int(world_size), int(rank), int(block_shift)),
                "setRegionOwnership")
        self._ownership = (int(world_size), int(rank), int(block_shift))

    def setRegionPartition(self, partition):
        """Partitioned multi-GPU mode (include/ohmhip.h "Partitioned map"): like setRegionOwnership, with the region blocks
        dealt by `partition` (ohm_amd.distributed.RegionPartition: a block -> rank table, or the block hash).  Call
        before the first integrateRays."""
        L.check(L.lib.ohmhip_map_set_region_partition(self._handle, C.byref(partition.c_struct(self_rank=True))),
                "setRegionPartition")
        self._partition = partition
        self._ownership = (partition.world_size, partition.rank, partition.block_shift)

    def regionOwners(self, keys):
        """Owner rank of each int16 (n, 3) region key under this map's partition."""
        keys = np.ascontiguousarray(keys, dtype=np.int16).reshape(-1, 3)
        owners = np.zeros(len(keys), dtype=np.uint32)
        L.check(L.lib.ohmhip_map_region_owners(self._handle, keys.ctypes.data, len(keys), owners.ctypes.data),
                "regionOwners")
        return owners

    def routeRays(self, d_rays, ray_count, d_routed, capacity, ray_update_flags=0, d_index=None):
        """ohmhip_map_route_rays: destinations of `ray_count` rays in device memory under this map's partition, compacted
        per destination into `d_routed` (device pointer, `capacity` rays).  Returns (counts per rank, voxel visits of the
        input rays, fits): fits is False when the routed rays exceed `capacity` (grow and repeat)."""
        world = max(1, self._ownership[0]) if getattr(self, "_ownership", None) else 1
        counts = np.zeros(world, dtype=np.uint32)
        visits = C.c_uint64(0)
        status = L.lib.ohmhip_map_route_rays(self._handle, d_rays, int(ray_count), int(ray_update_flags), d_routed,
                                             d_index, int(capacity), counts.ctypes.data, C.byref(visits))
        if status not in (L.OK, L.ERR_CAPACITY):
            L.check(status, "routeRays")
        return counts, int(visits.value), status == L.OK

    def lineKeys(self, lines, max_keys_per_line=1024):
        """LineKeysQueryGpu equivalent: voxel keys along each query line (start/end pairs, (2N, 3) float64).
        Returns (keys, counts): keys is (N, max_keys, 10) uint8 viewed as int16 region[3] + uint8 voxel[4]."""
        lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 6)
        n = lines.shape[0]
        keys = np.zeros((n, max_keys_per_line, 10), dtype=np.uint8)
        counts = np.zeros(n, dtype=np.uint32)
        L.check(L.lib.ohmhip_map_line_keys(self._handle, lines.ctypes.data, n, max_keys_per_line, keys.ctypes.data,
                                           counts.ctypes.data), "lineKeys")
        regions = keys[:, :, :6].copy().view(np.int16).reshape(n, max_keys_per_line, 3)
        voxels = keys[:, :, 6:9]
        return regions, voxels, counts

    def _upload_existing(self):
        """gpumap::enableGpu + GpuLayerCache::upload for regions the CPU map already holds."""
        self.uploadRegions()

    def uploadRegions(self, keys=None):
        """Push host chunks to the device: all of them, or the listed regions -- what GpuLayerCache::upload does for a
        region whose CPU copy is newer than the device's (ohmgpu/GpuLayerCache.cpp:172-182), e.g. after CPU-side
        integration into the host map.  Layers a chunk does not hold keep their device contents."""
        if not self._map.chunks:
            return 0
        if keys is None:
            keys = sorted(self._map.chunks.keys())
        keys = np.ascontiguousarray(keys, dtype=np.int16).reshape(-1, 3)
        keys = np.array([k for k in keys if tuple(int(v) for v in k) in self._map.chunks], dtype=np.int16).reshape(-1, 3)
        for name in self._map.layers:
            lid, dtype, comps = LAYERS[name]
            have = [k for k in keys if name in self._map.chunks[tuple(int(v) for v in k)]]
            if not have:
                continue
            sub = np.ascontiguousarray(np.array(have, dtype=np.int16).reshape(-1, 3))
            blocks = [np.ascontiguousarray(self._map.chunks[tuple(int(v) for v in k)][name], dtype=dtype) for k in sub]
            ptrs = (C.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
            L.check(L.lib.ohmhip_map_write_regions(self._handle, lid, sub.ctypes.data, len(blocks), ptrs), "upload")
        return len(keys)


class GpuNdtMap(GpuMap):
    """ohm::GpuNdtMap (ohmgpu/GpuNdtMap.h:63-132)."""
    _mode = L.MODE_NDT_OM

    def __init__(self, map_, borrowed_map=True, expected_element_count=2048, gpu_mem_size=0,
                 ndt_mode=NdtMode.kOccupancy, region_capacity=0):
        self._ndt_mode = NdtMode(ndt_mode)
        self._mode = L.MODE_NDT_TM if self._ndt_mode == NdtMode.kTraversability else L.MODE_NDT_OM
        # ohm/private/NdtMapDetail.h:20-45
        self.sensor_noise = 0.05
        self.sample_threshold = 3
        mp = float(map_.missProbability())
        # ohm/NdtMap.h:146-149
        self.adaptation_rate = float(np.float32(max(0.0, min(np.float32(2.0) * (np.float32(1.0) - np.float32(2.0) *
                                                                                   np.float32(mp)), 1.0))))
        self.reinitialise_covariance_threshold = float(probability_to_value(0.2))
        self.reinitialise_covariance_point_count = 100
        self.initial_intensity_covariance = 1.0
        super().__init__(map_, borrowed_map, expected_element_count, gpu_mem_size, region_capacity)

    def _configure_layers(self):
        # NdtMap::enableNdt adds mean + covariance (+ intensity, hit/miss for TM). ohm/NdtMap.cpp:194-213
        for name in ("occupancy", "mean", "covariance"):
            self._map.addLayer(name)
        if self._ndt_mode == NdtMode.kTraversability:
            self._map.addLayer("intensity")
            self._map.addLayer("hit_miss_count")

    def _fill_config(self, cfg):
        cfg.ndt_sensor_noise = self.sensor_noise
        cfg.ndt_sample_threshold = self.sample_threshold
        cfg.ndt_adaptation_rate = self.adaptation_rate
        cfg.ndt_reinit_threshold = self.reinitialise_covariance_threshold
        cfg.ndt_reinit_count = self.reinitialise_covariance_point_count
        cfg.ndt_initial_intensity_cov = self.initial_intensity_covariance

    def setSensorNoise(self, noise):
        """GpuNdtMap::setSensorNoise (ohmgpu/GpuNdtMap.h:63-132): applies from the next batch."""
        self.sensor_noise = float(noise)

    def sensorNoise(self):
        return self.sensor_noise


class GpuTsdfMap(GpuMap):
    """ohm::GpuTsdfMap (ohmgpu/GpuTsdfMap.h:37-94)."""
    _mode = L.MODE_TSDF

    def __init__(self, map_, borrowed_map=True, expected_element_count=2048, gpu_mem_size=0, max_weight=1e4,
                 default_truncation_distance=0.1, dropoff_epsilon=0.0, sparsity_compensation_factor=1.0,
                 region_capacity=0):
        self.tsdf_options = (max_weight, default_truncation_distance, dropoff_epsilon, sparsity_compensation_factor)
        super().__init__(map_, borrowed_map, expected_element_count, gpu_mem_size, region_capacity)

    def _configure_layers(self):
        self._map.addLayer("tsdf")

    def _fill_config(self, cfg):
        (cfg.tsdf_max_weight, cfg.tsdf_trunc, cfg.tsdf_dropoff, cfg.tsdf_sparsity) = self.tsdf_options

    # ohmgpu/GpuTsdfMap.h:64-79: the option setters / getters under the reference's names (they apply from the next batch;
    # `tsdf_options` = (max_weight, default_truncation_distance, dropoff_epsilon, sparsity_compensation_factor)).
    def setTsdfOptions(self, max_weight, default_truncation_distance, dropoff_epsilon, sparsity_compensation_factor):
        self.tsdf_options = (float(max_weight), float(default_truncation_distance), float(dropoff_epsilon),
                             float(sparsity_compensation_factor))

    def _set_option(self, index, value):
        opts = list(self.tsdf_options)
        opts[index] = float(value)
        self.tsdf_options = tuple(opts)

    def setMaxWeight(self, max_weight):
        self._set_option(0, max_weight)

    def maxWeight(self):
        return self.tsdf_options[0]

    def setDefaultTruncationDistance(self, default_truncation_distance):
        self._set_option(1, default_truncation_distance)

    def defaultTruncationDistance(self):
        return self.tsdf_options[1]

    def setDropoffEpsilon(self, dropoff_epsilon):
        self._set_option(2, dropoff_epsilon)

    def dropoffEpsilon(self):
        return self.tsdf_options[2]

    def setSparsityCompensationFactor(self, sparsity_compensation_factor):
        self._set_option(3, sparsity_compensation_factor)

    def sparsityCompensationFactor(self):
        return self.tsdf_options[3]


class GpuTransformSamples:
    """ohm::GpuTransformSamples (ohmgpu/GpuTransformSamples.h:30-83): local sensor samples + a timestamped trajectory ->
    world-frame ray pairs in a device buffer that integrateRaysDevice() consumes directly."""

    def __init__(self):
        self._buffer = L._vp()
        L.check(L.lib.ohmhip_buffer_create(C.byref(self._buffer), 48, 3), "buffer_create")

    def close(self):
        if self._buffer:
            L.lib.ohmhip_buffer_destroy(self._buffer)
            self._buffer = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, transform_times, transform_translations, transform_rotations_xyzw, sample_times, local_samples,
                  max_range=float("inf")):
        """Returns (device pointer, element_count): element_count = 2 x valid samples, as the reference returns."""
        times = np.ascontiguousarray(transform_times, dtype=np.float64)
        tr = np.ascontiguousarray(transform_translations, dtype=np.float64).reshape(-1, 3)
        rot = np.ascontiguousarray(transform_rotations_xyzw, dtype=np.float64).reshape(-1, 4)
        st = np.ascontiguousarray(sample_times, dtype=np.float64)
        pts = np.ascontiguousarray(local_samples, dtype=np.float64).reshape(-1, 3)
        count = C.c_uint32(0)
        L.check(L.lib.ohmhip_transform_samples(times.ctypes.data, tr.ctypes.data, rot.ctypes.data, times.shape[0],
                                               st.ctypes.data, pts.ctypes.data, pts.shape[0], float(max_range), None,
                                               self._buffer, C.byref(count)), "transform_samples")
        ptr = L._vp()
        L.check(L.lib.ohmhip_buffer_ptr(self._buffer, C.byref(ptr)), "buffer_ptr")
        return ptr, int(count.value)

    def read(self, element_count):
        """Copy the transformed rays back to the host ((element_count, 3) float64)."""
        out = np.zeros((element_count, 3), dtype=np.float64)
        if element_count:
            L.check(L.lib.ohmhip_buffer_read(self._buffer, out.ctypes.data, out.nbytes, 0, None, None, None), "buffer_read")
        return out


class LineKeysQueryGpu:
    """ohm::LineKeysQueryGpu (ohmgpu/LineKeysQueryGpu.h; interface of ohm/LineKeysQuery.h:47-101 + ohm/Query.h:51-121): the
    voxel keys along a set of query lines, walked on the device with the CPU walk's fp64 semantics
    (ohmhip_map_line_keys).  Given the GpuMap that holds the device handle (the query reads the map's geometry only)."""

    def __init__(self, gpu_map, query_flags=0):
        self._gpu_map = gpu_map
        self._query_flags = int(query_flags)
        self._rays = np.zeros((0, 3), dtype=np.float64)
        self.reset()

    def queryFlags(self):
        return self._query_flags

    def setQueryFlags(self, flags):
        self._query_flags = int(flags)

    def setRays(self, rays):
        """(2N, 3) start / end point pairs."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
        self._rays = rays[:rays.shape[0] & ~1].copy()

    def rays(self):
        return self._rays

    def rayPointCount(self):
        return self._rays.shape[0]

    def reset(self, hard_reset=True):
        self._indices = np.zeros(0, dtype=np.uint64)
        self._counts = np.zeros(0, dtype=np.uint64)
        self._regions = np.zeros((0, 3), dtype=np.int16)
        self._locals = np.zeros((0, 3), dtype=np.uint8)

    def execute(self):
        self.reset(False)
        n = self._rays.shape[0] // 2
        if n == 0:
            return True
        # worst case keys per line, as the reference sizes its buffer (ohmgpu/LineKeysQueryGpu.cpp:112-119)
        length = np.linalg.norm(self._rays[1::2] - self._rays[0::2], axis=1)
        max_keys = int(np.ceil(length.max() / self._gpu_map.map().resolution * math.sqrt(3.0))) + 4
        regions, voxels, counts = self._gpu_map.lineKeys(self._rays, max_keys_per_line=max_keys)
        counts = np.minimum(counts.astype(np.uint64), np.uint64(max_keys))
        self._counts = counts
        self._indices = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
        keep = np.arange(max_keys)[None, :] < counts[:, None]
        self._regions = regions[keep]
        self._locals = voxels[keep]
        return True

    def executeAsync(self):
        """The device call is synchronous: the asynchronous forms complete at once (ohm/Query.h:103-121)."""
        return self.execute()

    def wait(self, timeout_ms=0xffffffff):
        return True

    def numberOfResults(self):
        return int(self._counts.shape[0])

    def resultIndices(self):
        return self._indices

    def resultCounts(self):
        return self._counts

    def intersectedVoxels(self):
        """(regions (K, 3) int16, local keys (K, 3) uint8): all rays' keys in walk order, ray after ray."""
        return self._regions, self._locals

    def ranges(self):
        return None


def device_count():
    n = C.c_int(0)
    status = L.lib.ohmhip_device_count(C.byref(n))
    return n.value if status == L.OK else 0


def device_info(device=0):
    info = L.DeviceInfo()
    L.check(L.lib.ohmhip_device_get_info(device, C.byref(info)), "device_get_info")
    return {"name": info.name.decode(), "arch": info.arch.decode(), "total_memory": info.total_memory,
            "compute_units": info.compute_units, "lds_bytes_per_block": info.lds_bytes_per_block}
