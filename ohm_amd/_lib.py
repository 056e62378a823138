"""ctypes binding of libohmhip.so (the C ABI declared in include/ohmhip.h).

The product path is the HIP library: there is NO CPU fallback.  Importing this module when the shared library is
missing raises ImportError loudly; calling into it without a GPU returns OHMHIP_ERR_NO_DEVICE which is raised as
OhmHipError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OHMHIP_LIB: development knob for A/B timing runs of differently built libraries (scripts/build_variant.sh).
LIB_PATH = os.environ.get("OHMHIP_LIB") or os.path.join(_HERE, "lib", "libohmhip.so")


class OhmHipError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib.ohmhip_error_string(status).decode() if "lib" in globals() else str(status)
        super().__init__(f"{what}: [{status}] {msg}" if what else f"[{status}] {msg}")


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). ohm_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_NOT_FOUND, ERR_INTERNAL, ERR_PEER = (-1, -2, -3, -4,
                                                                                                       -5, -6, -7)
MERGE_SHARED_ONLY, MERGE_FULL_UNION = 0, 1

(LID_OCCUPANCY, LID_MEAN, LID_COVARIANCE, LID_TRAVERSAL, LID_TOUCH_TIME, LID_INCIDENT, LID_INTENSITY, LID_HIT_MISS,
 LID_TSDF, LID_COUNT) = range(10)
MODE_OCCUPANCY, MODE_NDT_OM, MODE_NDT_TM, MODE_TSDF = range(4)
FILTER_NONE, FILTER_GOOD, FILTER_CLIP = range(3)


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("arch", C.c_char * 64), ("total_memory", C.c_uint64),
                ("max_allocation", C.c_uint64), ("compute_units", C.c_int), ("lds_bytes_per_block", C.c_int),
                ("unified_memory", C.c_int)]


class MapConfig(C.Structure):
    _fields_ = [("resolution", C.c_double), ("region_dim", C.c_int * 3), ("origin", C.c_double * 3),
                ("layers", C.c_uint), ("mode", C.c_int), ("hit_value", C.c_float), ("miss_value", C.c_float),
                ("threshold_value", C.c_float), ("min_value", C.c_float), ("max_value", C.c_float),
                ("saturate_at_min", C.c_int), ("saturate_at_max", C.c_int), ("ray_filter", C.c_int),
                ("ray_filter_range", C.c_double), ("ndt_sensor_noise", C.c_float),
                ("ndt_sample_threshold", C.c_uint), ("ndt_adaptation_rate", C.c_float),
                ("ndt_reinit_threshold", C.c_float), ("ndt_reinit_count", C.c_uint),
                ("ndt_initial_intensity_cov", C.c_float), ("tsdf_max_weight", C.c_float), ("tsdf_trunc", C.c_float),
                ("tsdf_dropoff", C.c_float), ("tsdf_sparsity", C.c_float), ("gpu_mem_size", C.c_uint64),
                ("region_capacity", C.c_uint32)]


class MergeStats(C.Structure):
    _fields_ = [("regions_local", C.c_uint32), ("regions_union", C.c_uint32), ("regions_shared", C.c_uint32),
                ("payload_bytes", C.c_uint64), ("key_bytes", C.c_uint64), ("ms_total", C.c_float)]


class CacheStats(C.Structure):
    _fields_ = [("hits", C.c_uint64), ("misses", C.c_uint64), ("full", C.c_uint64), ("regions_resident", C.c_uint32),
                ("region_capacity", C.c_uint32), ("bytes_per_region", C.c_uint64), ("memory_limit", C.c_uint64),
                ("evictions", C.c_uint64), ("readmissions", C.c_uint64), ("regions_spilled", C.c_uint32),
                ("spill_enabled", C.c_uint32), ("writebacks", C.c_uint64), ("writeback_hits", C.c_uint64),
                ("writeback_stale", C.c_uint64)]


class Partition(C.Structure):
    _fields_ = [("world_size", C.c_uint32), ("rank", C.c_uint32), ("block_shift", C.c_int32),
                ("grid_origin", C.c_int32 * 3), ("grid_dims", C.c_uint32 * 3), ("owners", C.c_void_p)]


COMM_ID_BYTES = 128


class BatchStats(C.Structure):
    _fields_ = [("rays_in", C.c_uint64), ("rays_integrated", C.c_uint64), ("voxel_visits", C.c_uint64),
                ("ray_region_segments", C.c_uint64), ("regions_touched", C.c_uint32),
                ("regions_resident", C.c_uint32), ("ms_total", C.c_float), ("ms_setup", C.c_float),
                ("ms_walk", C.c_float), ("ms_apply", C.c_float)]


_vp = C.c_void_p
_sigs = {
    "ohmhip_error_string": (C.c_char_p, [C.c_int]),
    "ohmhip_build_id": (C.c_char_p, []),
    "ohmhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ohmhip_device_select": (C.c_int, [C.c_int]),
    "ohmhip_device_get_info": (C.c_int, [C.c_int, C.POINTER(DeviceInfo)]),
    "ohmhip_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "ohmhip_stream_destroy": (C.c_int, [_vp]),
    "ohmhip_stream_finish": (C.c_int, [_vp]),
    "ohmhip_stream_wait_event": (C.c_int, [_vp, _vp]),
    "ohmhip_event_create": (C.c_int, [C.POINTER(_vp)]),
    "ohmhip_event_destroy": (C.c_int, [_vp]),
    "ohmhip_event_record": (C.c_int, [_vp, _vp]),
    "ohmhip_event_wait": (C.c_int, [_vp]),
    "ohmhip_event_is_complete": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "ohmhip_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "ohmhip_buffer_create": (C.c_int, [C.POINTER(_vp), C.c_size_t, C.c_uint]),
    "ohmhip_buffer_destroy": (C.c_int, [_vp]),
    "ohmhip_buffer_resize": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ohmhip_buffer_size": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "ohmhip_buffer_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "ohmhip_buffer_write": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp]),
    "ohmhip_buffer_read": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp]),
    "ohmhip_buffer_fill": (C.c_int, [_vp, C.c_int, C.c_size_t, C.c_size_t, _vp]),
    "ohmhip_device_synchronize": (C.c_int, []),
    "ohmhip_buffer_fill_pattern": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp]),
    "ohmhip_buffer_copy": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp]),
    "ohmhip_buffer_flags": (C.c_int, [_vp, C.POINTER(C.c_uint)]),
    "ohmhip_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "ohmhip_host_free": (C.c_int, [_vp]),
    "ohmhip_layer_voxel_bytes": (C.c_size_t, [C.c_int]),
    "ohmhip_map_config_default": (None, [C.POINTER(MapConfig)]),
    "ohmhip_map_create": (C.c_int, [C.POINTER(_vp), C.POINTER(MapConfig)]),
    "ohmhip_map_destroy": (C.c_int, [_vp]),
    "ohmhip_map_integrate_rays": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_uint, C.POINTER(C.c_size_t)]),
    "ohmhip_map_integrate_rays_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_uint,
                                                   C.POINTER(C.c_size_t)]),
    "ohmhip_map_sync": (C.c_int, [_vp]),
    "ohmhip_map_last_stats": (C.c_int, [_vp, C.POINTER(BatchStats)]),
    "ohmhip_map_region_count": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "ohmhip_map_regions": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ohmhip_map_dirty_regions": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ohmhip_map_clear_dirty": (C.c_int, [_vp]),
    "ohmhip_map_read_regions": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp]),
    "ohmhip_map_write_regions": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp]),
    "ohmhip_map_clear": (C.c_int, [_vp]),
    "ohmhip_map_remove_regions": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ohmhip_transform_samples": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_double, _vp, _vp,
                                           C.POINTER(C.c_uint32)]),
    "ohmhip_map_batch_timings": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_float)]),
    "ohmhip_comm_exchange_side": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp]),
    "ohmhip_gather_rows": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "ohmhip_map_reserve_rays": (C.c_int, [_vp, C.c_size_t]),
    "ohmhip_map_set_first_ray_time": (C.c_int, [_vp, C.c_double]),
    "ohmhip_map_first_ray_time": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "ohmhip_map_set_phase_timing": (C.c_int, [_vp, C.c_int]),
    "ohmhip_map_batches_launched": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "ohmhip_map_rays_beyond_tiles": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "ohmhip_map_line_keys": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "ohmhip_map_device_layer_ptr": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "ohmhip_map_region_slot": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint32)]),
    "ohmhip_map_ensure_regions": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "ohmhip_map_mark_dirty": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ohmhip_map_integrate_rays_filtered": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_uint, _vp,
                                                    C.POINTER(C.c_size_t)]),
    "ohmhip_map_update_config": (C.c_int, [_vp, C.POINTER(MapConfig)]),
    "ohmhip_map_set_batch_coalescing": (C.c_int, [_vp, C.c_size_t]),
    "ohmhip_map_set_async_launch": (C.c_int, [_vp, C.c_int]),
    "ohmhip_map_set_region_ownership": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_int]),
    "ohmhip_region_owner": (C.c_int, [_vp, C.c_size_t, C.c_int, C.c_uint32, _vp]),
    "ohmhip_map_set_region_partition": (C.c_int, [_vp, C.POINTER(Partition)]),
    "ohmhip_map_region_owners": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "ohmhip_partition_owners": (C.c_int, [C.POINTER(Partition), _vp, C.c_size_t, _vp]),
    "ohmhip_map_route_rays": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint, _vp, _vp, C.c_size_t, _vp,
                                        C.POINTER(C.c_uint64)]),
    "ohmhip_comm_exchange_counts": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ohmhip_comm_exchange_rays": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ohmhip_map_cache_stats": (C.c_int, [_vp, C.POINTER(CacheStats), C.c_int]),
    "ohmhip_map_set_memory_limit": (C.c_int, [_vp, C.c_uint64]),
    "ohmhip_map_set_spill_to_host": (C.c_int, [_vp, C.c_int]),
    "ohmhip_map_set_spill_writeback": (C.c_int, [_vp, C.c_int]),
    "ohmhip_comm_unique_id": (C.c_int, [_vp]),
    "ohmhip_comm_init_rank": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "ohmhip_comm_destroy": (C.c_int, [_vp]),
    "ohmhip_map_enable_merge": (C.c_int, [_vp]),
    "ohmhip_map_merge_replicas": (C.c_int, [_vp, _vp, C.POINTER(MergeStats)]),
    "ohmhip_map_merge_keys": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ohmhip_map_merge_pack": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp]),
    "ohmhip_map_merge_apply": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp]),
    "ohmhip_map_merge_finish": (C.c_int, [_vp]),
    "ohmhip_map_set_merge_mode": (C.c_int, [_vp, C.c_int]),
}

EXPORTED_SYMBOLS = sorted(_sigs)

for _name, (_res, _args) in _sigs.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export what include/ohmhip.h declares
    _fn.restype = _res
    _fn.argtypes = _args


def check(status, what=""):
    if status != OK:
        raise OhmHipError(status, what)
