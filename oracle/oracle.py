"""ctypes wrapper of the CPU oracle (oracle/libohm_oracle.so) -- TEST INFRASTRUCTURE ONLY.

May be imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never from ohm_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libohm_oracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libohmref.so")


def build(force=False):
    """Compile the C restatement (and oracle/_ref when the reference checkout is present)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "ohm_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libohm_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/ohm") and (force or not os.path.exists(REF_LIB_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


build()
lib = C.CDLL(LIB_PATH)


class Key(C.Structure):
    _fields_ = [("region", C.c_int16 * 3), ("local", C.c_uint8 * 3), ("pad", C.c_uint8)]


LAYER_BITS = {"occupancy": 1 << 0, "mean": 1 << 1, "covariance": 1 << 2, "traversal": 1 << 3, "touch_time": 1 << 4,
              "incident_normal": 1 << 5, "intensity": 1 << 6, "hit_miss_count": 1 << 7, "tsdf": 1 << 8}
LAYER_IDS = {name: i for i, name in enumerate(LAYER_BITS)}
LAYER_DTYPES = {"occupancy": (np.float32, 1), "mean": (np.uint32, 2), "covariance": (np.float32, 6),
                "traversal": (np.float32, 1), "touch_time": (np.uint32, 1), "incident_normal": (np.uint32, 1),
                "intensity": (np.float32, 2), "hit_miss_count": (np.uint32, 2), "tsdf": (np.float32, 2)}

_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
lib.oracle_map_create.restype = _vp
lib.oracle_map_create.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint]
lib.oracle_map_destroy.argtypes = [_vp]
lib.oracle_map_set_origin.argtypes = [_vp, C.c_double, C.c_double, C.c_double]
for _n in ("hit_probability", "miss_probability", "threshold_probability", "hit_value", "miss_value"):
    getattr(lib, "oracle_map_set_" + _n).argtypes = [_vp, C.c_float]
lib.oracle_map_set_min_max.argtypes = [_vp, C.c_float, C.c_float]
lib.oracle_map_set_saturation.argtypes = [_vp, C.c_int, C.c_int]
lib.oracle_map_set_ray_filter.argtypes = [_vp, C.c_int, C.c_double]
lib.oracle_map_set_batch_filter_flags.argtypes = [_vp, _vp]
lib.oracle_map_set_batch_filter_flags.restype = None
lib.oracle_map_hit_value.restype = C.c_float
lib.oracle_map_hit_value.argtypes = [_vp]
lib.oracle_map_miss_value.restype = C.c_float
lib.oracle_map_miss_value.argtypes = [_vp]
lib.oracle_map_set_ndt.argtypes = [_vp, C.c_float, C.c_uint, C.c_float, C.c_float, C.c_uint, C.c_float, C.c_int]
lib.oracle_map_ndt_adaptation_rate.restype = C.c_float
lib.oracle_map_ndt_adaptation_rate.argtypes = [_vp]
lib.oracle_map_set_tsdf.argtypes = [_vp, C.c_float, C.c_float, C.c_float, C.c_float]
lib.oracle_voxel_key.restype = C.c_int
lib.oracle_voxel_key.argtypes = [_vp, _dp, C.POINTER(Key)]
lib.oracle_voxel_centre.argtypes = [_vp, C.POINTER(Key), _dp]
lib.oracle_walk_segment_keys.restype = C.c_size_t
lib.oracle_walk_segment_keys.argtypes = [_vp, _dp, _dp, C.c_uint, C.POINTER(Key), _dp, _dp, C.c_size_t]
lib.oracle_integrate_occupancy.restype = C.c_size_t
lib.oracle_integrate_occupancy.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_uint]
lib.oracle_integrate_ndt.restype = C.c_size_t
lib.oracle_integrate_ndt.argtypes = [_vp, _vp, C.c_size_t, _vp, _vp, C.c_uint]
lib.oracle_integrate_tsdf.restype = C.c_size_t
lib.oracle_integrate_tsdf.argtypes = [_vp, _vp, C.c_size_t]
lib.oracle_map_visit_count.restype = C.c_uint64
lib.oracle_map_visit_count.argtypes = [_vp]
lib.oracle_region_count.restype = C.c_size_t
lib.oracle_region_count.argtypes = [_vp]
lib.oracle_region_keys.restype = C.c_size_t
lib.oracle_region_keys.argtypes = [_vp, _vp, C.c_size_t]
lib.oracle_map_set_first_ray_time.argtypes = [_vp, C.c_double]
lib.oracle_region_layer.restype = _vp
lib.oracle_region_layer.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int]
for _n in ("hit", "miss", "up", "down"):
    getattr(lib, "oracle_occupancy_adjust_" + _n).argtypes = [_fp] + [C.c_float] * 6 + [C.c_int]
lib.oracle_point_to_region_coord.restype = C.c_int
lib.oracle_point_to_region_coord.argtypes = [C.c_double, C.c_double]
lib.oracle_point_to_region_voxel.restype = C.c_int
lib.oracle_point_to_region_voxel.argtypes = [C.c_double, C.c_double, C.c_double]
lib.oracle_sub_voxel_coord.restype = C.c_uint
lib.oracle_sub_voxel_coord.argtypes = [_dp, C.c_double]
lib.oracle_sub_voxel_to_local.argtypes = [C.c_uint, C.c_double, _dp]
lib.oracle_sub_voxel_update.restype = C.c_uint
lib.oracle_sub_voxel_update.argtypes = [C.c_uint, C.c_uint, _dp, C.c_double]
lib.oracle_transform_samples.restype = C.c_uint
lib.oracle_transform_samples.argtypes = [_dp, _dp, _dp, C.c_uint, _dp, _dp, C.c_uint, C.c_double, _dp]
lib.oracle_calculate_tsdf.restype = C.c_int
lib.oracle_calculate_tsdf.argtypes = [_dp, _dp, _dp, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp]
lib.oracle_calculate_hit_with_covariance.restype = C.c_int
lib.oracle_calculate_hit_with_covariance.argtypes = [_fp, _fp, _dp, _dp, C.c_uint, C.c_float, C.c_float, C.c_float,
                                                     C.c_float, C.c_uint]
lib.oracle_calculate_miss_ndt.argtypes = [_fp, _fp, C.POINTER(C.c_int), _dp, _dp, _dp, C.c_uint, C.c_float,
                                          C.c_float, C.c_float, C.c_float, C.c_uint]
lib.oracle_probability_to_value.restype = C.c_float
lib.oracle_probability_to_value.argtypes = [C.c_float]
lib.oracle_value_to_probability.restype = C.c_float
lib.oracle_value_to_probability.argtypes = [C.c_float]


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class OracleMap:
    """CPU restatement of OccupancyMap + RayMapperOccupancy / RayMapperNdt / RayMapperTsdf."""

    def __init__(self, resolution=0.1, region_dim=(32, 32, 32), layers=("occupancy",)):
        bits = 0
        for name in layers:
            bits |= LAYER_BITS[name]
        self.layers = list(layers)
        self.resolution = resolution
        self.region_dim = tuple(region_dim)
        self._h = lib.oracle_map_create(resolution, region_dim[0], region_dim[1], region_dim[2], bits)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.oracle_map_destroy(self._h)
            self._h = None

    def set_origin(self, o):
        lib.oracle_map_set_origin(self._h, *[float(v) for v in o])

    def set_ray_filter(self, mode, rng=0.0):
        lib.oracle_map_set_ray_filter(self._h, {"none": 0, "good": 1, "clip": 2}[mode], float(rng))

    def set_ndt(self, sensor_noise=0.05, sample_threshold=3, adaptation_rate=-1.0, reinit_threshold=None,
                reinit_count=100, initial_intensity_cov=1.0, ndt_tm=False):
        if reinit_threshold is None:
            reinit_threshold = lib.oracle_probability_to_value(0.2)
        lib.oracle_map_set_ndt(self._h, sensor_noise, sample_threshold, adaptation_rate, reinit_threshold,
                               reinit_count, initial_intensity_cov, int(ndt_tm))

    def set_tsdf(self, max_weight=1e4, trunc=0.1, dropoff=0.0, sparsity=1.0):
        lib.oracle_map_set_tsdf(self._h, max_weight, trunc, dropoff, sparsity)

    @property
    def handle(self):
        return self._h

    def hit_value(self):
        return lib.oracle_map_hit_value(self._h)

    def miss_value(self):
        return lib.oracle_map_miss_value(self._h)

    def voxel_key(self, p):
        k = Key()
        ok = lib.oracle_voxel_key(self._h, _d3(p), C.byref(k))
        return (tuple(k.region), tuple(k.local)) if ok else None

    def voxel_centre(self, region, local):
        k = Key((C.c_int16 * 3)(*region), (C.c_uint8 * 3)(*local), 0)
        out = (C.c_double * 3)()
        lib.oracle_voxel_centre(self._h, C.byref(k), out)
        return tuple(out)

    def walk(self, start, end, flags=0, cap=1 << 16):
        keys = (Key * cap)()
        enter = (C.c_double * cap)()
        exit_ = (C.c_double * cap)()
        n = lib.oracle_walk_segment_keys(self._h, _d3(start), _d3(end), flags, keys, enter, exit_, cap)
        n = min(n, cap)
        return ([(tuple(keys[i].region), tuple(keys[i].local)) for i in range(n)], list(enter[:n]), list(exit_[:n]))

    def _with_filter_flags(self, filter_flags, call):
        """Run one integrate call on rays the caller already filtered (per-ray RayFilterFlag bits)."""
        if filter_flags is None:
            return call()
        ff = np.ascontiguousarray(filter_flags, dtype=np.uint8)
        lib.oracle_map_set_batch_filter_flags(self._h, ff.ctypes.data)
        try:
            return call()
        finally:
            lib.oracle_map_set_batch_filter_flags(self._h, None)

    def integrate_occupancy(self, rays, timestamps=None, flags=0, filter_flags=None):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, dtype=np.float64)
        return self._with_filter_flags(filter_flags, lambda: lib.oracle_integrate_occupancy(
            self._h, rays.ctypes.data, rays.shape[0], None if ts is None else ts.ctypes.data, int(flags)))

    def integrate_ndt(self, rays, intensities=None, timestamps=None, flags=0, filter_flags=None):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
        ints = None if intensities is None else np.ascontiguousarray(intensities, dtype=np.float32)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, dtype=np.float64)
        return self._with_filter_flags(filter_flags, lambda: lib.oracle_integrate_ndt(
            self._h, rays.ctypes.data, rays.shape[0], None if ints is None else ints.ctypes.data,
            None if ts is None else ts.ctypes.data, int(flags)))

    def integrate_tsdf(self, rays, filter_flags=None):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
        return self._with_filter_flags(filter_flags,
                                       lambda: lib.oracle_integrate_tsdf(self._h, rays.ctypes.data, rays.shape[0]))

    def visit_count(self):
        return int(lib.oracle_map_visit_count(self._h))

    def region_keys(self):
        n = lib.oracle_region_count(self._h)
        keys = np.zeros((n, 3), dtype=np.int16)
        if n:
            lib.oracle_region_keys(self._h, keys.ctypes.data, n)
        return keys

    def region_layer(self, key, name):
        """Copy of one region's layer block (MapChunk layout)."""
        ptr = lib.oracle_region_layer(self._h, int(key[0]), int(key[1]), int(key[2]), LAYER_IDS[name])
        if not ptr:
            return None
        dtype, comps = LAYER_DTYPES[name]
        count = self.region_dim[0] * self.region_dim[1] * self.region_dim[2] * comps
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).copy()

    def set_first_ray_time(self, time):
        """OccupancyMap::setFirstRayTime: the base the touch-time layer is encoded against."""
        lib.oracle_map_set_first_ray_time(self._h, C.c_double(float(time)))

    def region_layer_view(self, key, name):
        """The live block itself (no copy): tests that restate a reference case which writes single voxels on the CPU
        side (Voxel<T>::write in tests/ohmtestgpu) edit the oracle's map through it.  None for an unknown region."""
        ptr = lib.oracle_region_layer(self._h, int(key[0]), int(key[1]), int(key[2]), LAYER_IDS[name])
        if not ptr:
            return None
        dtype, comps = LAYER_DTYPES[name]
        count = self.region_dim[0] * self.region_dim[1] * self.region_dim[2] * comps
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype)

    def chunks(self, names=None):
        names = names or self.layers
        return {tuple(int(v) for v in k): {n: self.region_layer(k, n) for n in names} for k in self.region_keys()}


def transform_samples(times, translations, rotations_xyzw, sample_times, local_samples, max_range=float("inf")):
    """GpuTransformSamples semantics on the CPU (fp64).  Returns the (2 * valid, 3) ray array."""
    times = np.ascontiguousarray(times, dtype=np.float64)
    tr = np.ascontiguousarray(translations, dtype=np.float64).reshape(-1, 3)
    rot = np.ascontiguousarray(rotations_xyzw, dtype=np.float64).reshape(-1, 4)
    st = np.ascontiguousarray(sample_times, dtype=np.float64)
    pts = np.ascontiguousarray(local_samples, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((2 * pts.shape[0], 3), dtype=np.float64)
    as_dp = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
    n = lib.oracle_transform_samples(as_dp(times), as_dp(tr), as_dp(rot), times.shape[0], as_dp(st), as_dp(pts),
                                     pts.shape[0], max_range, as_dp(out))
    return out[:n]
