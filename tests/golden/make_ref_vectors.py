"""Generate tests/golden/ref_vectors.npz: inputs and the outputs of the REAL reference code for them.

The outputs come from oracle/_ref/libohmref.so, which oracle/Makefile compiles from the reference's own headers where
they lie under /root/reference (ohm/MapCoord.h, ohm/VoxelOccupancyCompute.h, ohm/VoxelTsdfCompute.h,
ohm/VoxelTouchTimeCompute.h, ohm/RayFlag.h, ohmgpu/GpuKey.h); run this script only where that library can be built.  The fixture is data only (seeded
inputs + the reference's results) and is what tests/test_oracle_golden.py holds the oracle to, bit for bit, on machines
that have neither the reference checkout nor the library.

    python tests/golden/make_ref_vectors.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from ohm_amd import synth  # noqa: E402

ref = C.CDLL(O.REF_LIB_PATH)
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
ref.ref_point_to_region_coord.restype = C.c_int
ref.ref_point_to_region_coord.argtypes = [C.c_double, C.c_double]
ref.ref_point_to_region_voxel.restype = C.c_int
ref.ref_point_to_region_voxel.argtypes = [C.c_double, C.c_double, C.c_double]
ref.ref_region_centre_coord.restype = C.c_double
ref.ref_region_centre_coord.argtypes = [C.c_int, C.c_double]
ref.ref_encode_touch_time.restype = C.c_uint
ref.ref_encode_touch_time.argtypes = [C.c_double, C.c_double]
for n in ("hit", "miss", "up", "down"):
    getattr(ref, "ref_occupancy_adjust_" + n).argtypes = [_fp] + [C.c_float] * 6 + [C.c_int]
ref.ref_calculate_tsdf.restype = C.c_int
ref.ref_calculate_tsdf.argtypes = [_dp, _dp, _dp, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp]


def u(seed, n, stream):
    return synth.uniform01(seed, np.arange(n, dtype=np.uint64), stream)


def main():
    out = {}
    # --- key quantisation (ohm/MapCoord.h:32-93)
    n = 4000
    coords = (u(11, n, 0) - 0.5) * 400.0
    region_res = np.array([3.2, 4.0, 12.8, 1.6])
    out["coord_in"] = coords
    out["coord_region_res"] = region_res
    out["coord_region"] = np.array([[ref.ref_point_to_region_coord(c, r) for c in coords] for r in region_res],
                                   dtype=np.int32)
    specials = np.array([-1e-6, -9e-7, -1e-15, 0.0, 3.2, 3.2 + 5e-7, 3.2 + 9.9e-7, 3.2 - 1e-15, 3.2 + 1.1e-6, -1.1e-6])
    local = np.concatenate([u(12, n, 1) * 3.2, specials])
    out["local_in"] = local
    out["local_voxel"] = np.array([ref.ref_point_to_region_voxel(c, 0.1, 3.2) for c in local], dtype=np.int32)
    rc = np.arange(-300, 300, 7, dtype=np.int32)
    out["centre_region"] = rc
    out["centre_coord"] = np.array([[ref.ref_region_centre_coord(int(c), r) for c in rc] for r in region_res])
    # --- log-odds adjustments (ohm/VoxelOccupancyCompute.h:44-153): full cross product of the interesting values
    inf = np.float32(np.inf)
    lowest, fmax = np.float32(-3.4028234663852886e38), np.float32(3.4028234663852886e38)
    values = np.array([inf, 0.0, -2.0, 3.511, -1.95, 3.4, 0.3, -0.2006707787513733, 2.1972243785858154, -2.1, 3.6],
                      dtype=np.float32)
    rnd = ((u(13, 64, 0) - 0.5) * 8).astype(np.float32)
    values = np.concatenate([values, rnd])
    adjs = np.array([-0.2006707787513733, 2.1972243785858154, 0.0, inf, -5.0, 5.0, 0.4054651, -0.8472978],
                    dtype=np.float32)
    sats = np.array([(lowest, fmax), (-2.0, fmax), (lowest, 3.511), (-2.0, 3.511)], dtype=np.float32)
    rows = []
    for ki, (kind, limit) in enumerate((("hit", 3.511), ("up", 3.511), ("miss", -2.0), ("down", -2.0))):
        fn = getattr(ref, "ref_occupancy_adjust_" + kind)
        for v in values:
            for a in adjs:
                for smin, smax in sats:
                    for null in (0, 1):
                        y = C.c_float(v)
                        fn(C.byref(y), v, a, inf, np.float32(limit), smin, smax, null)
                        rows.append((ki, np.float32(v).view(np.uint32), np.float32(a).view(np.uint32),
                                     np.float32(limit).view(np.uint32), np.float32(smin).view(np.uint32),
                                     np.float32(smax).view(np.uint32), null, np.float32(y.value).view(np.uint32)))
    out["adjust_rows"] = np.array(rows, dtype=np.uint32)
    # --- touch time (ohm/VoxelTouchTimeCompute.h:24-37)
    base = 1.6e9 + u(14, 200, 0) * 1e3
    stamp = base + u(14, 200, 1) * 5e4
    out["touch_in"] = np.stack([base, stamp], axis=1)
    out["touch_out"] = np.array([ref.ref_encode_touch_time(b, s) for b, s in zip(base, stamp)], dtype=np.uint32)
    # --- TSDF update (ohm/VoxelTsdfCompute.h:57-136)
    n = 3000
    sensor = np.stack([(u(15, n, s) - 0.5) * 4 for s in range(3)], axis=1)
    sample = np.stack([(u(15, n, 3 + s) - 0.5) * 40 for s in range(3)], axis=1)
    frac = u(15, n, 6)
    jitter = np.stack([(u(15, n, 7 + s) - 0.5) * 0.1 for s in range(3)], axis=1)
    centre = sensor + (sample - sensor) * frac[:, None] + jitter
    w0 = (u(15, n, 10) * 50).astype(np.float32)
    d0 = ((u(15, n, 11) - 0.5) * 0.2).astype(np.float32)
    params = np.array([(0.1, 1e4, 0.0, 1.0), (0.3, 20.0, 0.05, 2.5), (10.0, 1e4, 0.0, 0.0)], dtype=np.float32)
    res = np.zeros((len(params), n, 3), dtype=np.uint32)
    for pi, (trunc, maxw, drop, sparse) in enumerate(params):
        for i in range(n):
            w, d = C.c_float(w0[i]), C.c_float(d0[i])
            r = ref.ref_calculate_tsdf((C.c_double * 3)(*sensor[i]), (C.c_double * 3)(*sample[i]),
                                       (C.c_double * 3)(*centre[i]), trunc, maxw, drop, sparse, C.byref(w), C.byref(d))
            res[pi, i] = (r, np.float32(w.value).view(np.uint32), np.float32(d.value).view(np.uint32))
    out.update(tsdf_sensor=sensor, tsdf_sample=sample, tsdf_centre=centre, tsdf_w0=w0, tsdf_d0=d0, tsdf_params=params,
               tsdf_out=res)
    # --- the device key record and the ray flags (ohmgpu/GpuKey.h:37-46, ohm/RayFlag.h:16-60): layout and values
    layout = (C.c_uint * 4)()
    ref.ref_gpukey_layout(layout)
    out["gpukey_layout"] = np.array(list(layout), dtype=np.uint32)  # sizeof, alignof, offsetof(region), offsetof(voxel)
    rk = ((u(16, 64, 0) - 0.5) * 65535).astype(np.int16).reshape(-1)
    regions = np.stack([rk, np.roll(rk, 7), np.roll(rk, 19)], axis=1)
    regions[:4] = [(-32768, 32767, 0), (0, 0, 0), (-1, 1, -1), (32767, -32768, 255)]
    voxels = (u(16, 64, 1)[:, None] * np.array([255, 254, 253, 1.999]) + np.arange(4)).astype(np.uint8) % 255
    voxels[:, 3] = voxels[:, 3] & 1
    raw = np.zeros((64, int(layout[0])), dtype=np.uint8)
    for i in range(64):
        buf = (C.c_ubyte * int(layout[0]))()
        ref.ref_gpukey_bytes((C.c_short * 3)(*[int(v) for v in regions[i]]), (C.c_ubyte * 4)(*[int(v) for v in voxels[i]]),
                             buf)
        raw[i] = np.frombuffer(buf, dtype=np.uint8)
    out.update(gpukey_regions=regions, gpukey_voxels=voxels, gpukey_bytes=raw)
    flags = (C.c_uint * 12)()
    ref.ref_ray_flags(flags)
    out["ray_flags"] = np.array(list(flags), dtype=np.uint32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
