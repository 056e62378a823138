"""Oracle restatement vs the REAL reference code: oracle/_ref/libohmref.so is compiled from the reference's own
glm-free headers where they lie (ohm/MapCoord.h, ohm/VoxelOccupancyCompute.h, ohm/VoxelTsdfCompute.h,
ohm/VoxelTouchTimeCompute.h; recipe oracle/Makefile + oracle/ref_shim.cpp).  Bit-exact agreement required.
CPU only; skipped when the prebuilt reference library is absent (it cannot be rebuilt without /root/reference)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from ohm_amd import synth

if not os.path.exists(O.REF_LIB_PATH):
    pytest.skip("oracle/_ref/libohmref.so not built (reference checkout absent)", allow_module_level=True)

ref = C.CDLL(O.REF_LIB_PATH)
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
ref.ref_point_to_region_coord.restype = C.c_int
ref.ref_point_to_region_coord.argtypes = [C.c_double, C.c_double]
ref.ref_point_to_region_voxel.restype = C.c_int
ref.ref_point_to_region_voxel.argtypes = [C.c_double, C.c_double, C.c_double]
for n in ("hit", "miss", "up", "down"):
    getattr(ref, "ref_occupancy_adjust_" + n).argtypes = [_fp] + [C.c_float] * 6 + [C.c_int]
ref.ref_calculate_tsdf.restype = C.c_int
ref.ref_calculate_tsdf.argtypes = [_dp, _dp, _dp, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp]


def _u(seed, n, stream):
    return synth.uniform01(seed, np.arange(n, dtype=np.uint64), stream)


def test_region_coord_and_voxel_quantisation():
    n = 20000
    coords = (_u(1, n, 0) - 0.5) * 200.0
    for res in (0.1 * 32, 0.25 * 16, 0.4 * 32):
        for c in coords:
            assert O.lib.oracle_point_to_region_coord(c, res) == ref.ref_point_to_region_coord(c, res)
    # local coordinates incl. the +-1e-6 boundary fix-ups (ohm/MapCoord.h:45-80)
    specials = [-1e-6, -9e-7, -1e-15, 0.0, 3.2, 3.2 + 5e-7, 3.2 + 9.9e-7, 3.2 - 1e-15, 3.2 + 1.1e-6, -1.1e-6]
    locals_ = list(_u(2, n, 1) * 3.2) + specials
    for c in locals_:
        assert O.lib.oracle_point_to_region_voxel(c, 0.1, 3.2) == ref.ref_point_to_region_voxel(c, 0.1, 3.2)


def test_occupancy_adjust_functions_bit_exact():
    inf = float("inf")
    lowest, fmax = -3.4028234663852886e38, 3.4028234663852886e38
    values = [inf, 0.0, -2.0, 3.511, -1.95, 3.4, 0.3, -0.2006707787513733, 2.1972243785858154, -2.1, 3.6]
    adjs = [-0.2006707787513733, 2.1972243785858154, 0.0, inf, -5.0, 5.0]
    sats = [(lowest, fmax), (-2.0, fmax), (lowest, 3.511), (-2.0, 3.511)]
    for kind, limit in (("hit", 3.511), ("up", 3.511), ("miss", -2.0), ("down", -2.0)):
        of = getattr(O.lib, "oracle_occupancy_adjust_" + kind)
        rf = getattr(ref, "ref_occupancy_adjust_" + kind)
        for v in values:
            for a in adjs:
                for smin, smax in sats:
                    for null in (0, 1):
                        x, y = C.c_float(v), C.c_float(v)
                        of(C.byref(x), v, a, inf, limit, smin, smax, null)
                        rf(C.byref(y), v, a, inf, limit, smin, smax, null)
                        xb = np.float32(x.value).view(np.uint32)
                        yb = np.float32(y.value).view(np.uint32)
                        assert xb == yb, (kind, v, a, smin, smax, null, x.value, y.value)


def test_calculate_tsdf_bit_exact():
    n = 5000
    sensor = np.stack([(_u(3, n, s) - 0.5) * 4 for s in range(3)], axis=1)
    sample = np.stack([(_u(3, n, 3 + s) - 0.5) * 40 for s in range(3)], axis=1)
    frac = _u(3, n, 6)
    jitter = np.stack([(_u(3, n, 7 + s) - 0.5) * 0.1 for s in range(3)], axis=1)
    centre = sensor + (sample - sensor) * frac[:, None] + jitter
    w0 = (_u(3, n, 10) * 50).astype(np.float32)
    d0 = ((_u(3, n, 11) - 0.5) * 0.2).astype(np.float32)
    for trunc, maxw, drop, sparse in ((0.1, 1e4, 0.0, 1.0), (0.3, 20.0, 0.05, 2.5), (10.0, 1e4, 0.0, 0.0)):
        for i in range(n):
            a = [C.c_float(w0[i]), C.c_float(d0[i])]
            b = [C.c_float(w0[i]), C.c_float(d0[i])]
            args = ((C.c_double * 3)(*sensor[i]), (C.c_double * 3)(*sample[i]), (C.c_double * 3)(*centre[i]), trunc,
                    maxw, drop, sparse)
            ra = O.lib.oracle_calculate_tsdf(*args, C.byref(a[0]), C.byref(a[1]))
            rb = ref.ref_calculate_tsdf(*args, C.byref(b[0]), C.byref(b[1]))
            assert ra == rb
            assert np.float32(a[0].value).view(np.uint32) == np.float32(b[0].value).view(np.uint32)
            assert np.float32(a[1].value).view(np.uint32) == np.float32(b[1].value).view(np.uint32)


def test_gpukey_layout_and_ray_flags_fixture_is_what_the_reference_header_compiles_to():
    """The committed fixture (tests/golden/ref_vectors.npz: gpukey_layout, gpukey_bytes, ray_flags) against the live
    reference library: ohmgpu/GpuKey.h:37-46 and ohm/RayFlag.h:16-60 compiled where they lie."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))
    layout = (C.c_uint * 4)()
    ref.ref_gpukey_layout(layout)
    assert list(layout) == [int(v) for v in g["gpukey_layout"]]
    for region, voxel, expect in zip(g["gpukey_regions"], g["gpukey_voxels"], g["gpukey_bytes"]):
        buf = (C.c_ubyte * int(layout[0]))()
        ref.ref_gpukey_bytes((C.c_short * 3)(*[int(v) for v in region]), (C.c_ubyte * 4)(*[int(v) for v in voxel]), buf)
        assert bytes(buf) == expect.tobytes()
    flags = (C.c_uint * 12)()
    ref.ref_ray_flags(flags)
    assert list(flags) == [int(v) for v in g["ray_flags"]]
