"""The oracle against committed golden vectors of the REAL reference code (tests/golden/ref_vectors.npz: seeded inputs
and the outputs of the reference's own headers compiled in place -- generator tests/golden/make_ref_vectors.py).
Bit-exact agreement required.  CPU only; needs neither the reference checkout nor oracle/_ref."""
import ctypes as C
import os

import numpy as np

from oracle import oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))


def test_golden_key_quantisation():
    for ri, res in enumerate(G["coord_region_res"]):
        got = np.array([O.lib.oracle_point_to_region_coord(float(c), float(res)) for c in G["coord_in"]], dtype=np.int32)
        assert np.array_equal(got, G["coord_region"][ri])
    got = np.array([O.lib.oracle_point_to_region_voxel(float(c), 0.1, 3.2) for c in G["local_in"]], dtype=np.int32)
    assert np.array_equal(got, G["local_voxel"])


def test_golden_occupancy_adjustments():
    names = ("hit", "up", "miss", "down")
    fns = [getattr(O.lib, "oracle_occupancy_adjust_" + n) for n in names]
    inf = float("inf")
    rows = G["adjust_rows"]
    f = lambda bits: float(np.uint32(bits).view(np.float32))  # noqa: E731
    bad = 0
    for kind, v, a, limit, smin, smax, null, expect in rows:
        x = C.c_float(f(v))
        fns[int(kind)](C.byref(x), f(v), f(a), inf, f(limit), f(smin), f(smax), int(null))
        bad += int(np.float32(x.value).view(np.uint32) != np.uint32(expect))
    assert bad == 0
    assert len(rows) > 4000


def test_golden_tsdf_update():
    sensor, sample, centre = G["tsdf_sensor"], G["tsdf_sample"], G["tsdf_centre"]
    w0, d0 = G["tsdf_w0"], G["tsdf_d0"]
    for pi, (trunc, maxw, drop, sparse) in enumerate(G["tsdf_params"]):
        expect = G["tsdf_out"][pi]
        for i in range(sensor.shape[0]):
            w, d = C.c_float(w0[i]), C.c_float(d0[i])
            r = O.lib.oracle_calculate_tsdf((C.c_double * 3)(*sensor[i]), (C.c_double * 3)(*sample[i]),
                                            (C.c_double * 3)(*centre[i]), float(trunc), float(maxw), float(drop),
                                            float(sparse), C.byref(w), C.byref(d))
            got = (r, int(np.float32(w.value).view(np.uint32)), int(np.float32(d.value).view(np.uint32)))
            assert got == tuple(int(v) for v in expect[i]), (pi, i)


def test_golden_gpukey_layout():
    """the reference's device key record (ohmgpu/GpuKey.h:37-46) is 10 bytes, 2-byte aligned, region at 0,
    voxel at 6 -- taken from the compiled reference header -- and the record every mirror of this build packs / unpacks
    (numpy view in GpuMap.lineKeys, `GpuKeyOut` in the library, `ohm::GpuKey` in ohm_amd/host) is that layout, byte for
    byte on 64 seeded keys."""
    size, align, off_region, off_voxel = (int(v) for v in G["gpukey_layout"])
    assert (size, align, off_region, off_voxel) == (10, 2, 0, 6)
    record = np.dtype([("region", "<i2", (3,)), ("voxel", "u1", (4,))])
    assert record.itemsize == size and record.fields["region"][1] == off_region and record.fields["voxel"][1] == off_voxel
    packed = np.zeros(len(G["gpukey_regions"]), dtype=record)
    packed["region"] = G["gpukey_regions"]
    packed["voxel"] = G["gpukey_voxels"]
    assert np.array_equal(packed.view(np.uint8).reshape(-1, size), G["gpukey_bytes"])
    # the slicing GpuMap.lineKeys applies to the library's records
    raw = G["gpukey_bytes"]
    assert np.array_equal(raw[:, :6].copy().view(np.int16).reshape(-1, 3), G["gpukey_regions"])
    assert np.array_equal(raw[:, 6:10], G["gpukey_voxels"])
    # the library's own record and the C++ host mirror's declare the same members in the same order
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "ohm_amd", "csrc", "replay_kernels.h")).read()
    assert "static_assert(sizeof(GpuKeyOut) == 10" in src and "offsetof(GpuKeyOut, voxel) == 6" in src


def test_golden_ray_flags():
    """RayFlag bit values of the compiled reference header (ohm/RayFlag.h:16-60) against the C ABI's
    OHMHIP_RF_* macros and both host mirrors."""
    import re
    names = ["Default", "EndPointAsFree", "StopOnFirstOccupied", "ExcludeOrigin", "ExcludeSample", "ExcludeRay",
             "ExcludeUnobserved", "ExcludeFree", "ExcludeOccupied", "ReverseWalk"]
    expect = {n: int(v) for n, v in zip(names, G["ray_flags"][:10])}
    assert expect["Default"] == 0 and expect["ReverseWalk"] == 256
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "ohmhip.h")).read()
    macro = {"Default": "DEFAULT", "EndPointAsFree": "END_POINT_AS_FREE", "StopOnFirstOccupied": "STOP_ON_FIRST_OCCUPIED",
             "ExcludeOrigin": "EXCLUDE_ORIGIN", "ExcludeSample": "EXCLUDE_SAMPLE", "ExcludeRay": "EXCLUDE_RAY",
             "ExcludeUnobserved": "EXCLUDE_UNOBSERVED", "ExcludeFree": "EXCLUDE_FREE",
             "ExcludeOccupied": "EXCLUDE_OCCUPIED", "ReverseWalk": "REVERSE_WALK"}
    for name, value in expect.items():
        m = re.search(r"#define OHMHIP_RF_%s (\(1u << (\d+)\)|0u)" % macro[name], header)
        assert m, name
        got = 0 if m.group(1) == "0u" else 1 << int(m.group(2))
        assert got == value, name
    from ohm_amd.gpumap import RayFlag
    for name, value in expect.items():
        assert getattr(RayFlag, "kRf" + name) == value
    cpp = open(os.path.join(root, "ohm_amd", "host", "OhmGpuMap.h")).read()
    for name, value in expect.items():
        m = re.search(r"kRf%s\s*=\s*([^,\n]+)" % name, cpp)
        assert m, name
        text = m.group(1).strip()
        got = eval(text.replace("u", ""), {"__builtins__": {}}, dict(OHMHIP_RF_DEFAULT=0))  # "(1 << 3)" / "0"
        assert got == value, (name, text)
