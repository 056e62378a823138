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
