"""Pin the CPU oracle (oracle/ohm_oracle.c) against the reference's OWN known-answer tests for this path.

Each test restates the checks of one reference test (cited) with the same inputs / expected values; random inputs come
from this repo's hash PRNG because std::*_distribution output is implementation defined (the reference tests accept
that variance through their tolerances).  CPU only.
"""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as O
from oracle.oracle import OracleMap
from ohm_amd import synth


# ---------------------------------------------------------------------------------------------------------------------
# tests/ohmtest/LineWalkTests.cpp:43-197 (testWalk), :200-231 (Random), :234-290 (Walk)
# ---------------------------------------------------------------------------------------------------------------------
def _global_voxel(key, dims=(32, 32, 32)):
    return np.array([key[0][a] * dims[a] + key[1][a] for a in range(3)], dtype=np.int64)


def _ray_hits_box(start, direction, lo, hi):
    tmin, tmax = -np.inf, np.inf
    for a in range(3):
        if abs(direction[a]) < 1e-300:
            if start[a] < lo[a] or start[a] > hi[a]:
                return False
            continue
        t1 = (lo[a] - start[a]) / direction[a]
        t2 = (hi[a] - start[a]) / direction[a]
        tmin = max(tmin, min(t1, t2))
        tmax = min(tmax, max(t1, t2))
    return tmax >= tmin and tmax >= 0


def check_walk(m, start, end, include_end):
    start_key = m.voxel_key(start)
    end_key = m.voxel_key(end)
    keys, enter, exit_ = m.walk(start, end, 0 if include_end else 2)
    res = m.resolution
    d = np.asarray(end) - np.asarray(start)
    direction = d / np.linalg.norm(d) if np.linalg.norm(d) > 0 else d
    last = None
    last_dist = -1.0
    for i, k in enumerate(keys):
        g = _global_voxel(k)
        to_end = _global_voxel(end_key) - g
        if i == 0:
            assert k == start_key  # first voxel is the start key and contains the start point
            last_dist = np.linalg.norm(to_end)
        else:
            assert k != start_key
            step = g - _global_voxel(last)
            assert abs(np.linalg.norm(step) - 1.0) < 1e-6  # exactly one orthogonal voxel step
            dist = np.linalg.norm(to_end)
            assert dist < last_dist  # monotonically closer to the end voxel
            last_dist = dist
        centre = np.array(m.voxel_centre(*k))
        pad = 0.5 * (res + 1e-3)
        assert _ray_hits_box(np.asarray(start), direction, centre - pad, centre + pad)
        assert exit_[i] >= enter[i] - 1e-12
        last = k
    if include_end:
        assert last == end_key
    elif start_key != end_key:
        assert abs(np.linalg.norm(_global_voxel(end_key) - _global_voxel(last)) - 1.0) < 1e-6
    else:
        assert not keys


def test_linewalk_random():
    m = OracleMap(0.1)
    n = 1000
    i = np.arange(n, dtype=np.uint64)
    pts = np.stack([2.0 * synth.uniform01(1153297050, i, s) - 1.0 for s in range(6)], axis=1)
    for origin in ((0.0, 0.0, 0.0), (0.05, 0.05, 0.05)):
        m.set_origin(origin)
        for row in pts:
            check_walk(m, row[:3], row[3:], True)
            check_walk(m, row[:3], row[3:], False)


def test_linewalk_walk_axes_and_diagonals():
    m = OracleMap(0.1)
    for origin in ((0.0, 0.0, 0.0), (0.05, 0.05, 0.05)):
        m.set_origin(origin)
        for scale in range(1, 11):
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        if dx == dy == dz == 0:
                            continue
                        end = np.array([dx, dy, dz], dtype=np.float64) * scale
                        check_walk(m, (0.0, 0.0, 0.0), end, True)
                        check_walk(m, (0.0, 0.0, 0.0), end, False)


def test_walk_visit_count_is_manhattan_plus_one():
    m = OracleMap(0.1)
    rays = synth.random_rays(500, extent=4.0, seed=3)
    for s, e in zip(rays[0::2], rays[1::2]):
        ks, ke = m.voxel_key(s), m.voxel_key(e)
        keys, _, _ = m.walk(s, e, 0)
        assert len(keys) == int(np.abs(_global_voxel(ke) - _global_voxel(ks)).sum()) + 1


# ---------------------------------------------------------------------------------------------------------------------
# tests/ohmtest/KeyTests.cpp:224-280
# ---------------------------------------------------------------------------------------------------------------------
def test_keys_conversion():
    # Keys.Conversion: centre -> key round trip over one 16^3 region at 0.25 m
    m = OracleMap(0.25, (16, 16, 16))
    for z in range(16):
        for y in range(16):
            for x in range(16):
                c = m.voxel_centre((0, 0, 0), (x, y, z))
                assert m.voxel_key(c) == ((0, 0, 0), (x, y, z))


def test_keys_quantisation():
    # Keys.Quantisation: coordinate a hair below the upper boundary of region -1 must not index voxel == region size
    region_size, resolution = 32, 0.4
    bad = region_size * resolution * -0.5 - 1e-15
    r = O.lib.oracle_point_to_region_coord(bad, region_size * resolution)
    rmin = r * (region_size * resolution) - 0.5 * region_size * resolution
    v = O.lib.oracle_point_to_region_voxel(bad - rmin, resolution, region_size * resolution)
    assert v < region_size


def test_keys_indexing_central_region():
    # Keys.Indexing: points inside +-half a region map to region 0, outside to a neighbour (resolution 0.3)
    m = OracleMap(0.3)
    half = 0.5 * 32 * 0.3
    i = np.arange(2000, dtype=np.uint64)
    pts = np.stack([(2.0 * synth.uniform01(77, i, s) - 1.0) for s in range(3)], axis=1)
    for p in pts * (half - 1e-3):
        assert m.voxel_key(p)[0] == (0, 0, 0)
    for p in pts:
        q = p.copy()
        a = int(np.argmax(np.abs(q)))
        q[a] = math.copysign(half + 0.3 * (1 + abs(q[a])), q[a])
        assert m.voxel_key(q)[0] != (0, 0, 0)


# ---------------------------------------------------------------------------------------------------------------------
# tests/ohmtest/MapTests.cpp:34-79: one hit == hitValue(), one miss == missValue()
# ---------------------------------------------------------------------------------------------------------------------
def test_map_hit_and_miss_values():
    m = OracleMap(0.25)
    ray = np.array([[0.1, 0.1, 0.1], [0.9, 0.1, 0.1]])
    m.integrate_occupancy(ray)
    chunks = m.chunks()
    occ = chunks[(0, 0, 0)]["occupancy"]
    ks = m.voxel_key(ray[0])[1]
    ke = m.voxel_key(ray[1])[1]
    hit = occ[ke[0] + 32 * ke[1] + 1024 * ke[2]]
    miss = occ[ks[0] + 32 * ks[1] + 1024 * ks[2]]
    assert hit == np.float32(m.hit_value()) and hit > 0
    assert miss == np.float32(m.miss_value()) and miss < 0
    # default probabilities 0.9 / 0.45 (ohm/OccupancyMap.cpp:211-212)
    assert abs(hit - math.log(0.9 / 0.1)) < 1e-6 and abs(miss - math.log(0.45 / 0.55)) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# tests/ohmtest/NdtTests.cpp + tests/ohmtestcommon/CovarianceTestUtil.cpp:43-117
# ---------------------------------------------------------------------------------------------------------------------
def _independent_update_hit(cov, mean, count, sample):
    """numpy restatement of ohmtestutil::updateHit (the reference's independent test oracle)."""
    num_pt = float(count)
    inv = 1.0 / (num_pt + 1.0)
    diff = sample - mean
    sc1 = math.sqrt(num_pt * inv) if num_pt else 1.0
    sc2 = inv * math.sqrt(num_pt)
    A = np.zeros(9)
    A[:6] = sc1 * cov
    A[6:] = sc2 * diff
    first = (0, 1, 3)

    def pdot(j, k):
        d = A[6 + k] * A[6 + j]
        for i in range(min(j, k) + 1):
            d += A[first[j] + i] * A[first[k] + i]
        return d
    out = cov.copy()
    for k in range(3):
        ind1 = (k * (k + 3)) >> 1
        indk = ind1 - k
        ak = math.sqrt(pdot(k, k))
        out[ind1] = np.float32(ak)
        if ak > 0:
            aki = 1.0 / ak
            for j in range(k + 1, 3):
                indj = (j * (j + 1)) >> 1
                c = pdot(j, k) * aki
                out[indj + k] = np.float32(c)
                c *= aki
                A[j + 6] -= c * A[k + 6]
                for l in range(k + 1):
                    A[indj + l] -= c * A[indk + l]
    return out, (num_pt * mean + sample) * inv, count + 1


def _ndt_samples_gaussian(n, seed):
    i = np.arange(n, dtype=np.uint64)
    u = [np.maximum(synth.uniform01(seed, i, s), 1e-300) for s in range(6)]
    g = [np.sqrt(-2 * np.log(u[2 * a])) * np.cos(2 * np.pi * u[2 * a + 1]) for a in range(3)]
    L = np.array([[0.25, 0, 0], [0.08, 0.15, 0], [-0.05, 0.04, 0.1]])
    pts = np.stack(g, axis=1) @ L.T + np.array([1.0, 1.0, 1.0])
    return np.clip(pts, 0.02, 1.98)


def test_ndt_hit_matches_independent_reference():
    # Ndt.Hit: 10000 samples in one 2 m voxel; covariance within 1e-2, mean within 1e-1 of the reference oracle.
    res = 2.0
    samples = _ndt_samples_gaussian(10000, 1153297050)
    m = OracleMap(res, layers=("occupancy", "mean", "covariance"))
    m.set_ndt()
    rays = np.empty((2 * len(samples), 3))
    rays[0::2] = 0.0
    rays[1::2] = samples
    m.integrate_ndt(rays, flags=1 << 4)  # kRfExcludeRay: samples only, as integrateNdtHit does
    ch = m.chunks()[(0, 0, 0)]
    vi = 16 + 32 * 16 + 1024 * 16
    cov = ch["covariance"].reshape(-1, 6)[vi]
    coord, count = ch["mean"].reshape(-1, 2)[vi]
    local = (C.c_double * 3)()
    O.lib.oracle_sub_voxel_to_local(int(coord), res, local)
    mean = np.array(local) + np.array(m.voxel_centre((0, 0, 0), (16, 16, 16)))
    # independent reference run
    rcov = np.zeros(6, dtype=np.float64)
    rmean = np.zeros(3)
    rcount = 0
    for s in samples:
        if rcount == 0:
            rcov = np.array([0.1 * res, 0, 0.1 * res, 0, 0, 0.1 * res])  # initialiseCovariance
            rmean = s.copy()  # first sample: sample_to_mean == 0
        rcov, rmean, rcount = _independent_update_hit(rcov, rmean if rcount else s, rcount, s)
    assert int(count) == rcount == 10000
    assert np.all(np.abs(cov - rcov) < 1e-2)
    assert np.linalg.norm(mean - rmean) < 1e-1
    # and against the population covariance: P = C C^T
    Cm = np.array([[cov[0], 0, 0], [cov[1], cov[2], 0], [cov[3], cov[4], cov[5]]], dtype=np.float64)
    pop = np.cov(samples.T, bias=True)
    assert np.all(np.abs(Cm @ Cm.T - pop) < 2e-3)


def test_ndt_hit_step_by_step_against_the_independent_reference():
    """the reference's own Ndt.Hit holds only the END of 10 000 updates to 1e-2 / 1e-1.  Here every
    single step of the oracle's calculateHitWithCovariance (ohm/CovarianceVoxelCompute.h:301-375) is compared with
    ohmtestutil::updateHit (tests/ohmtestcommon/CovarianceTestUtil.cpp:43-117, restated above) on the SAME input state
    -- packed factor, mean, count, sample -- over 10 000 samples: the factor's six terms to 1e-6 relative.  Two
    trajectories: the state carried by the oracle (what the mapper does) and by the reference (so a drift of one cannot
    hide in the other's state)."""
    res = 2.0
    samples = _ndt_samples_gaussian(10000, 1153297050)
    worst = 0.0
    for carrier in ("oracle", "reference"):
        cov = np.zeros(6, dtype=np.float32)
        mean = np.zeros(3)
        count = 0
        for s in samples:
            # the reference's harness initialises the factor before the first update (initialiseTestVoxel / first-sample
            # rule of the mapper: covariance reset, sample_to_mean = 0)
            if count == 0:
                ref_in = np.array([0.1 * res, 0, 0.1 * res, 0, 0, 0.1 * res])
                ref_mean_in = s.copy()
            else:
                ref_in = cov.astype(np.float64)
                ref_mean_in = mean
            ref_cov, ref_mean, _ = _independent_update_hit(ref_in, ref_mean_in, count, s)
            ocov = (C.c_float * 6)(*cov)
            value = C.c_float(0.5)
            reset = O.lib.oracle_calculate_hit_with_covariance(ocov, C.byref(value), (C.c_double * 3)(*s),
                                                               (C.c_double * 3)(*mean), count, 0.1, float("inf"),
                                                               res, -1e30, 1 << 30)
            assert bool(reset) == (count == 0)
            got = np.array(list(ocov), dtype=np.float64)
            ref32 = ref_cov.astype(np.float32).astype(np.float64)
            err = np.abs(got - ref32) / np.maximum(np.abs(ref32), 1e-3)
            worst = max(worst, float(err.max()))
            assert np.all(err <= 1e-6), (carrier, count, got, ref32)
            cov = np.array(list(ocov), dtype=np.float32) if carrier == "oracle" else ref_cov.astype(np.float32)
            mean = ref_mean if count else s.copy()
            count += 1
    assert worst <= 1e-6


def _build_ndt_voxel(samples, res, origin, sensor_noise):
    m = OracleMap(res, layers=("occupancy", "mean", "covariance"))
    m.set_origin(origin)
    O.lib.oracle_map_set_hit_probability(m.handle, 0.55)
    O.lib.oracle_map_set_miss_probability(m.handle, 0.45)
    m.set_ndt(sensor_noise=sensor_noise, adaptation_rate=1.0)
    rays = np.empty((2 * len(samples), 3))
    rays[0::2] = np.array([0.0, 0.0, 5.0])
    rays[1::2] = samples
    m.integrate_ndt(rays, flags=1 << 4)
    key = m.voxel_key(samples[0])
    ch = m.chunks()[key[0]]
    vi = key[1][0] + 32 * key[1][1] + 1024 * key[1][2]
    cov = np.ascontiguousarray(ch["covariance"].reshape(-1, 6)[vi])
    coord, count = ch["mean"].reshape(-1, 2)[vi]
    assert int(count) == len(samples), "all samples must fall in one voxel"
    local = (C.c_double * 3)()
    O.lib.oracle_sub_voxel_to_local(int(coord), res, local)
    mean = np.array(local) + np.array(m.voxel_centre(*key))
    return m, cov, mean, int(count), float(ch["occupancy"][vi])


def _miss_probability(m, cov, mean, count, value, start, end, sensor_noise):
    v = C.c_float(value)
    is_miss = C.c_int(0)
    covc = (C.c_float * 6)(*cov)
    O.lib.oracle_calculate_miss_ndt(covc, C.byref(v), C.byref(is_miss), (C.c_double * 3)(*start),
                                    (C.c_double * 3)(*end), (C.c_double * 3)(*mean), count, float("inf"),
                                    m.miss_value(), 1.0, sensor_noise, 3)
    # integrateNdtMiss then clamps through occupancyAdjustDown (ohm/CovarianceVoxel.cpp, min value -2): the
    # reference's expected probabilities are for the CLAMPED adjustment.
    occ = C.c_float(value)
    O.lib.oracle_occupancy_adjust_down(C.byref(occ), value, v.value, float("inf"), -2.0, -3.4028234663852886e38,
                                       3.4028234663852886e38, 0)
    return float(O.lib.oracle_value_to_probability(np.float32(occ.value) - np.float32(value)))


def test_ndt_miss_planar():
    # Ndt.MissPlanar (NdtTests.cpp:268-330): plane z = 1 in a 2 m voxel; expected probabilities from the reference.
    i = np.arange(10000, dtype=np.uint64)
    samples = np.stack([0.01 + 1.98 * synth.uniform01(1153297050, i, 0), 0.01 + 1.98 * synth.uniform01(1153297050, i, 1),
                        np.ones(10000)], axis=1)
    m, cov, mean, count, value = _build_ndt_voxel(samples, 2.0, (0, 0, 0), 0.05)
    cases = [((1, 1, 5), (1, 1, -5), 0.004, 0.001), ((1, 1, -5), (1, 1, 5), 0.004, 0.001),
             ((-5, 1, 0.25), (5, 1, 0.25), 0.5, 0.001), ((1, 5, 1.01), (1, -5, 1.01), 0.5, 0.001),
             ((-5, 1, 2), (5, 1, 1), 0.5, 0.001), ((-5, 1, 2), (5, 1, 0.5), 0.23, 0.02)]
    for start, end, expect, tol in cases:
        p = _miss_probability(m, cov, mean, count, value, start, end, 0.05)
        assert abs(p - expect) <= tol, (start, end, p, expect)


def _cylinder_samples(n=10000, radius=0.3, seed=1153297050):
    """The sample cloud of Ndt.MissCylindrical (NdtTests.cpp:343-373), drawn from the reference's own random stream
    (tests/stdrandom.py): uniform points of the voxel pushed onto a cylinder about the z axis, radius uniform in
    [radius - noise, radius + noise] with noise = 0.05f."""
    from stdrandom import MinStdRand0
    rng = MinStdRand0(seed)
    noise = float(np.float32(0.05))
    pts = np.zeros((n, 3))
    for i in range(n):
        x, y, z = rng.uniform(-0.99, 0.99), rng.uniform(-0.99, 0.99), rng.uniform(-0.99, 0.99)
        length_xy = math.sqrt(x * x + y * y)
        if length_xy > 1e-6:
            r = rng.uniform(radius - noise, radius + noise)
            x, y = r * x / length_xy, r * y / length_xy
        pts[i] = (x, y, z)
    return pts


# (start, end, expected probability, tolerance) as the reference lists them, NdtTests.cpp:375-405
NDT_MISS_CYLINDRICAL_CASES = [
    ((0, 0, 5), (0, 0, -5), 0.004, 0.001),          # straight down the axis
    ((0, 0, -5), (0, 0, 5), 0.004, 0.001),          # the same reversed
    ((0.3, 0.3, 5), (0.3, 0.3, -5), 0.425, 0.01),   # parallel to the cylinder near its edge: a near hit
    ((0.45, 0.45, -5), (0.6, 0.6, 5), 0.499, 0.001),  # parallel, outside: a miss
    ((2, -0.3, 0), (-2, -0.3, 0), 0.312, 0.005),    # across the middle of the cylinder
    ((2, -0.3, 0.85), (-2, -0.3, 0.85), 0.436, 0.003),  # across its top end
    ((2, 0.6, 0.85), (-2, 0.6, 0.85), 0.497, 0.001),    # across the voxel, past the cylinder
]


def test_ndt_miss_cylindrical():
    # Ndt.MissCylindrical (NdtTests.cpp:335-408): the one reference case with curvature about a single axis, on the
    # reference's own 10 000 samples and to the reference's own tolerances.
    samples = _cylinder_samples()
    m, cov, mean, count, value = _build_ndt_voxel(samples, 2.0, (-1.0, -1.0, -1.0), 0.05)
    for start, end, expect, tol in NDT_MISS_CYLINDRICAL_CASES:
        p = _miss_probability(m, cov, mean, count, value, start, end, 0.05)
        assert abs(p - expect) <= tol, (start, end, p, expect)


def test_ndt_miss_spherical():
    # Ndt.MissSpherical (NdtTests.cpp:407-470): shell of radius 0.3 +- noise around the origin
    n = 10000
    i = np.arange(n, dtype=np.uint64)
    d = np.stack([-0.99 + 1.98 * synth.uniform01(1153297050, i, s) for s in range(3)], axis=1)
    d = d / np.linalg.norm(d, axis=1)[:, None]
    r = 0.25 + 0.1 * synth.uniform01(1153297050, i, 3)
    samples = d * r[:, None]
    m, cov, mean, count, value = _build_ndt_voxel(samples, 2.0, (-1.0, -1.0, -1.0), 0.05)
    R = 0.3
    cases = [((0, 0, 5), (0, 0, -5), 0.004, 0.001), ((0, 0, -5), (0, 0, 5), 0.004, 0.001),
             ((R, R, 5), (R, R, -5), 0.469, 0.006), ((1.5 * R, 1.5 * R, -5), (2 * R, 2 * R, 5), 0.5, 0.001)]
    for start, end, expect, tol in cases:
        p = _miss_probability(m, cov, mean, count, value, start, end, 0.05)
        assert abs(p - expect) <= tol, (start, end, p, expect)


# ---------------------------------------------------------------------------------------------------------------------
# tests/ohmtest/TsdfTests.cpp:18-137
# ---------------------------------------------------------------------------------------------------------------------
TSDF_DIRS = [(1, 0, 0), (-1, 0, 0), (1, 1, 0), (-1, 1, 0), (1, 0, 0), (1, -1, 0), (-1, 0, 1), (1, 1, 1), (-1, 1, 1),
             (1, 0, 1), (1, -1, 1), (-1, 0, 1), (1, 1, -1), (-1, 1, -1), (1, 0, -1), (1, -1, -1)]


def _compute_distance(sensor, sample, centre):
    sv = np.asarray(centre) - np.asarray(sensor)
    ss = np.asarray(sample) - np.asarray(sensor)
    g = np.float32(math.sqrt(float(ss @ ss)))
    return np.float32(g - np.float32(np.float32(float(sv @ ss)) / g))


@pytest.mark.parametrize("trunc,passes", [(10.0, 1), (0.1, 2)])
def test_tsdf_basic_and_truncation(trunc, passes):
    for d in TSDF_DIRS:
        m = OracleMap(0.1, layers=("tsdf",))
        m.set_origin((-0.05, -0.05, -0.05))
        m.set_tsdf(trunc=trunc)
        ray = np.array([[0.0, 0.0, 0.0], d], dtype=np.float64)
        for _ in range(passes):
            m.integrate_tsdf(ray)
        ch = m.chunks()
        keys, _, _ = m.walk(ray[0], ray[1], 0)
        for k in keys:
            tsdf = ch[k[0]]["tsdf"].reshape(-1, 2)[k[1][0] + 32 * k[1][1] + 1024 * k[1][2]]
            expect = min(np.float32(trunc), _compute_distance(ray[0], ray[1], m.voxel_centre(*k)))
            assert abs(tsdf[1] - expect) < 1e-6
            assert tsdf[0] == passes


# ---------------------------------------------------------------------------------------------------------------------
# tests/ohmtestcommon/TraversalTest.cpp:22-188 (exact path lengths through the voxel at the origin)
# ---------------------------------------------------------------------------------------------------------------------
def _signs():
    ortho = [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    d2 = [(-1, -1, 0), (1, -1, 0), (-1, 1, 0), (1, 1, 0), (-1, 0, -1), (1, 0, -1), (-1, 0, 1), (1, 0, 1), (0, -1, -1),
          (0, 1, -1), (0, -1, 1), (0, 1, 1)]
    d3 = [(-1, -1, -1), (1, -1, -1), (-1, 1, -1), (1, 1, -1), (-1, -1, 1), (1, -1, 1), (-1, 1, 1), (1, 1, 1)]
    return ortho, d2, d3


def test_traversal_into_origin_voxel():
    res = 0.1
    m = OracleMap(res, layers=("occupancy", "traversal"))
    m.set_origin((-0.05, -0.05, -0.05))
    ortho, d2, d3 = _signs()
    expected = 0.0
    for dirs, rate in ((ortho, 0.5 * res), (d2, 0.5 * math.sqrt(2) * res), (d3, 0.5 * math.sqrt(3) * res)):
        for d in dirs:
            m.integrate_occupancy(np.array([d, (0, 0, 0)], dtype=np.float64))
            expected += np.float32(rate)
            key = m.voxel_key((0, 0, 0))
            trav = m.chunks()[key[0]]["traversal"][key[1][0] + 32 * key[1][1] + 1024 * key[1][2]]
            assert abs(trav - expected) < 1e-3


def test_traversal_through_origin_voxel():
    res = 0.1
    m = OracleMap(res, layers=("occupancy", "traversal"))
    m.set_origin((-0.05, -0.05, -0.05))
    pairs = [((-1, 0, 0), (1, 0, 0), 1.0), ((0, -1, 0), (0, 1, 0), 1.0), ((0, 0, -1), (0, 0, 1), 1.0),
             ((-1, -1, 0), (1, 1, 0), math.sqrt(2)), ((1, -1, 0), (-1, 1, 0), math.sqrt(2)),
             ((-1, 0, -1), (1, 0, 1), math.sqrt(2)), ((0, -1, -1), (0, 1, 1), math.sqrt(2)),
             ((-1, -1, -1), (1, 1, 1), math.sqrt(3)), ((1, -1, -1), (-1, 1, 1), math.sqrt(3))]
    expected = 0.0
    for s, e, rate in pairs:
        m.integrate_occupancy(np.array([s, e], dtype=np.float64))
        expected += np.float32(rate * res)
        key = m.voxel_key((0, 0, 0))
        trav = m.chunks()[key[0]]["traversal"][key[1][0] + 32 * key[1][1] + 1024 * key[1][2]]
        assert abs(trav - expected) < 1e-3


def test_voxel_mean_round_trip():
    # VoxelMean tests: a single sample's mean decodes to within one quantum (res / 1023) of the sample offset.
    res = 0.1
    for off in ((0.0, 0.0, 0.0), (0.03, -0.02, 0.049), (-0.05, 0.05, 0.0)):
        v = (C.c_double * 3)(*off)
        pat = O.lib.oracle_sub_voxel_update(0, 0, v, res)
        out = (C.c_double * 3)()
        O.lib.oracle_sub_voxel_to_local(pat, res, out)
        assert pat & (1 << 31)
        assert np.all(np.abs(np.array(out) - np.array(off)) <= res / 1023 + 1e-12)


def test_transform_samples_round_trip():
    # tests/ohmtestgpu/GpuTests.cpp:32-228 (the CPU half of the reference's own test): samples moved into a moving
    # sensor frame with pose = lerp(translation), rot[from] * slerp(rot[from], rot[to], f) and transformed back by the
    # function under test must land on the original points to 1e-7.
    from ohm_amd import synth
    n = 5000
    idx = np.arange(n)
    global_pts = np.stack([(synth.uniform01(9, idx, s) - 0.5) * 30.0 for s in range(3)], axis=1)
    count = 10
    base, dt = 1.7e9, 1e-3
    times = base + (n * dt + 1.5 * dt) * np.arange(count) / (count - 1)
    translations = np.array([-0.42] * 3) + (np.arange(count) / (count - 1))[:, None] * 10.42
    angles = np.pi * np.arange(count) / (count - 1)
    rotations = np.stack([np.zeros(count), np.zeros(count), np.sin(angles / 2), np.cos(angles / 2)], axis=1)
    sample_times = base + 0.67 * dt + dt * idx
    local = np.zeros_like(global_pts)
    tidx = 0
    for i in range(n):
        while times[tidx + 1] < sample_times[i]:
            tidx += 1
        f = (sample_times[i] - times[tidx]) / (times[tidx + 1] - times[tidx])
        pos = translations[tidx] + f * (translations[tidx + 1] - translations[tidx])
        ang = angles[tidx] + (angles[tidx] + f * (angles[tidx + 1] - angles[tidx]))
        c, s = np.cos(-ang), np.sin(-ang)
        d = global_pts[i] - pos
        local[i] = [c * d[0] - s * d[1], s * d[0] + c * d[1], d[2]]
    rays = O.transform_samples(times, translations, rotations, sample_times, local)
    assert rays.shape == (2 * n, 3)
    assert np.max(np.linalg.norm(rays[1::2] - global_pts, axis=1)) < 1e-7
    # rejected samples are skipped, order kept (GpuTransformSamples.cpp:131-142)
    local[3, 0] = np.nan
    local[7] = [100.0, 0.0, 0.0]
    rays2 = O.transform_samples(times, translations, rotations, sample_times, local, max_range=50.0 * 50.0)
    assert rays2.shape == (2 * (n - 2), 3)
    keep = np.ones(n, dtype=bool)
    keep[[3, 7]] = False
    assert np.array_equal(rays2[1::2], rays[1::2][keep])


# ---------------------------------------------------------------------------------------------------------------------
# Hand-derived exact pins (round 2): values worked out on paper from the reference's rules, with inputs chosen so every
# intermediate is exact in binary floating point -- no tolerance anywhere.
# ---------------------------------------------------------------------------------------------------------------------
def _global_seq(m, start, end, flags=0):
    keys, _, _ = m.walk(start, end, flags)
    return [tuple(int(v) for v in _global_voxel(k)) for k in keys]


def test_walk_tie_breaking_exact_key_sequences():
    """walkSelectNextAxis (ohm/LineWalkCompute.h:282-289): `axis = (t[axis] < t[1]) ? axis : 1; axis = (t[axis] < t[2]) ?
    axis : 2` -- equal times go to the HIGHER axis.  1 m voxels with the map origin at 0 put every voxel face on an
    integer, so a ray from a voxel centre along a diagonal crosses exact corners; its direction components are bitwise
    equal, hence so are the competing times.  Voxel g covers [g, g + 1) (region 0 spans [-16, 16))."""
    m = OracleMap(1.0)
    off = 16  # global voxel index of the voxel [0, 1)

    def seq(*pts):
        return [(x + off, y + off, z + off) for x, y, z in pts]

    # x-y diagonal: at every corner y steps before x
    assert _global_seq(m, (0.5, 0.5, 0.5), (3.5, 3.5, 0.5)) == seq((0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 2, 0), (2, 2, 0),
                                                                   (2, 3, 0), (3, 3, 0))
    # space diagonal: z, then y, then x
    assert _global_seq(m, (0.5, 0.5, 0.5), (2.5, 2.5, 2.5)) == seq((0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 1, 2),
                                                                   (1, 2, 2), (2, 2, 2))
    # the rule is about the axis index, not the direction of travel
    assert _global_seq(m, (0.5, 0.5, 0.5), (-2.5, -2.5, 0.5)) == seq((0, 0, 0), (0, -1, 0), (-1, -1, 0), (-1, -2, 0),
                                                                     (-2, -2, 0), (-2, -3, 0), (-3, -3, 0))
    assert _global_seq(m, (0.5, 0.5, 0.5), (2.5, 0.5, -1.5)) == seq((0, 0, 0), (0, 0, -1), (1, 0, -1), (1, 0, -2),
                                                                    (2, 0, -2))
    # x-z tie with y fixed: z first
    assert _global_seq(m, (0.5, 0.5, 0.5), (2.5, 0.5, 2.5)) == seq((0, 0, 0), (0, 0, 1), (1, 0, 1), (1, 0, 2), (2, 0, 2))
    # kExcludeEndVoxel (2) drops exactly the last voxel, kExcludeStartVoxel (1) the first
    assert _global_seq(m, (0.5, 0.5, 0.5), (2.5, 2.5, 2.5), 2) == seq((0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1),
                                                                      (1, 1, 2), (1, 2, 2))
    assert _global_seq(m, (0.5, 0.5, 0.5), (2.5, 2.5, 2.5), 1)[0] == (off, off, off + 1)
    # a 2:1 slope (no ties): x crosses at t = 0.5, 1.5, 2.5 (in units of the x extent), y at 1, 3 -> x, y, x, x, y, x...
    assert _global_seq(m, (0.5, 0.5, 0.5), (4.5, 2.5, 0.5)) == seq((0, 0, 0), (1, 0, 0), (1, 1, 0), (2, 1, 0), (3, 1, 0),
                                                                   (3, 2, 0), (4, 2, 0))
    # and across a region boundary (region 0 ends at voxel 15): the sequence is unaffected by the region split
    assert _global_seq(m, (14.5, 14.5, 0.5), (17.5, 17.5, 0.5)) == seq((14, 14, 0), (14, 15, 0), (15, 15, 0),
                                                                       (15, 16, 0), (16, 16, 0), (16, 17, 0), (17, 17, 0))
    keys, _, _ = m.walk((14.5, 14.5, 0.5), (17.5, 17.5, 0.5), 0)
    assert keys[2] == ((0, 0, 0), (31, 31, 16)) and keys[3] == ((0, 1, 0), (31, 0, 16)) and keys[4] == ((1, 1, 0), (0, 0, 16))


def test_walk_axis_aligned_exact_sequences_and_ranges():
    m = OracleMap(1.0)
    keys, enter, exit_ = m.walk((0.5, 0.5, 0.5), (3.5, 0.5, 0.5), 0)
    assert [tuple(int(v) for v in _global_voxel(k)) for k in keys] == [(16 + i, 16, 16) for i in range(4)]
    assert enter == [0.0, 0.5, 1.5, 2.5] and exit_ == [0.5, 1.5, 2.5, 3.0]  # the end voxel's exit is the ray length
    keys, enter, exit_ = m.walk((0.5, 0.5, 0.5), (0.5, 0.5, -1.5), 0)
    assert [tuple(int(v) for v in _global_voxel(k)) for k in keys] == [(16, 16, 16), (16, 16, 15), (16, 16, 14)]
    assert enter == [0.0, 0.5, 1.5] and exit_ == [0.5, 1.5, 2.0]


def test_voxel_mean_exact_patterns():
    """subVoxelCoord / subVoxelUpdate (ohm/VoxelMeanCompute.h:69-92, 134-152) with a 1023 m voxel: the mean grid step is
    exactly 1 m and the offset exactly 511.5 m, so positions are plain integers: pos = floor(v + 511.5 + 0.5)."""
    res = 1023.0
    used = 1 << 31

    def pattern(px, py, pz):
        return used | (pz << 20) | (py << 10) | px

    v = (C.c_double * 3)(-511.5, 0.0, 511.5)
    assert O.lib.oracle_sub_voxel_coord(v, res) == pattern(0, 512, 1023) == 0xBFF80000
    v = (C.c_double * 3)(0.25, -100.75, 300.49)
    assert O.lib.oracle_sub_voxel_coord(v, res) == pattern(512, 411, 812)
    v = (C.c_double * 3)(-600.0, 600.0, 0.0)  # outside the voxel: clamped to the grid's ends
    assert O.lib.oracle_sub_voxel_coord(v, res) == pattern(0, 1023, 512)
    # first sample of a voxel: the update IS the sample (count 0 -> mean + (v - mean) / 1)
    v = (C.c_double * 3)(0.5, -100.5, 300.5)
    first = O.lib.oracle_sub_voxel_update(0, 0, v, res)
    assert first == pattern(512, 411, 812)
    # second sample: decoded mean (0.5, -100.5, 300.5) moves half way to (10.5, -90.5, 290.5) = (5.5, -95.5, 295.5)
    v = (C.c_double * 3)(10.5, -90.5, 290.5)
    assert O.lib.oracle_sub_voxel_update(first, 1, v, res) == pattern(517, 416, 807)
    # third sample with count 2: mean (5.5, -95.5, 295.5) + ((35.5, -95.5, 265.5) - mean) / 3 = (15.5, -95.5, 285.5)
    v = (C.c_double * 3)(35.5, -95.5, 265.5)
    assert O.lib.oracle_sub_voxel_update(pattern(517, 416, 807), 2, v, res) == pattern(527, 416, 797)
    out = (C.c_double * 3)()
    O.lib.oracle_sub_voxel_to_local(pattern(527, 416, 797), res, out)
    assert tuple(out) == (15.5, -95.5, 285.5)


def test_ndt_hit_satisfies_the_covariance_recursion_to_float_rounding():
    """calculateHitWithCovariance (ohm/CovarianceVoxelCompute.h:301-375) implements, through a square-root factor,
        P_new = n / (n + 1) * P + n / (n + 1)^2 * (z - mu)(z - mu)^T          (its own comment, :323-331)
    Checked here as an identity in float64 on the oracle's single-step output -- C_new C_new^T against the formula
    applied to C C^T -- to 1e-6 relative (the factor is stored in float32), for random factors, means, samples and
    counts; the reference's own test only holds the end result of 10 000 steps to 1e-2."""
    rng = np.random.default_rng(1153297050)
    worst = 0.0
    for trial in range(400):
        L_ = np.tril(rng.uniform(-0.5, 0.5, (3, 3)))
        L_[np.diag_indices(3)] = rng.uniform(0.05, 0.8, 3)
        cov = np.array([L_[0, 0], L_[1, 0], L_[1, 1], L_[2, 0], L_[2, 1], L_[2, 2]], dtype=np.float32)
        n = int(rng.integers(1, 5000))
        mean = rng.uniform(-1, 1, 3)
        z = mean + rng.normal(0, 0.3, 3)
        Cm = np.array([[cov[0], 0, 0], [cov[1], cov[2], 0], [cov[3], cov[4], cov[5]]], dtype=np.float64)
        P = Cm @ Cm.T
        d = (z - mean).reshape(3, 1)
        expected = n / (n + 1.0) * P + n / (n + 1.0) ** 2 * (d @ d.T)
        covc = (C.c_float * 6)(*cov)
        value = C.c_float(1.0)
        reinit = O.lib.oracle_calculate_hit_with_covariance(covc, C.byref(value), (C.c_double * 3)(*z),
                                                            (C.c_double * 3)(*mean), n, 0.4, float("inf"), 2.0, -1.386,
                                                            100)
        assert reinit == 0 and abs(value.value - 1.4) < 1e-6
        out = np.array(list(covc), dtype=np.float64)
        Cn = np.array([[out[0], 0, 0], [out[1], out[2], 0], [out[3], out[4], out[5]]])
        assert out[0] > 0 and out[2] > 0 and out[5] > 0
        got = Cn @ Cn.T
        worst = max(worst, float(np.max(np.abs(got - expected)) / np.max(np.abs(expected))))
    assert worst < 1e-6, worst
    # (re)initialisation: count 0 starts from 0.1 * resolution * I and a zero sample-to-mean (:309-316, :333-341; :90-98)
    covc = (C.c_float * 6)(*([9.0] * 6))
    value = C.c_float(float("inf"))
    reinit = O.lib.oracle_calculate_hit_with_covariance(covc, C.byref(value), (C.c_double * 3)(0.3, 0.2, 0.1),
                                                        (C.c_double * 3)(0, 0, 0), 0, 0.4, float("inf"), 2.0, -1.386, 100)
    assert reinit == 1 and value.value == np.float32(0.4)
    # n = 0: unpackCovariance scales the fresh factor by 1, not sqrt(n / (n + 1)) (:157), and the sample-to-mean term by
    # 0: the first sample leaves the initial factor 0.1 * resolution * I exactly
    d = float(np.float32(0.1 * 2.0))
    assert list(covc) == [d, 0.0, d, 0.0, 0.0, d]
