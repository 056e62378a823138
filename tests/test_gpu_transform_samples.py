"""-m gpu: GpuTransformSamples equivalent (ohmhip_transform_samples) vs the oracle's fp64 restatement of the reference
kernel, and the reference test's own property (tests/ohmtestgpu/GpuTests.cpp:32-228: samples pushed into a moving sensor
frame and transformed back must land on the original points).  Positions are bit exact (plain fp64 lerp); the rotated
sample goes through acos / sin, where the device maths library and glibc may differ in the last place: 1e-12 relative."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuTransformSamples, OccupancyMap, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def quat_rotate(q, v):
    x, y, z, w = q
    m = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return m @ v


def trajectory(count, base_time, end_time):
    times = base_time + (end_time - base_time) * np.arange(count) / (count - 1)
    translations = np.array([-0.42, -0.42, -0.42]) + (np.arange(count) / (count - 1))[:, None] * 10.42
    angles = np.pi * np.arange(count) / (count - 1)
    # rotation about z by `angle`, as (x, y, z, w)
    rotations = np.stack([np.zeros(count), np.zeros(count), np.sin(angles / 2), np.cos(angles / 2)], axis=1)
    return times, translations, rotations


def test_transform_matches_oracle_and_round_trips(gpu):
    n = 20000
    idx = np.arange(n)
    global_pts = np.stack([(synth.uniform01(5, idx, s) - 0.5) * 40.0 for s in range(3)], axis=1)
    base = 1.7e9
    dt = 1e-3
    times, translations, rotations = trajectory(10, base, base + n * dt + 1.5 * dt)
    sample_times = base + 0.67 * dt + dt * idx
    # global -> local with the reference's pose rule (its test does the same on the CPU)
    local = np.zeros_like(global_pts)
    tidx = 0
    for i in range(n):
        while times[tidx + 1] < sample_times[i]:
            tidx += 1
        f = (sample_times[i] - times[tidx]) / (times[tidx + 1] - times[tidx])
        pos = translations[tidx] + f * (translations[tidx + 1] - translations[tidx])
        # pose rotation = rot[from] * slerp(rot[from], rot[to], f); both are rotations about z, so compose angles
        a0 = 2 * np.arctan2(rotations[tidx][2], rotations[tidx][3])
        a1 = 2 * np.arctan2(rotations[tidx + 1][2], rotations[tidx + 1][3])
        ang = a0 + (a0 + f * (a1 - a0))
        q = np.array([0.0, 0.0, np.sin(ang / 2), np.cos(ang / 2)])
        q_inv = q * np.array([-1, -1, -1, 1])
        local[i] = quat_rotate(q_inv, global_pts[i] - pos)
    # a few rejects: NaN component, beyond max_range (dot > max_range, as the reference compares)
    local[17, 1] = np.nan
    local[123] = [80.0, 0.0, 0.0]
    max_range = 60.0 * 60.0
    expect = O.transform_samples(times, translations, rotations, sample_times, local, max_range)
    ts = GpuTransformSamples()
    ptr, count = ts.transform(times, translations, rotations, sample_times, local, max_range)
    assert count == expect.shape[0] == 2 * (n - 2)
    got = ts.read(count)
    assert np.array_equal(got[0::2], expect[0::2])  # sensor positions: bit exact
    scale = np.maximum(np.abs(expect[1::2]), 1.0)
    assert np.max(np.abs(got[1::2] - expect[1::2]) / scale) < 1e-12
    # round trip (GpuTests.cpp:206-222 uses 1e-4 for its fp32 kernel; fp64 gets 1e-7 like the reference's CPU check)
    keep = np.ones(n, dtype=bool)
    keep[[17, 123]] = False
    assert np.max(np.linalg.norm(got[1::2] - global_pts[keep], axis=1)) < 1e-7
    # the device buffer feeds the integration path directly
    map_ = OccupancyMap(0.25)
    gm = GpuMap(map_)
    assert gm.integrateRaysDevice(ptr, count) == count
    gm.syncVoxels()
    assert len(map_.chunks) > 0
    ts.close()


def test_transform_edge_cases(gpu):
    ts = GpuTransformSamples()
    times, translations, rotations = trajectory(4, 100.0, 103.0)
    pts = np.array([[1.0, 0.0, 0.0], [0.0, 2.0, 0.0], [0.0, 0.0, 3.0]])
    # sample times before, inside (exactly on a transform stamp) and after the trajectory
    st = np.array([99.0, 101.0, 200.0])
    expect = O.transform_samples(times, translations, rotations, st, pts)
    ptr, count = ts.transform(times, translations, rotations, st, pts)
    got = ts.read(count)
    assert count == 6
    assert np.all(np.isfinite(got))
    assert np.allclose(got, expect, rtol=1e-12, atol=1e-12)
    # two transforms: no search; one transform: that pose
    for k in (2, 1):
        expect = O.transform_samples(times[:k], translations[:k], rotations[:k], st, pts)
        ptr, count = ts.transform(times[:k], translations[:k], rotations[:k], st, pts)
        assert np.allclose(ts.read(count), expect, rtol=1e-12, atol=1e-12)
    # nothing in -> nothing out
    ptr, count = ts.transform(times, translations, rotations, np.zeros(0), np.zeros((0, 3)))
    assert count == 0
    ts.close()
