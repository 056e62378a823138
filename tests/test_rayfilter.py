"""CPU: the stock RayFilterFunctions restated in ohm_amd/rayfilter.py (ohm/RayFilter.cpp, ohm/Aabb.h) and the oracle's
handling of caller-filtered batches, pinned on the reference's own GpuMap.ClipBox expectations
(tests/ohmtestgpu/GpuMapTest.cpp:633-760): rays through the box leave only free voxels, all inside the box; rays that
end in the box leave their sample voxel occupied."""
import numpy as np

from ohm_amd import rayfilter as RF
from oracle.oracle import OracleMap


def _apply(filt, rays):
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 3)
    keep, starts, ends, flags = filt(rays[0::2].copy(), rays[1::2].copy())
    out = np.empty((2 * int(keep.sum()), 3))
    out[0::2] = starts[keep]
    out[1::2] = ends[keep]
    return out, flags[keep]


def test_clip_bounded_matches_reference_clipbox_cases():
    box = RF.Aabb((-1.0, -1.0, -1.0), (2.0, 2.0, 2.0))
    through = [(-2, 0, 0), (3, 0, 0), (0, -2, 0), (0, 3, 0), (0, 0, 3), (0, 0, -2)]
    rays, flags = _apply(RF.clip_bounded(box), through)
    assert np.array_equal(rays, [(-1, 0, 0), (2, 0, 0), (0, -1, 0), (0, 2, 0), (0, 0, 2), (0, 0, -1)])
    assert list(flags) == [RF.kRffClippedStart | RF.kRffClippedEnd] * 3
    into = [(-2, 0, 0), (0, 0, 0), (0, -2, 0), (0, 0, 0), (0, 0, 3), (0, 0, 0)]
    rays, flags = _apply(RF.clip_bounded(box), into)
    assert np.array_equal(rays[1::2], np.zeros((3, 3))) and list(flags) == [RF.kRffClippedStart] * 3
    # leaving the box: only the end moves; fully inside: untouched; degenerate: untouched
    rays, flags = _apply(RF.clip_bounded(box), [(0, 0, 0), (0, 5, 0), (0.5, 0.5, 0.5), (1, 1, 1), (0, 0, 0), (0, 0, 1e-6)])
    assert np.array_equal(rays[1], (0, 2, 0)) and list(flags) == [RF.kRffClippedEnd, 0, 0]
    # a ray which misses the box is not clipped and therefore kept as it is (ohm/RayFilter.cpp:64-70)
    rays, flags = _apply(RF.clip_bounded(box), [(5, 5, 5), (6, 6, 7)])
    assert np.array_equal(rays, [(5, 5, 5), (6, 6, 7)]) and list(flags) == [0]
    # diagonal through a corner region: entry and exit on different faces
    rays, flags = _apply(RF.clip_bounded(box), [(-3, -2, 0.5), (5, 4, 0.5)])
    assert np.allclose(rays, [(-1, -0.5, 0.5), (2, 1.75, 0.5)]) and flags[0] == (RF.kRffClippedStart | RF.kRffClippedEnd)


def test_clip_to_bounds_and_length_filters():
    box = RF.Aabb((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))
    rays, flags = _apply(RF.clip_to_bounds(box), [(5, 0, 0), (0.5, 0, 0), (5, 0, 0), (3, 0, 0)])
    assert list(flags) == [RF.kRffClippedEnd, 0] and np.array_equal(rays[1], (0.5, 0, 0))
    rays, flags = _apply(RF.clip_ray_filter(2.0), [(0, 0, 0), (0, 0, 10), (0, 0, 0), (1, 0, 0), (0, 0, 0), (np.nan, 0, 0)])
    assert rays.shape[0] == 4 and np.array_equal(rays[1], (0, 0, 2)) and list(flags) == [RF.kRffClippedEnd, 0]
    rays, flags = _apply(RF.good_ray_filter(5.0), [(0, 0, 0), (0, 0, 10), (0, 0, 0), (1, 0, 0), (np.inf, 0, 0), (1, 0, 0)])
    assert np.array_equal(rays, [(0, 0, 0), (1, 0, 0)])


def test_oracle_clipbox_expectations():
    box = RF.Aabb((-1.0, -1.0, -1.0), (2.0, 2.0, 2.0))
    res = 0.2
    through = [(-2, 0, 0), (3, 0, 0), (0, -2, 0), (0, 3, 0), (0, 0, 3), (0, 0, -2)]
    om = OracleMap(res)
    rays, flags = _apply(RF.clip_bounded(box), through)
    om.integrate_occupancy(rays, filter_flags=flags)
    touched = 0
    for key, layers in om.chunks().items():
        occ = layers["occupancy"].reshape(32, 32, 32)  # [z][y][x]
        seen = np.isfinite(occ)
        touched += int(seen.sum())
        assert np.all(occ[seen] < 0.0), "clipped rays must leave no occupied voxel"
        zs, ys, xs = np.nonzero(seen)
        centre = (np.stack([xs, ys, zs], axis=1) + 0.5) * res + (np.array(key) * 32 * res - 16 * res)
        assert np.all(centre + 0.5 * res >= box.min - 1e-9) and np.all(centre - 0.5 * res <= box.max + 1e-9)
    assert touched > 30
    # rays ending at the origin keep their sample: exactly one occupied voxel, the one holding (0, 0, 0)
    om2 = OracleMap(res)
    rays, flags = _apply(RF.clip_bounded(box), [(-2, 0, 0), (0, 0, 0), (0, -2, 0), (0, 0, 0), (0, 0, 3), (0, 0, 0)])
    om2.integrate_occupancy(rays, filter_flags=flags)
    occupied = sum(int(np.count_nonzero(np.isfinite(l["occupancy"]) & (l["occupancy"] > 0))) for l in om2.chunks().values())
    assert occupied == 1


def test_cpp_mirror_filters_agree_with_the_numpy_restatement(tmp_path):
    """The two host mirrors restate ohm/RayFilter.cpp + ohm::Aabb::clipLine independently (C++ per ray, numpy per
    batch): same decisions, same flags, same clipped points bit for bit.  Runs the driver's host-only filter mode."""
    import os
    import struct
    import subprocess

    import ohm_amd
    from ohm_amd import synth
    driver = os.path.join(os.path.dirname(ohm_amd.LIB_PATH), "gpumap_driver")
    assert os.path.exists(driver), "gpumap_driver missing: run __graft_entry__.build()"
    rays = synth.random_rays(4000, extent=4.0, seed=91, origin_spread=3.0)
    rays[10] = np.nan
    rays[21] = np.inf
    rays[30:32] = [[0.5, 0.5, 0.5], [0.5, 0.5, 0.5]]      # degenerate
    rays[40:42] = [[-3.0, 0.0, 0.0], [4.0, 0.0, 0.0]]     # axis aligned (infinite slab times on two axes)
    rp, op = tmp_path / "rays.bin", tmp_path / "out.bin"
    with open(rp, "wb") as f:
        f.write(struct.pack("<Q", rays.shape[0]))
        f.write(np.ascontiguousarray(rays).tobytes())
    box = RF.Aabb((-1.0, -1.0, -1.0), (2.0, 2.0, 2.0))
    cases = {"filter:clipbounded": (RF.clip_bounded(box), 0.0), "filter:cliptobounds": (RF.clip_to_bounds(box), 0.0),
             "filter:clipray": (RF.clip_ray_filter(2.5), 2.5), "filter:goodray": (RF.good_ray_filter(3.0), 3.0)}
    rec = np.dtype([("ok", np.uint8), ("flags", np.uint8), ("start", np.float64, 3), ("end", np.float64, 3)])
    for mode, (filt, param) in cases.items():
        res = subprocess.run([driver, mode, repr(param), "0", str(rp), str(op)], capture_output=True, text=True)
        assert res.returncode == 0, (mode, res.stderr)
        got = np.frombuffer(open(op, "rb").read(), dtype=rec)
        keep, starts, ends, flags = filt(rays[0::2].copy(), rays[1::2].copy())
        assert np.array_equal(got["ok"].astype(bool), keep), mode
        k = keep
        # the per-ray form sets kRffInvalid on rejected rays; compare the flags of the accepted ones
        assert np.array_equal(got["flags"][k], flags[k]), mode
        assert np.array_equal(got["start"][k].view(np.uint64), np.asarray(starts)[k].view(np.uint64)), mode
        assert np.array_equal(got["end"][k].view(np.uint64), np.asarray(ends)[k].view(np.uint64)), mode
