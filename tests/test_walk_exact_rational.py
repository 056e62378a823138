"""An independent exact check of the walk's tie-breaking.

tests/exact_walk.py evaluates the reference rule -- smallest exit time, ties to the higher axis, time = initial +
delta * |stepped| (/root/reference/ohm/LineWalkCompute.h:282-307; its own tests: tests/ohmtest/LineWalkTests.cpp:43-197)
-- in exact integer arithmetic on lattice-aligned rays (dyadic 0.125 m voxels, coordinates on a 1/64-voxel lattice, so
the inputs are exact in fp64 as well).  >= 10^4 generated tie-rich rays: plane / space diagonals through voxel centres
and corners, 2:1 / 3:1 slopes, negative directions, zero components, walks across region boundaries.  The CPU oracle
(here) and the HIP line-key query (-m gpu) must both produce exactly the exact walker's key sequences."""
import numpy as np
import pytest

from exact_walk import generate, walk, key_of, Undecidable, SUB

RES = 0.125
N_RAYS = 12000


@pytest.fixture(scope="module")
def exact_rays():
    rays, discarded = generate(N_RAYS, resolution=RES)
    return rays, discarded


def test_exact_walker_on_hand_cases():
    """The exact walker itself, on sequences worked out by hand (global voxel coordinates, 32^3 regions centred on 0)."""
    c = SUB // 2

    def voxels(keys):
        return [tuple(32 * r + l - 16 for r, l in zip(region, local)) for region, local in keys]

    # x-y diagonal from a voxel centre: exact ties at every corner, the HIGHER axis (y) steps first
    assert voxels(walk((c, c, c), (c + 2 * SUB, c + 2 * SUB, c))) == [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 2, 0), (2, 2, 0)]
    # space diagonal: z first, then y, then x
    assert voxels(walk((c, c, c), (c + SUB, c + SUB, c + SUB))) == [(0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1)]
    # negative directions tie the same way
    assert voxels(walk((c, c, c), (c - SUB, c - SUB, c))) == [(0, 0, 0), (0, -1, 0), (-1, -1, 0)]
    # 2:1 slope from a generic offset: no ties; x crossings at u = 27/128, 91/128, y at 45/64 = 90/128 -- y just before x
    assert voxels(walk((37, 19, c), (37 + 2 * SUB, 19 + SUB, c))) == [(0, 0, 0), (1, 0, 0), (1, 1, 0), (2, 1, 0)]
    # a 2:1 slope that DOES tie exactly is refused: fp64 has no defined answer there
    with pytest.raises(Undecidable):
        walk((0, 0, c), (4 * SUB, 2 * SUB, c))  # x: (1 + k) / 4, y: (1 + k) / 2 -> both 1/2
    # region / local keys: voxel -17 is local 31 of region -1, voxel 16 is local 0 of region 1
    assert key_of((-17, 16, 0)) == ((-1, 1, 0), (31, 0, 16))


def test_generator_covers_the_tie_cases(exact_rays):
    rays, discarded = exact_rays
    assert len(rays) == N_RAYS
    d = np.array([np.subtract(e, s) for s, e, _ in rays])
    assert (np.sum(d == 0, axis=1) > 0).sum() > 1000          # zero components
    assert (d < 0).any(axis=1).sum() > 4000                   # negative directions
    ad = np.abs(d)
    diag = ((ad[:, 0] == ad[:, 1]) & (ad[:, 0] > 0)) | ((ad[:, 1] == ad[:, 2]) & (ad[:, 1] > 0)) | \
           ((ad[:, 0] == ad[:, 2]) & (ad[:, 0] > 0))
    assert diag.sum() > 3000                                  # equal slopes: structural ties
    crossings = sum(1 for _, _, keys in rays if len({k[0] for k in keys}) > 1)
    assert crossings > 3000                                   # walks through more than one region
    assert sum(len(k) for _, _, k in rays) > 300000
    assert discarded < 2 * N_RAYS


def test_oracle_walk_equals_the_exact_walker(exact_rays):
    from oracle.oracle import OracleMap
    rays, _ = exact_rays
    om = OracleMap(RES)
    for i, (s, e, keys) in enumerate(rays):
        got, _, _ = om.walk(s, e, 0)
        assert got == keys, (i, s, e, got[:6], keys[:6])


@pytest.mark.gpu
def test_hip_line_keys_equal_the_exact_walker(gpu, exact_rays):
    from ohm_amd import GpuMap, OccupancyMap
    rays, _ = exact_rays
    lines = np.array([list(s) + list(e) for s, e, _ in rays], dtype=np.float64)
    longest = max(len(k) for _, _, k in rays)
    gm = GpuMap(OccupancyMap(RES))
    regions, voxels, counts = gm.lineKeys(lines, max_keys_per_line=longest + 2)
    for i, (_, _, keys) in enumerate(rays):
        assert counts[i] == len(keys), (i, lines[i], counts[i], len(keys))
        got = [(tuple(int(v) for v in regions[i, j]), tuple(int(v) for v in voxels[i, j])) for j in range(len(keys))]
        assert got == keys, (i, lines[i])
