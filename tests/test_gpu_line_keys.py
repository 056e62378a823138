"""-m gpu: the fp64 device line walk reports exactly the CPU walk's voxel keys, in order (LineKeysQueryGpu equivalent;
the key-parity probe SURVEY 8f names).  Covers random segments, axis-aligned and diagonal lines through voxel corners
(tie cases), sub-epsilon lines and both map origins of tests/ohmtest/LineWalkTests.cpp."""
import numpy as np
import pytest

from ohm_amd import GpuMap, OccupancyMap, synth
from oracle.oracle import OracleMap

pytestmark = pytest.mark.gpu


def _lines():
    rng = synth.random_rays(3000, extent=3.0, seed=31, origin_spread=3.0).reshape(-1, 6)
    special = []
    for s in range(1, 6):
        for d in [(1, 0, 0), (0, -1, 0), (0, 0, 1), (1, 1, 0), (-1, 1, 0), (1, 1, 1), (-1, -1, 1), (1, -1, -1)]:
            special.append([0, 0, 0] + [v * s * 0.7 for v in d])
            special.append([0.05, 0.05, 0.05] + [0.05 + v * s * 0.4 for v in d])
    for k in range(50):
        c = 0.1 * k
        special.append([c - 1e-9, c, c, c + 1e-9, c, c])
        special.append([c, c, c, c, c, c])
    return np.concatenate([rng, np.array(special, dtype=np.float64)])


@pytest.mark.parametrize("origin", [(0.0, 0.0, 0.0), (0.05, 0.05, 0.05)])
def test_line_keys_match_cpu_walk(gpu, origin):
    lines = _lines()
    map_ = OccupancyMap(0.1)
    map_.setOrigin(origin)
    gm = GpuMap(map_)
    regions, voxels, counts = gm.lineKeys(lines, max_keys_per_line=256)
    om = OracleMap(0.1)
    om.set_origin(origin)
    total = 0
    for i, ln in enumerate(lines):
        keys, _, _ = om.walk(ln[:3], ln[3:], 0)
        assert counts[i] == len(keys), (i, ln)
        got = [(tuple(int(v) for v in regions[i, j]), tuple(int(v) for v in voxels[i, j])) for j in range(len(keys))]
        assert got == keys, (i, ln)
        total += len(keys)
    assert total > 100000


def test_line_keys_query_gpu_interface(gpu):
    """The query object of the reference (ohmgpu/LineKeysQueryGpu.h; tests/ohmtestgpu/GpuLineKeysTests.cpp compares it
    with the CPU LineKeysQuery ray by ray): setRays / execute / numberOfResults / resultIndices / resultCounts /
    intersectedVoxels, against the oracle's CPU walk."""
    from ohm_amd import LineKeysQueryGpu
    lines = _lines()[:800]
    map_ = OccupancyMap(0.1)
    gm = GpuMap(map_)
    query = LineKeysQueryGpu(gm)
    query.setRays(lines.reshape(-1, 3))
    assert query.rayPointCount() == 2 * lines.shape[0]
    assert query.execute() and query.wait()
    assert query.numberOfResults() == lines.shape[0]
    regions, local = query.intersectedVoxels()
    indices, counts = query.resultIndices(), query.resultCounts()
    om = OracleMap(0.1)
    for i, ln in enumerate(lines):
        keys, _, _ = om.walk(ln[:3], ln[3:], 0)
        assert counts[i] == len(keys)
        first = int(indices[i])
        got = [(tuple(int(v) for v in regions[first + j]), tuple(int(v) for v in local[first + j])) for j in range(len(keys))]
        assert got == keys, (i, ln)
    assert int(indices[-1] + counts[-1]) == regions.shape[0]
    query.reset()
    assert query.numberOfResults() == 0
