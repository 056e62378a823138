"""-m gpu: the replica merge of include/ohmhip.h ("Replica merge") on real device tiles.

* one process, two maps on the test GPU standing in for two ranks: the transport-agnostic steps (merge_keys / merge_pack
  / merge_apply / merge_finish) with the payloads summed on the host -- checked bit for bit against the additive rule
  evaluated with numpy from two CPU-oracle maps, over two rounds (the second on a non-trivial base);
* two processes (gloo), each with its own map on the GPU: the protocol of ohm_amd.distributed.ReplicaMerger -- key
  exchange, shared set, payload all-reduce -- end to end.
The RCCL path of the library (ohmhip_map_merge_replicas) runs the same pack / apply kernels; with the single test GPU
it is exercised at world size 1 in tests/test_gpu_distributed.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from ohm_amd import GpuMap, OccupancyMap, synth
from ohm_amd import _lib as L

from parity import make_oracle

pytestmark = pytest.mark.gpu

VOXELS = 32 ** 3


def _keys(gm):
    n = C.c_size_t(0)
    L.check(L.lib.ohmhip_map_merge_keys(gm._handle, None, 0, C.byref(n)))
    keys = np.zeros((max(n.value, 1), 3), dtype=np.int16)
    L.check(L.lib.ohmhip_map_merge_keys(gm._handle, keys.ctypes.data, n.value, C.byref(n)))
    return keys[:n.value]


class _Payload:
    def __init__(self, n):
        self.n = n
        self.delta, self.obs = L._vp(), L._vp()
        L.check(L.lib.ohmhip_buffer_create(C.byref(self.delta), 4 * n * VOXELS, 3))
        L.check(L.lib.ohmhip_buffer_create(C.byref(self.obs), n * VOXELS, 3))
        self.d_delta, self.d_obs = L._vp(), L._vp()
        L.check(L.lib.ohmhip_buffer_ptr(self.delta, C.byref(self.d_delta)))
        L.check(L.lib.ohmhip_buffer_ptr(self.obs, C.byref(self.d_obs)))

    def read(self):
        d = np.zeros(self.n * VOXELS, dtype=np.float32)
        o = np.zeros(self.n * VOXELS, dtype=np.uint8)
        L.check(L.lib.ohmhip_buffer_read(self.delta, d.ctypes.data, d.nbytes, 0, None, None, None))
        L.check(L.lib.ohmhip_buffer_read(self.obs, o.ctypes.data, o.nbytes, 0, None, None, None))
        return d, o

    def write(self, d, o):
        L.check(L.lib.ohmhip_buffer_write(self.delta, d.ctypes.data, d.nbytes, 0, None, None, None))
        L.check(L.lib.ohmhip_buffer_write(self.obs, o.ctypes.data, o.nbytes, 0, None, None, None))

    def close(self):
        L.lib.ohmhip_buffer_destroy(self.delta)
        L.lib.ohmhip_buffer_destroy(self.obs)


def _merge_two(gms):
    """The library's steps for two replicas in one process; returns the shared keys."""
    key_sets = [set(map(tuple, _keys(gm).tolist())) for gm in gms]
    shared = np.array(sorted(key_sets[0] & key_sets[1]), dtype=np.int16).reshape(-1, 3)
    n = len(shared)
    if n:
        payloads = [_Payload(n) for _ in gms]
        parts = []
        for gm, p in zip(gms, payloads):
            L.check(L.lib.ohmhip_map_merge_pack(gm._handle, shared.ctypes.data, n, p.d_delta, p.d_obs), "merge_pack")
            parts.append(p.read())
        d_sum = parts[0][0] + parts[1][0]
        o_sum = parts[0][1] + parts[1][1]
        for gm, p in zip(gms, payloads):
            p.write(d_sum, o_sum)
            L.check(L.lib.ohmhip_map_merge_apply(gm._handle, shared.ctypes.data, n, p.d_delta, p.d_obs), "merge_apply")
            p.close()
    for gm in gms:
        L.check(L.lib.ohmhip_map_merge_finish(gm._handle), "merge_finish")
        gm.wait()
        assert len(_keys(gm)) == 0
    return shared


def _zero_where_unobserved(x):
    return np.where(np.isinf(x), np.float32(0), x).astype(np.float32)


def test_two_replicas_merge_by_the_additive_rule_over_two_rounds(gpu):
    origins = [(0.05, 0.05, 0.05), (4.05, 0.05, 0.05)]
    maps = [OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",)) for _ in origins]
    gms = [GpuMap(m) for m in maps]
    oracles = [make_oracle(m) for m in maps]  # what each replica integrates on its own
    for gm in gms:
        L.check(L.lib.ohmhip_map_enable_merge(gm._handle), "enable_merge")
    base = {}  # region key -> merged tile after the previous round (absent: unobserved)
    inf = np.float32(np.inf)
    for rnd in range(2):
        previous = [dict(om.chunks()) for om in oracles] if rnd else [{}, {}]
        for r, (gm, om, origin) in enumerate(zip(gms, oracles, origins)):
            rays = synth.rays_c0(n=4000, origin=origin, length=5.0, seed=700 + 10 * rnd + r)
            assert gm.integrateRays(rays) == rays.shape[0]
            om.integrate_occupancy(rays)
        shared = _merge_two(gms)
        assert len(shared) > 0, "the two 5 m spheres, 4 m apart, must share regions"
        for gm in gms:
            gm.syncVoxels()
        # Expected: per rank delta = (value after its own rays, started from `base`) - base.  The oracle maps integrate
        # in isolation (never see the merged values), so their deltas are exact only in the first round; later rounds
        # check replica equality and the rule against the replicas' own pre-merge values instead (below).
        for key in map(tuple, shared.tolist()):
            a, b = maps[0].chunks[key]["occupancy"], maps[1].chunks[key]["occupancy"]
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "replicas differ after the merge"
            if rnd == 0:
                xs = [om.chunks()[key]["occupancy"] for om in oracles]
                delta = _zero_where_unobserved(xs[0]) + _zero_where_unobserved(xs[1])
                observed = ~np.isinf(xs[0]) | ~np.isinf(xs[1])
                expected = np.where(observed, np.clip(delta, np.float32(-2.0), np.float32(maps[0].max_voxel_value)), inf)
                assert np.array_equal(a.view(np.uint32), expected.astype(np.float32).view(np.uint32))
            base[key] = a.copy()
        # regions only one replica touched keep that replica's own (oracle) values in the first round
        if rnd == 0:
            for r, (m, om) in enumerate(zip(maps, oracles)):
                for key, layers in om.chunks().items():
                    if key not in set(map(tuple, shared.tolist())):
                        assert np.array_equal(m.chunks[key]["occupancy"].view(np.uint32),
                                              layers["occupancy"].view(np.uint32))
        del previous


def test_merge_rule_on_a_non_trivial_base(gpu):
    """Second-round arithmetic checked exactly: both replicas start from the same uploaded base tiles."""
    key = np.array([[0, 0, 0]], dtype=np.int16)
    rng = np.random.default_rng(5)
    base = rng.uniform(-1.5, 2.5, VOXELS).astype(np.float32)
    base[rng.random(VOXELS) < 0.3] = np.inf
    maps = [OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",)) for _ in range(2)]
    gms = [GpuMap(m) for m in maps]
    values = []
    for r, gm in enumerate(gms):
        src = (C.c_void_p * 1)(base.ctypes.data)
        L.check(L.lib.ohmhip_map_write_regions(gm._handle, L.LID_OCCUPANCY, key.ctypes.data, 1, src), "write_regions")
        L.check(L.lib.ohmhip_map_enable_merge(gm._handle), "enable_merge")  # base := what was uploaded
        rays = synth.rays_c0(n=3000, origin=(0.05 + 0.5 * r, 0.05, 0.05), length=1.2, seed=40 + r)
        gm.integrateRays(rays)
        gm.syncVoxels()
        values.append(maps[r].chunks[(0, 0, 0)]["occupancy"].copy())
    shared = _merge_two(gms)
    assert (0, 0, 0) in set(map(tuple, shared.tolist()))
    for gm in gms:
        gm.syncVoxels()
    b0 = _zero_where_unobserved(base)
    deltas = [np.where(np.isinf(v), np.float32(0), v - b0).astype(np.float32) for v in values]
    observed = ~np.isinf(values[0]) | ~np.isinf(values[1])
    expected = np.where(observed, np.clip(b0 + (deltas[0] + deltas[1]), np.float32(-2.0),
                                          np.float32(maps[0].max_voxel_value)), np.float32(np.inf)).astype(np.float32)
    for m in maps:
        got = m.chunks[(0, 0, 0)]["occupancy"]
        assert np.array_equal(got.view(np.uint32), expected.view(np.uint32))


def test_two_process_gloo_merge_on_device_tiles(gpu, tmp_path):
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_merge2_worker.py")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, worker, str(rank), "2", str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rank in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0 and "MERGE2_OK" in out, (p.returncode, out[-2000:], err[-4000:])
    shared0 = np.load(tmp_path / "shared_0.npy")
    shared1 = np.load(tmp_path / "shared_1.npy")
    assert np.array_equal(shared0, shared1) and len(shared0) > 0
    t0, t1 = np.load(tmp_path / "tiles_0.npy"), np.load(tmp_path / "tiles_1.npy")
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)), "replicas must be bit-identical on the shared regions"
    # and equal to the additive rule over the two ranks' own (CPU oracle) maps
    x0, x1 = np.load(tmp_path / "own_0.npy"), np.load(tmp_path / "own_1.npy")
    delta = _zero_where_unobserved(x0) + _zero_where_unobserved(x1)
    observed = ~np.isinf(x0) | ~np.isinf(x1)
    expected = np.where(observed, np.clip(delta, np.float32(-2.0), np.float32(3.511)), np.float32(np.inf))
    assert np.array_equal(t0.view(np.uint32), expected.astype(np.float32).view(np.uint32))
