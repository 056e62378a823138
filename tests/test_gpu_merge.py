"""-m gpu: the replica merge of include/ohmhip.h ("Replica merge") on real device tiles.

* one process, two maps on the test GPU standing in for two ranks: the transport-agnostic steps (merge_keys / merge_pack
  / merge_apply / merge_finish) with the payloads summed on the host -- checked bit for bit against the additive rule
  evaluated with numpy from two CPU-oracle maps, over two rounds (the second on a non-trivial base);
* two processes (gloo), each with its own map on the GPU: the protocol of ohm_amd.distributed.ReplicaMerger -- key
  exchange, shared set, payload all-reduce -- end to end.
* three maps whose sensors MOVE: a region only one replica touched in an earlier round is entered by a
  second one later, a third replica that never touched it still has to apply it -- held to an independent numpy model
  of the shared base, and end to end against the sequential oracle where no clamp engaged.
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
        o_sum = np.maximum(parts[0][1], parts[1][1])
        for gm, p in zip(gms, payloads):
            p.write(d_sum, o_sum)
            L.check(L.lib.ohmhip_map_merge_apply(gm._handle, shared.ctypes.data, n, p.d_delta, p.d_obs), "merge_apply")
            p.close()
    exchanged = set(map(tuple, shared.tolist()))
    for gm, before in zip(gms, key_sets):
        L.check(L.lib.ohmhip_map_merge_finish(gm._handle), "merge_finish")
        gm.wait()
        # exchanged regions are settled; what only this replica modified stays pending on its shared base
        assert set(map(tuple, _keys(gm).tolist())) == before - exchanged
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


def _tiles(map_, keys):
    inf = np.float32(np.inf)
    return {k: (map_.chunks[k]["occupancy"].reshape(-1).copy() if k in map_.chunks else np.full(VOXELS, inf, np.float32))
            for k in keys}


def test_three_replicas_with_moving_sensors_keep_one_shared_base(gpu):
    """round 0 -- every sensor alone (nothing shared; regions stay pending).  Round 1 -- sensor 1 moves
    next to sensor 0: regions replica 0 touched ALONE in round 0 are now shared; its whole pending delta (rounds 0 + 1)
    must travel, and replica 2, which never saw those regions, must end up with the same tiles.  Round 2 -- sensor 2
    moves in while replica 0 is idle.  Every exchanged tile is checked bit for bit against a numpy model that tracks
    the shared base on its own; at the end everything is exchanged (full union) and the three replicas must be
    bit-identical and equal the sequential oracle wherever no clamp engaged."""
    from ohm_amd import distributed as D
    maps = [OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",)) for _ in range(3)]
    gms = [GpuMap(m) for m in maps]
    for gm in gms:
        L.check(L.lib.ohmhip_map_enable_merge(gm._handle), "enable_merge")
    home = (0.05, 0.05, 0.05)
    plan = [  # per round: origin of each replica's sensor (None: no rays)
        [home, (40.05, 0.05, 0.05), (0.05, 40.05, 0.05)],
        [home, (3.05, 0.05, 0.05), (0.05, 40.05, 0.05)],
        [None, (3.05, 0.05, 0.05), (0.05, 3.05, 0.05)],
    ]
    lo, hi = np.float32(-2.0), np.float32(maps[0].max_voxel_value)
    inf = np.float32(np.inf)
    model_base = {}      # region -> tile all replicas share (absent: unobserved)
    all_rays = []        # (round, rank) order == the sequential order the additive rule is compared with
    exchanged_ever = set()
    for rnd, origins in enumerate(plan):
        for r, origin in enumerate(origins):
            if origin is None:
                continue
            rays = synth.rays_c0(n=1500, origin=origin, length=4.0, seed=9100 + 10 * rnd + r)
            assert gms[r].integrateRays(rays) == rays.shape[0]
            all_rays.append(rays)
        for gm in gms:
            gm.syncVoxels()
        pending = [set(map(tuple, _keys(gm).tolist())) for gm in gms]
        counts = {}
        for p in pending:
            for k in p:
                counts[k] = counts.get(k, 0) + 1
        expect_shared = sorted(k for k, c in counts.items() if c > 1)
        pre = [_tiles(m, expect_shared) for m in maps]  # values before the merge
        shared, stats = D.merge_in_process(gms)
        assert [tuple(k) for k in shared.tolist()] == expect_shared
        if rnd == 0:
            assert len(expect_shared) == 0
        else:
            assert len(expect_shared) > 0
        if rnd == 1:
            # the point of the test: regions replica 0 alone had touched in round 0 are exchanged now
            assert any(k in pending[0] and k in pending[1] and k not in pending[2] for k in expect_shared)
        for gm in gms:
            gm.syncVoxels()
        for k in expect_shared:
            b0 = _zero_where_unobserved(model_base.get(k, np.full(VOXELS, inf, np.float32)))
            d_sum = np.zeros(VOXELS, dtype=np.float32)
            observed = np.zeros(VOXELS, dtype=bool)
            for r in range(3):
                x = pre[r][k]
                if k in pending[r]:
                    d_sum += np.where(np.isinf(x), np.float32(0), x - b0).astype(np.float32)
                    observed |= ~np.isinf(x)
                else:  # not pending: the replica holds the shared base (or nothing)
                    assert np.array_equal(_zero_where_unobserved(x).view(np.uint32), b0.view(np.uint32)) or k not in maps[r].chunks
            expected = np.where(observed, np.clip(b0 + d_sum, lo, hi), inf).astype(np.float32)
            for r in range(3):
                got = maps[r].chunks[k]["occupancy"].reshape(-1)
                assert np.array_equal(got.view(np.uint32), expected.view(np.uint32)), (rnd, r, k)
            model_base[k] = expected
            exchanged_ever.add(k)
        for r in range(3):  # still pending: exactly what was pending and not exchanged
            assert set(map(tuple, _keys(gms[r]).tolist())) == pending[r] - set(expect_shared)
    # final: exchange everything that is still pending anywhere -> three identical maps
    for gm in gms:
        L.check(L.lib.ohmhip_map_set_merge_mode(gm._handle, L.MERGE_FULL_UNION), "merge_mode")
    D.merge_in_process(gms, full_union=True)
    for gm in gms:
        assert len(_keys(gm)) == 0
        gm.syncVoxels()
    assert set(maps[0].chunks) == set(maps[1].chunks) == set(maps[2].chunks)
    for k, c in maps[0].chunks.items():
        for other in maps[1:]:
            assert np.array_equal(c["occupancy"].view(np.uint32), other.chunks[k]["occupancy"].view(np.uint32)), k
    # end to end: the sequential CPU integration of all rays; equal (to float summation order) wherever no clamp engaged
    om = make_oracle(maps[0])
    for rays in all_rays:
        om.integrate_occupancy(rays)
    seq = om.chunks()
    assert set(seq) == set(maps[0].chunks)
    dev = D.merge_deviation(maps[0].chunks, seq)
    assert dev["voxels_state_differs"] == 0
    checked = off = 0
    for k, c in seq.items():
        a = maps[0].chunks[k]["occupancy"].reshape(-1)
        b = c["occupancy"].reshape(-1)
        # values well inside (min, max) have almost never met a clamp on the way (a voxel pushed to the minimum and hit
        # afterwards can end here too, hence "almost": such voxels are counted, not excluded)
        free = np.isfinite(b) & (b > lo + 0.5) & (b < hi - 1.0)
        off += int((~np.isclose(a[free], b[free], rtol=1e-5, atol=1e-5)).sum())
        checked += int(free.sum())
    assert checked > 10000 and len(exchanged_ever) > 0
    assert off <= 0.002 * checked, (off, checked, dev)
    for gm in gms:
        gm.close()
