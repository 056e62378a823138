"""CPU tests of the partitioned-map protocol (ohm_amd/distributed.py: RegionPartition, territories_from_origins,
exchange_routed_rays): the ownership rule pinned by a numpy restatement, the territory builder, and -- world_size 2 over
gloo -- route -> all-to-all -> integrate-what-you-own against the sequential CPU oracle.  The routing on the GPU is a
kernel of the library (tests/test_gpu_partitioned.py holds it to the same rule); here the per-rank router and mapper
are the oracle (its line walk names the regions a ray touches), the exchange is the product code."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from ohm_amd import synth  # noqa: E402
from ohm_amd import distributed as D  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _table_rule(part, keys):
    """numpy restatement of partitionOwner() (ohm_amd/csrc/ohmhip_internal.h) for a table partition."""
    k = np.asarray(keys, dtype=np.int64) >> part.block_shift
    cell = k - np.asarray(part.grid_origin, dtype=np.int64)
    cell = np.clip(cell, 0, np.asarray(part.table.shape, dtype=np.int64) - 1)
    return part.table[cell[:, 0], cell[:, 1], cell[:, 2]].astype(np.uint32)


def test_partition_table_rule_and_clamping():
    rng = np.random.default_rng(11)
    table = rng.integers(0, 5, size=(7, 4, 3)).astype(np.uint8)
    for shift in (0, 1, 3):
        part = D.RegionPartition(5, 2, shift, (-3, 2, -1), table)
        keys = rng.integers(-200, 200, size=(4000, 3)).astype(np.int16)
        keys[:4] = [[-32768, -32768, -32768], [32767, 32767, 32767], [-3 << shift, 2 << shift, -1 << shift], [0, 0, 0]]
        got = part.owners(keys)
        assert np.array_equal(got, _table_rule(part, keys))
    # outside the grid: the nearest cell's owner
    part = D.RegionPartition(5, 0, 0, (0, 0, 0), table)
    far = np.array([[-100, 1, 1], [100, 1, 1], [3, -50, 2], [3, 50, 2], [3, 1, -9], [3, 1, 9]], dtype=np.int16)
    near = np.array([[0, 1, 1], [6, 1, 1], [3, 0, 2], [3, 3, 2], [3, 1, 0], [3, 1, 2]], dtype=np.int16)
    assert np.array_equal(part.owners(far), part.owners(near))
    # no table: the block hash of ohmhip_region_owner
    hashed = D.RegionPartition(5, 1, 2)
    keys = rng.integers(-3000, 3000, size=(500, 3)).astype(np.int16)
    assert np.array_equal(hashed.owners(keys), D.region_owner(keys, 5, 2))
    # world_size 1: everything is rank 0's
    assert np.all(D.RegionPartition(1, 0).owners(keys) == 0)


def test_invalid_partitions_are_rejected():
    import ctypes as C
    from ohm_amd import _lib as L
    keys = np.zeros((2, 3), dtype=np.int16)
    owners = np.zeros(2, dtype=np.uint32)
    p = D.RegionPartition(4, 0, 1, (0, 0, 0), np.zeros((2, 2, 2), dtype=np.uint8)).c_struct()
    assert L.lib.ohmhip_partition_owners(C.byref(p), keys.ctypes.data, 2, owners.ctypes.data) == L.OK
    p.block_shift = 16
    assert L.lib.ohmhip_partition_owners(C.byref(p), keys.ctypes.data, 2, owners.ctypes.data) == L.ERR_INVALID_ARG
    p.block_shift = 0
    p.owners = None  # a grid without a table
    assert L.lib.ohmhip_partition_owners(C.byref(p), keys.ctypes.data, 2, owners.ctypes.data) == L.ERR_INVALID_ARG
    assert L.lib.ohmhip_partition_owners(None, keys.ctypes.data, 2, owners.ctypes.data) == L.ERR_INVALID_ARG
    assert L.lib.ohmhip_map_set_region_partition(None, C.byref(p)) == L.ERR_INVALID_ARG
    assert L.lib.ohmhip_map_route_rays(None, None, 0, 0, None, None, 0, owners.ctypes.data, None) == L.ERR_INVALID_ARG
    assert L.lib.ohmhip_comm_exchange_counts(None, owners.ctypes.data, owners.ctypes.data, None) == L.ERR_INVALID_ARG


def test_territories_follow_the_nearest_origin():
    world = 8
    parts = [D.territories_from_origins(synth.C4_ORIGINS, world, r, 3.2, block_shift=1) for r in range(world)]
    for p in parts[1:]:  # every rank computes the same table
        assert np.array_equal(p.table, parts[0].table) and p.grid_origin == parts[0].grid_origin
    part = parts[0]
    assert set(np.unique(part.table).tolist()) == set(range(world))
    # the region a sensor stands in belongs to its rank, and so does everything within 10 m of it (the origins are 40 m
    # apart: a 6.4 m block whose centre is nearer to another origin lies more than 10 m away)
    rng = np.random.default_rng(3)
    for r, o in enumerate(synth.C4_ORIGINS):
        pts = np.asarray(o) + rng.uniform(-10.0, 10.0, size=(500, 3))
        keys = np.floor(pts / 3.2 + 0.5).astype(np.int16)
        assert np.all(part.owners(keys) == r)
    # far outside the table the nearest cell decides: 1 km west of the west-most origins is theirs
    west = np.array([np.floor(np.array([-1000.0, y, 0.0]) / 3.2 + 0.5) for y in (-20.0, 20.0)]).astype(np.int16)
    assert part.owners(west).tolist() == [0, 4]
    # several origins per rank / fewer ranks than origins
    two = D.territories_from_origins(synth.C4_ORIGINS, 2, 0, 3.2, block_shift=0)
    assert set(np.unique(two.table).tolist()) == {0, 1}


from partition_ref import route_reference as _route  # noqa: E402


def _worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import OracleMap
        origins = [(0.05, 0.05, 0.05), (6.45, 0.05, 0.05)]
        part = D.territories_from_origins(origins, world, rank, 3.2, block_shift=0, margin=8.0)
        om = OracleMap(0.1, layers=("occupancy", "touch_time"))
        received_total = 0
        for rnd in range(2):
            n = 900 + 300 * rank + 50 * rnd  # ragged
            local = synth.rays_c0(n=n, origin=origins[rank], length=5.0, seed=400 + 10 * rnd + rank)
            # the side array of the batch (time stamps; every rank's clock runs on from the previous rank's, so that
            # "rank 0's batch, then rank 1's" is also the order of the stamps)
            stamps = 100.0 + 10.0 * rnd + 4.0 * rank + 0.001 * np.arange(n, dtype=np.float64)
            routed, counts, index = _route(OracleMap(0.1), part, local, world, with_index=True)
            got, recv_counts = D.exchange_routed_rays(torch.from_numpy(routed), counts)
            got_stamps = D.exchange_routed_side(torch.from_numpy(stamps[index]), counts, recv_counts)
            if rnd == 0:
                # one time base for the partitioned map: the first stamp of rank 0's first batch, as ONE map integrating
                # the ranks' batches in rank order would take it (PartitionedIntegrator._agree_on_time_base)
                base = torch.tensor([stamps[0]], dtype=torch.float64)
                dist.broadcast(base, src=0)
                om.set_first_ray_time(float(base.item()))
            stream = got.numpy().reshape(-1, 3)
            np.save(os.path.join(result_dir, f"local_{rnd}_{rank}.npy"), local)
            np.save(os.path.join(result_dir, f"stamps_{rnd}_{rank}.npy"), stamps)
            np.save(os.path.join(result_dir, f"stream_{rnd}_{rank}.npy"), stream)
            np.save(os.path.join(result_dir, f"counts_{rnd}_{rank}.npy"), np.array([counts, recv_counts]))
            assert got_stamps.shape[0] == stream.shape[0] // 2
            om.integrate_occupancy(stream, timestamps=got_stamps.numpy())
            received_total += stream.shape[0] // 2
        chunks = om.chunks()
        keys = np.array(sorted(chunks.keys()), dtype=np.int16).reshape(-1, 3)
        mine = part.owners(keys) == rank  # the ownership filter of the map: only own regions are kept
        np.save(os.path.join(result_dir, f"keys_{rank}.npy"), keys[mine])
        np.save(os.path.join(result_dir, f"occ_{rank}.npy"),
                np.stack([chunks[tuple(int(v) for v in k)]["occupancy"] for k in keys[mine]]))
        np.save(os.path.join(result_dir, f"touch_{rank}.npy"),
                np.stack([chunks[tuple(int(v) for v in k)]["touch_time"] for k in keys[mine]]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_route_exchange_integrate_matches_sequential(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle.oracle import OracleMap
    seq = OracleMap(0.1, layers=("occupancy", "touch_time"))
    for rnd in range(2):
        locals_ = [np.load(tmp_path / f"local_{rnd}_{r}.npy") for r in range(world)]
        for r, lr in enumerate(locals_):  # rank order, then ray order -- the side array (time stamps) with its rays
            seq.integrate_occupancy(lr, timestamps=np.load(tmp_path / f"stamps_{rnd}_{r}.npy"))
        counts = [np.load(tmp_path / f"counts_{rnd}_{r}.npy") for r in range(world)]
        # what rank s addressed to rank d is what d received from s, and some -- not all -- rays travel
        for s in range(world):
            for d in range(world):
                assert counts[s][0][d] == counts[d][1][s]
        assert 0 < counts[0][0][1] < locals_[0].shape[0] // 2 and 0 < counts[1][0][0] < locals_[1].shape[0] // 2
        # a rank's received stream is in (source rank, ray) order and a subsequence of the sources' rays
        for d in range(world):
            stream = np.load(tmp_path / f"stream_{rnd}_{d}.npy").reshape(-1, 6)
            at = 0
            for s in range(world):
                part = stream[at:at + counts[d][1][s]]
                src = locals_[s].reshape(-1, 6)
                idx = [np.flatnonzero((src == row).all(axis=1))[0] for row in part]
                assert idx == sorted(idx)
                at += counts[d][1][s]
    seq_chunks = seq.chunks()
    seen = set()
    for r in range(world):
        keys, occ = np.load(tmp_path / f"keys_{r}.npy"), np.load(tmp_path / f"occ_{r}.npy")
        touch = np.load(tmp_path / f"touch_{r}.npy")
        assert len(keys) > 0 and int(np.count_nonzero(touch)) > 0
        for k, tile, stamp_tile in zip(keys, occ, touch):
            key = tuple(int(v) for v in k)
            assert key not in seen
            seen.add(key)
            assert np.array_equal(tile.view(np.uint32), seq_chunks[key]["occupancy"].view(np.uint32)), key
            # the routed time stamps arrived with their rays: the touch-time layer of the partition is the sequential one
            assert np.array_equal(stamp_tile, seq_chunks[key]["touch_time"]), key
    assert seen == set(seq_chunks.keys())


def test_territories_by_load_balance_one_sensor():
    """One sensor's stream dealt over 8 ranks by measured load: every rank gets about an eighth of the segments, every
    region has exactly one owner, the regions around the sensor (which every ray crosses) are spread over the ranks."""
    rays = synth.rays_c1(n=1_000_000)                                  # one whole revolution
    loads = D.estimate_region_loads(rays, 3.2, ray_stride=16)
    assert loads[(0, 0, 0)] == pytest.approx(1_000_000, rel=0.01)    # every ray starts in the sensor's region
    part = D.territories_by_load(loads, 8, 3, (0.05, 0.05, 0.05), 3.2)
    keys = np.array(sorted(loads), dtype=np.int16)
    owners = part.owners(keys)
    weight = np.array([loads[tuple(int(v) for v in k)] for k in keys])
    share = np.array([weight[owners == r].sum() for r in range(8)]) / weight.sum()
    assert share.min() > 0.10 and share.max() < 0.15, share
    hub = keys[np.abs(keys).max(axis=1) <= 1]
    assert len(np.unique(part.owners(hub))) >= 6
    # contiguous arcs: a rank's rim regions span a limited range of azimuth
    rim = np.abs(keys).max(axis=1) > 1
    az = np.arctan2(keys[rim][:, 1] * 3.2 - 0.05, keys[rim][:, 0] * 3.2 - 0.05)
    for r in range(1, 7):
        mine = az[owners[rim] == r]
        assert mine.max() - mine.min() < 2 * np.pi * 0.4


def _failure_worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        outcome = "ok"
        if rank == 1:
            # what PartitionedIntegrator does when its own routing raised: take part in the count exchange with the marker
            class _Failing:
                comm, group = None, None
            D.PartitionedIntegrator._announce_failure(_Failing(), np.zeros(world, dtype=np.uint32))
            outcome = "announced"
        else:
            try:
                D.exchange_routed_rays(torch.zeros((3, 6), dtype=torch.float64), [1, 2])
            except RuntimeError as exc:
                outcome = "peer failure: " + str(exc)
        with open(os.path.join(result_dir, f"outcome_{rank}.txt"), "w") as fh:
            fh.write(outcome)
    finally:
        dist.destroy_process_group()


def test_a_rank_local_failure_ends_the_step_on_every_rank(tmp_path):
    """a rank whose routing fails still takes part in the step's count exchange, with the failure marker, so
    its peers raise from the same call instead of blocking in the payload collective (world 2, gloo)."""
    mp.spawn(_failure_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "outcome_1.txt").read_text() == "announced"
    assert (tmp_path / "outcome_0.txt").read_text().startswith("peer failure")
