"""world_size-2 gloo tests (CPU) of the multi-GPU protocol in ohm_amd/distributed.py: ray sharding by sensor origin,
region-key union, and the additive occupancy-delta merge.  Per-rank maps come from the CPU oracle; the collective is
gloo, exactly the code path that runs over RCCL on GPUs (backend-agnostic torch.distributed calls)."""
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


def _two_origin_rays(n_per_origin):
    a = synth.rays_c0(n=n_per_origin, origin=(0.05, 0.05, 0.05), length=5.0, seed=901)
    b = synth.rays_c0(n=n_per_origin, origin=(4.05, 0.05, 0.05), length=5.0, seed=902)
    return np.concatenate([a, b])


def _worker(rank, world, port, n_per_origin, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import OracleMap
        rays = _two_origin_rays(n_per_origin)
        shard = D.shard_rays_by_origin(rays, world, rank)
        om = OracleMap(0.1)
        om.integrate_occupancy(shard)
        chunks = om.chunks()
        local_keys = np.array(sorted(chunks.keys()), dtype=np.int16).reshape(-1, 3)
        union = D.union_region_keys(local_keys)
        voxels = 32 ** 3
        inf = np.float32(np.inf)
        local = np.full((len(union), voxels), inf, dtype=np.float32)
        for i, k in enumerate(union):
            c = chunks.get(tuple(int(v) for v in k))
            if c is not None:
                local[i] = c["occupancy"]
        base = np.full_like(local, inf)
        merged = D.merge_occupancy_deltas(torch.from_numpy(base), torch.from_numpy(local), -2.0, 3.511).numpy()
        np.save(os.path.join(result_dir, f"merged_{rank}.npy"), merged)
        np.save(os.path.join(result_dir, f"union_{rank}.npy"), union)
        np.save(os.path.join(result_dir, f"shard_{rank}.npy"), shard)
    finally:
        dist.destroy_process_group()


def test_shard_by_origin_is_a_partition():
    rays = _two_origin_rays(500)
    s0 = D.shard_rays_by_origin(rays, 2, 0)
    s1 = D.shard_rays_by_origin(rays, 2, 1)
    assert s0.shape[0] + s1.shape[0] == rays.shape[0]
    assert len(np.unique(s0[0::2], axis=0)) == 1 and len(np.unique(s1[0::2], axis=0)) == 1
    assert not np.array_equal(s0[0], s1[0])


def test_two_rank_gloo_merge_matches_additive_rule_and_sequential_where_unclamped(tmp_path):
    world, n = 2, 3000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    m0, m1 = np.load(tmp_path / "merged_0.npy"), np.load(tmp_path / "merged_1.npy")
    u0, u1 = np.load(tmp_path / "union_0.npy"), np.load(tmp_path / "union_1.npy")
    assert np.array_equal(u0, u1), "every rank must derive the same union order"
    assert np.array_equal(m0.view(np.uint32), m1.view(np.uint32)), "replicas must be bit-identical after the merge"

    # expected additive merge computed in one process from the two shards
    from oracle.oracle import OracleMap
    shards = [np.load(tmp_path / f"shard_{r}.npy") for r in range(world)]
    singles = []
    for s in shards:
        om = OracleMap(0.1)
        om.integrate_occupancy(s)
        singles.append(om.chunks())
    seq = OracleMap(0.1)
    for s in shards:
        seq.integrate_occupancy(s)
    seq_chunks = seq.chunks()
    inf = np.float32(np.inf)
    unclamped_checked = 0
    for i, k in enumerate(u0):
        key = tuple(int(v) for v in k)
        tiles = [c[key]["occupancy"] if key in c else np.full(32 ** 3, inf, np.float32) for c in singles]
        obs = [np.isfinite(t) for t in tiles]
        total = sum(np.where(o, t, 0).astype(np.float32) for t, o in zip(tiles, obs))
        expect = np.where(obs[0] | obs[1], np.clip(total, -2.0, 3.511), inf).astype(np.float32)
        assert np.allclose(m0[i], expect, rtol=1e-6, atol=1e-6, equal_nan=True)
        # Where provably no clamp engaged, the additive merge equals the sequential CPU integration of shard 0 then
        # shard 1: (a) voxels only one shard observed; (b) voxels both shards only ever missed (negative, monotone
        # histories) whose sum stays above the min clamp.
        st = seq_chunks[key]["occupancy"]
        one_only = obs[0] ^ obs[1]
        miss_only = obs[0] & obs[1] & (tiles[0] < 0) & (tiles[1] < 0) & (tiles[0] > -1.7) & (tiles[1] > -1.7) & \
            (total > -1.9)
        safe = one_only | miss_only
        assert np.allclose(m0[i][safe], st[safe], rtol=1e-5, atol=1e-5)
        unclamped_checked += int(miss_only.sum())
    assert unclamped_checked > 100


# ---------------------------------------------------------------------------------------------------------------------
# Owner-computes (exact) mode: region ownership rule, ray all-gather, partition == single map.
# ---------------------------------------------------------------------------------------------------------------------
def _owner_formula(keys, world, shift):
    """numpy restatement of regionOwner() (ohm_amd/csrc/ohmhip_internal.h) to pin the exported C function."""
    k = np.asarray(keys, dtype=np.int64) >> shift
    m32 = np.uint64(0xFFFFFFFF)
    x, y, z = [(k[:, a].astype(np.int64) & 0xFFFFFFFF).astype(np.uint64) for a in range(3)]
    h = (x * np.uint64(0x9E3779B1)) & m32
    h = ((h ^ y) * np.uint64(0x85EBCA77)) & m32
    h = ((h ^ z) * np.uint64(0xC2B2AE3D)) & m32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x27D4EB2F)) & m32
    h ^= h >> np.uint64(13)
    return (h % np.uint64(world)).astype(np.uint32)


def test_region_owner_rule():
    rng = np.random.default_rng(5)
    keys = rng.integers(-32768, 32768, size=(5000, 3)).astype(np.int16)
    keys[:6] = [[0, 0, 0], [-1, -1, -1], [32767, -32768, 5], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    for world in (2, 3, 8):
        for shift in (0, 1, 3):
            owners = D.region_owner(keys, world, shift)
            assert owners.max() < world
            assert np.array_equal(owners, _owner_formula(keys, world, shift))
            assert len(np.unique(owners)) == world
    assert np.all(D.region_owner(keys, 1, 0) == 0)
    # blocks of 2^shift regions share an owner
    base = np.array([[4, -6, 2]], dtype=np.int16)
    block = base + np.array([[dx, dy, dz] for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], dtype=np.int16)
    assert len(np.unique(D.region_owner(block, 8, 1))) == 1


def _owner_worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import OracleMap
        # ranks hold different numbers of rays; two rounds
        chunks = {}
        om = OracleMap(0.1)
        for rnd in range(2):
            n = 1500 + 700 * rank + 100 * rnd
            local = synth.rays_c0(n=n, origin=(0.05 + 4.0 * rank, 0.05, 0.05), length=5.0, seed=950 + 10 * rnd + rank)
            stream = D.gather_rays(torch.from_numpy(local)).numpy()
            np.save(os.path.join(result_dir, f"stream_{rnd}_{rank}.npy"), stream)
            np.save(os.path.join(result_dir, f"local_{rnd}_{rank}.npy"), local)
            om.integrate_occupancy(stream)
        # the CPU stand-in for a map with the ownership filter: integrate the stream, keep the regions this rank owns
        all_chunks = om.chunks()
        keys = np.array(sorted(all_chunks.keys()), dtype=np.int16).reshape(-1, 3)
        mine = D.region_owner(keys, world, 0) == rank
        np.save(os.path.join(result_dir, f"keys_{rank}.npy"), keys[mine])
        np.save(os.path.join(result_dir, f"occ_{rank}.npy"),
                np.stack([all_chunks[tuple(int(v) for v in k)]["occupancy"] for k in keys[mine]]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_owner_computes_stream_and_partition(tmp_path):
    world = 2
    mp.spawn(_owner_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle.oracle import OracleMap
    seq = OracleMap(0.1)
    for rnd in range(2):
        streams = [np.load(tmp_path / f"stream_{rnd}_{r}.npy") for r in range(world)]
        locals_ = [np.load(tmp_path / f"local_{rnd}_{r}.npy") for r in range(world)]
        expect = np.concatenate(locals_)  # rank order, then ray order; ragged counts
        for s in streams:
            assert np.array_equal(s, expect)
        seq.integrate_occupancy(expect)
    seq_chunks = seq.chunks()
    seen = set()
    for r in range(world):
        keys, occ = np.load(tmp_path / f"keys_{r}.npy"), np.load(tmp_path / f"occ_{r}.npy")
        assert len(keys) > 0
        for k, tile in zip(keys, occ):
            key = tuple(int(v) for v in k)
            assert key not in seen
            seen.add(key)
            assert np.array_equal(tile.view(np.uint32), seq_chunks[key]["occupancy"].view(np.uint32))
    assert seen == set(seq_chunks.keys())
