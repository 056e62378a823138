"""Multi-GPU support for the ray-integration path -- no reference equivalent: ohm is single device.

Sharding: rays are partitioned by SENSOR ORIGIN (one origin / contiguous block of origins per rank).  Every rank
integrates its shard into its own resident map with no data-path collective; that is what `bench.py --gpus N` times
(weak scaling).

Merging replicas (explicit, on demand): `merge_occupancy_deltas` reconciles per-rank occupancy layers with the additive
log-odds rule  merged = clamp(base + sum_r (x_r - base), min, max)  over the UNION of regions the ranks touched since
the common base state, using ONE all-reduce(sum) of the packed delta tiles (RCCL over xGMI when the tensors live on
GPUs, gloo in the CPU tests).  An unobserved voxel (+inf) contributes base 0 and stays unobserved only if no rank
observed it.

Caveat, stated where the results are used: the additive rule equals the sequential CPU integration of the concatenated
shards exactly when no min/max clamp engages between the shards' updates of a voxel (log-odds updates commute until
they saturate).  Where clamps interact the merged value is the order-free sum, which is the standard map-merge
semantic, not the sequential one.  NDT / TSDF layers are not additive: replicas only.

Exact alternative, all map types -- "owner computes" (`region_owner`, `gather_rays`, `OwnerComputesIntegrator`): regions
are dealt to ranks by a block hash, every rank sees the same ray stream (an all-gather of the ranks' batches, 48 B per
ray) and integrates only the ray segments and samples inside its own regions (`GpuMap.setRegionOwnership`).  No voxel
ever crosses a link and the union of the ranks' regions is bit-identical to one map integrating the gathered stream,
because a voxel's update sequence depends only on the rays that reach it, in order.  The line-walk work (the dominant
kernel) is partitioned, the per-ray front half (set-up and binning) is repeated on every rank.

Partitioned map -- the default of `bench.py --gpus N` (`RegionPartition`, `territories_from_origins`,
`exchange_routed_rays`, `PartitionedIntegrator`): owner-computes with the rays ROUTED instead of all-gathered.  Every
rank owns a spatially coherent territory of region blocks (a table: each block belongs to the rank whose sensor origin
is nearest), the library finds for every local ray the ranks owning a region its walk touches -- exactly, with the
integration's own enumeration (`GpuMap.routeRays`) -- and the ranks exchange only those rays (one all-to-all of 48 B per
routed ray; C4: a quarter of the rays cross into a neighbour's territory).  Each rank integrates what is addressed to it
in (source rank, ray) order, so the union of the ranks' regions is bit-identical to one map integrating rank 0's batch,
then rank 1's, ... -- clamps included, every map type.

The functions below are backend agnostic (any torch device / process group) so the protocol itself is covered by
world_size-2 gloo tests on CPU (tests/test_distributed_cpu.py).
"""
import numpy as np


def shard_rays_by_origin(rays, world_size, rank):
    """Partition a (2N, 3) origin/sample array by sensor-origin region: rays are grouped by their origin's 3.2 m cell
    (32 voxels of 0.1 m) and cells are dealt to ranks round-robin in sorted order, so all rays of one sensor origin
    land on one rank.  Deterministic and identical on every rank."""
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 3)
    origins = rays[0::2]
    cells = np.floor(origins / 3.2 + 0.5).astype(np.int64)
    uniq, inverse = np.unique(cells, axis=0, return_inverse=True)
    owner = np.arange(len(uniq)) % world_size
    keep = owner[inverse.reshape(-1)] == rank
    out = np.empty((2 * int(keep.sum()), 3), dtype=np.float64)
    out[0::2] = origins[keep]
    out[1::2] = rays[1::2][keep]
    return out


def _pack_keys(keys):
    """int16 (n, 3) region keys -> sortable int64."""
    k = np.asarray(keys, dtype=np.int64).reshape(-1, 3)
    return ((k[:, 0] + 32768) << 32) | ((k[:, 1] + 32768) << 16) | (k[:, 2] + 32768)


def _unpack_keys(packed):
    p = np.asarray(packed, dtype=np.int64)
    return np.stack([(p >> 32) - 32768, ((p >> 16) & 0xFFFF) - 32768, (p & 0xFFFF) - 32768], axis=1).astype(np.int16)


def union_region_keys(local_keys, group=None):
    """All-gather the ranks' touched-region key lists and return the sorted union (identical on every rank)."""
    import torch
    import torch.distributed as dist
    packed = torch.from_numpy(np.ascontiguousarray(_pack_keys(local_keys)))
    world = dist.get_world_size(group)
    gathered = [None] * world
    dist.all_gather_object(gathered, packed.numpy().tolist(), group=group)
    union = sorted(set(k for lst in gathered for k in lst))
    return _unpack_keys(np.array(union, dtype=np.int64)).reshape(-1, 3)


def merge_occupancy_deltas(base_tiles, local_tiles, min_value, max_value, group=None):
    """Additive log-odds merge of one region set.

    base_tiles, local_tiles: float32 tensors (n_union_regions, region_voxels) on any device, rows in the SAME (union)
    order on every rank; rows a rank did not touch must equal its base rows.  Returns the merged tensor (same on every
    rank).  One all-reduce(sum) of 2 * n * V floats (delta + observed count)."""
    import torch
    import torch.distributed as dist
    inf = float("inf")
    base_obs = base_tiles != inf
    local_obs = local_tiles != inf
    base0 = torch.where(base_obs, base_tiles, torch.zeros_like(base_tiles))
    local0 = torch.where(local_obs, local_tiles, torch.zeros_like(local_tiles))
    # delta of this rank; a voxel the rank did not observe contributes nothing
    delta = torch.where(local_obs, local0 - base0, torch.zeros_like(base0))
    payload = torch.stack([delta, local_obs.to(delta.dtype)])
    dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=group)
    merged = torch.clamp(base0 + payload[0], min=min_value, max=max_value)
    observed = payload[1] > 0
    return torch.where(observed, merged, torch.full_like(merged, inf))


def region_owner(keys, world_size, block_shift=0):
    """Owner rank of each int16 (n, 3) region key: the library's own rule (ohmhip_region_owner), so host code, tests
    and kernels cannot disagree."""
    import ctypes as C
    from . import _lib as L
    keys = np.ascontiguousarray(keys, dtype=np.int16).reshape(-1, 3)
    owners = np.zeros(len(keys), dtype=np.uint32)
    L.check(L.lib.ohmhip_region_owner(keys.ctypes.data, len(keys), int(block_shift), int(world_size),
                                      owners.ctypes.data), "region_owner")
    return owners


def gather_rays(local_rays, group=None):
    """All-gather the ranks' ray batches into ONE stream ordered by rank (then by ray): torch tensor (2N_r, 3) float64 on
    any device -> (sum_r 2N_r, 3) on the same device.  Ranks may hold different counts (padded to the maximum for the
    collective).  RCCL all_gather_into_tensor for device tensors, gloo on CPU."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    local = local_rays.reshape(-1, 3).contiguous()
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[dist.get_rank(group)] = local.shape[0]
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    counts = [int(c) for c in counts.tolist()]
    longest = max(counts)
    if longest == 0:
        return local
    padded = torch.zeros((longest, 3), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty((world * longest, 3), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    if all(c == longest for c in counts):
        return out
    return torch.cat([out[r * longest:r * longest + counts[r]] for r in range(world)])


class OwnerComputesIntegrator:
    """Exact multi-rank integration: usage on every rank
        integ = OwnerComputesIntegrator(gpu_map); integ.integrateRays(local_rays_tensor, flags)
    `gpu_map` (GpuMap / GpuNdtMap / GpuTsdfMap) must be empty; it ends up holding this rank's regions of the map a
    single device would have built from the rank-ordered stream of all ranks' batches."""

    def __init__(self, gpu_map, group=None, block_shift=0):
        import torch.distributed as dist
        self.gpu_map = gpu_map
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        gpu_map.setRegionOwnership(self.world, self.rank, block_shift)

    def integrateRays(self, local_rays, ray_update_flags=0):
        import torch
        stream = gather_rays(local_rays, self.group)
        if stream.is_cuda:
            torch.cuda.current_stream().synchronize()  # the map integrates on its own HIP stream
            self.gpu_map.wait()                        # the previous gathered batch may still be being read
            self._inflight = stream                    # keep the device memory alive while the batch runs
            return self.gpu_map.integrateRaysDevice(stream.data_ptr(), stream.shape[0], ray_update_flags)
        return self.gpu_map.integrateRays(stream.numpy(), ray_update_flags=ray_update_flags)


class _DeviceArray:
    """Expose a raw HIP device pointer through __cuda_array_interface__ so torch can alias it (no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def occupancy_tensor(gpu_map):
    """The map's resident occupancy layer as a (region_capacity_in_use, region_voxels) float32 torch tensor aliasing
    device memory (valid until the pool is re-allocated by growth)."""
    import ctypes as C
    import torch
    from . import _lib as L
    gpu_map.wait()
    ptr = L._vp()
    stride = C.c_size_t(0)
    L.check(L.lib.ohmhip_map_device_layer_ptr(gpu_map._handle, L.LID_OCCUPANCY, C.byref(ptr), C.byref(stride)))
    n = C.c_size_t(0)
    L.check(L.lib.ohmhip_map_region_count(gpu_map._handle, C.byref(n)))
    voxels = stride.value // 4
    return torch.as_tensor(_DeviceArray(ptr.value, (max(n.value, 1), voxels), "<f4"), device="cuda")


class Communicator:
    """RCCL communicator owned by the library (include/ohmhip.h: ohmhip_comm_*).  The unique id travels over whatever
    control plane the ranks already share -- here a torch.distributed process group (any backend)."""

    def __init__(self, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib as L
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        uid = (C.c_ubyte * L.COMM_ID_BYTES)()
        if self.rank == 0:
            L.check(L.lib.ohmhip_comm_unique_id(uid), "comm_unique_id")
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = (C.c_ubyte * L.COMM_ID_BYTES).from_buffer_copy(box[0])
        self._handle = L._vp()
        L.check(L.lib.ohmhip_comm_init_rank(C.byref(self._handle), uid, self.world, self.rank), "comm_init_rank")

    def close(self):
        from . import _lib as L
        if self._handle:
            L.lib.ohmhip_comm_destroy(self._handle)
            self._handle = None


class ReplicaMerger:
    """Replica merge of a GpuMap's occupancy layer (include/ohmhip.h "Replica merge").  The map keeps, per region, the
    state ALL replicas share (`base`: identical on every rank, changed only when the region is exchanged -- and then on
    every rank) and the set of PENDING regions (modified since they were last exchanged).  merge() exchanges the regions
    pending on more than one rank (`full_union=True`: on any rank, which keeps all replicas bit-identical maps); a
    region pending on one rank only stays pending, on its shared base, until a second rank reaches it.

    usage:  merger = ReplicaMerger(gpu_map, comm=Communicator())   # RCCL inside the library, everything on device
            ... integrateRays on every rank ...;  stats = merger.merge()
    Without `comm` the same steps run over a torch.distributed group of any backend (gloo in the tests): key lists by
    all_gather_object, payloads by all_reduce on host copies of the library's device buffers.  Either way the ranks
    agree on the outcome of their local steps BEFORE the payload collective, so a rank-local failure raises on every
    rank (OhmHipError with ERR_PEER on the ranks that did not fail themselves) instead of leaving the peers blocked."""

    def __init__(self, gpu_map, group=None, comm=None, full_union=False):
        from . import _lib as L
        self.gpu_map = gpu_map
        self.group = group
        self.comm = comm
        self.full_union = bool(full_union)
        L.check(L.lib.ohmhip_map_enable_merge(gpu_map._handle), "enable_merge")
        L.check(L.lib.ohmhip_map_set_merge_mode(gpu_map._handle,
                                                L.MERGE_FULL_UNION if full_union else L.MERGE_SHARED_ONLY), "merge_mode")

    def local_keys(self):
        import ctypes as C
        from . import _lib as L
        n = C.c_size_t(0)
        L.check(L.lib.ohmhip_map_merge_keys(self.gpu_map._handle, None, 0, C.byref(n)), "merge_keys")
        keys = np.zeros((max(n.value, 1), 3), dtype=np.int16)
        L.check(L.lib.ohmhip_map_merge_keys(self.gpu_map._handle, keys.ctypes.data, n.value, C.byref(n)), "merge_keys")
        return keys[:n.value]

    def merge(self):
        """Collective.  Returns the merge statistics (regions_local / regions_union / regions_shared / payload_bytes)."""
        import ctypes as C
        from . import _lib as L
        if self.comm is not None:
            st = L.MergeStats()
            L.check(L.lib.ohmhip_map_merge_replicas(self.gpu_map._handle, self.comm._handle, C.byref(st)),
                    "merge_replicas")
            return {name: getattr(st, name) for name, _ in L.MergeStats._fields_}
        return self._merge_over_group()

    def _merge_over_group(self):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib as L
        gm = self.gpu_map
        world = dist.get_world_size(self.group)
        # local, fallible: the pending list.  A failure is carried into the first collective instead of raised here.
        local_err, local = None, np.zeros((0, 3), dtype=np.int16)
        try:
            local = self.local_keys()
        except Exception as exc:
            local_err = exc
        gathered = [None] * world
        dist.all_gather_object(gathered, None if local_err else _pack_keys(local).tolist(), group=self.group)
        if any(lst is None for lst in gathered):
            raise local_err if local_err else L.OhmHipError(L.ERR_PEER, "merge_keys failed on another rank")
        counts = {}
        for lst in gathered:
            for k in lst:
                counts[k] = counts.get(k, 0) + 1
        need = 0 if self.full_union else 1
        shared = _unpack_keys(np.array(sorted(k for k, c in counts.items() if c > need), dtype=np.int64)).reshape(-1, 3)
        shared = np.ascontiguousarray(shared, dtype=np.int16)
        n = len(shared)
        voxels = n * int(np.prod(gm._map.region_voxel_dimensions))
        delta = obs = None
        h_delta = np.zeros(voxels, dtype=np.float32)
        h_obs = np.zeros(voxels, dtype=np.uint8)
        if n:
            try:
                delta, obs = L._vp(), L._vp()
                L.check(L.lib.ohmhip_buffer_create(C.byref(delta), 4 * voxels, 3), "buffer_create")
                L.check(L.lib.ohmhip_buffer_create(C.byref(obs), voxels, 3), "buffer_create")
                d_delta, d_obs = L._vp(), L._vp()
                L.check(L.lib.ohmhip_buffer_ptr(delta, C.byref(d_delta)))
                L.check(L.lib.ohmhip_buffer_ptr(obs, C.byref(d_obs)))
                L.check(L.lib.ohmhip_map_merge_pack(gm._handle, shared.ctypes.data, n, d_delta, d_obs), "merge_pack")
                # reduce through host tensors: works for every backend (the RCCL path of the library stays on the device)
                L.check(L.lib.ohmhip_buffer_read(delta, h_delta.ctypes.data, 4 * voxels, 0, None, None, None))
                L.check(L.lib.ohmhip_buffer_read(obs, h_obs.ctypes.data, voxels, 0, None, None, None))
            except Exception as exc:
                local_err = exc
        # the ranks agree on the outcome of the local steps before the payload collective
        status = torch.tensor([1 if local_err else 0], dtype=torch.int32)
        dist.all_reduce(status, op=dist.ReduceOp.MAX, group=self.group)
        if int(status.item()):
            for b in (delta, obs):
                if b:
                    L.lib.ohmhip_buffer_destroy(b)
            raise local_err if local_err else L.OhmHipError(L.ERR_PEER, "merge_pack failed on another rank")
        if n:
            t_delta = torch.from_numpy(h_delta)
            t_obs = torch.from_numpy(h_obs)
            dist.all_reduce(t_delta, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(t_obs, op=dist.ReduceOp.MAX, group=self.group)  # as the library: only non-zero matters
            L.check(L.lib.ohmhip_buffer_write(delta, t_delta.numpy().ctypes.data, 4 * voxels, 0, None, None, None))
            L.check(L.lib.ohmhip_buffer_write(obs, t_obs.numpy().ctypes.data, voxels, 0, None, None, None))
            L.check(L.lib.ohmhip_map_merge_apply(gm._handle, shared.ctypes.data, n, d_delta, d_obs), "merge_apply")
            L.lib.ohmhip_buffer_destroy(delta)
            L.lib.ohmhip_buffer_destroy(obs)
        gm.wait()
        return {"regions_local": len(local), "regions_union": len(counts), "regions_shared": n,
                "payload_bytes": 5 * voxels}


def merge_in_process(gpu_maps, full_union=False, chunk_regions=256):
    """The replica merge for several merge-enabled GpuMaps living in ONE process (stand-ins for ranks on a single GPU:
    tests, and the C4 deviation figure of bench.py at --gpus 1).  Same library steps as ReplicaMerger -- merge_keys ->
    agreed list -> merge_pack on every map -> payloads summed (deltas, in rank order) / max-ed (observer flags) on the
    host -> merge_apply on EVERY map.  Returns (exchanged_keys (n, 3) int16, stats dict)."""
    import ctypes as C
    from . import _lib as L
    pend = []
    for gm in gpu_maps:
        n = C.c_size_t(0)
        L.check(L.lib.ohmhip_map_merge_keys(gm._handle, None, 0, C.byref(n)), "merge_keys")
        keys = np.zeros((max(n.value, 1), 3), dtype=np.int16)
        L.check(L.lib.ohmhip_map_merge_keys(gm._handle, keys.ctypes.data, n.value, C.byref(n)), "merge_keys")
        pend.append(_pack_keys(keys[:n.value]).tolist())
    counts = {}
    for lst in pend:
        for k in lst:
            counts[k] = counts.get(k, 0) + 1
    need = 0 if full_union else 1
    picked = np.array(sorted(k for k, c in counts.items() if c > need), dtype=np.int64)
    shared = np.ascontiguousarray(_unpack_keys(picked).reshape(-1, 3), dtype=np.int16)
    voxels_per_region = int(np.prod(gpu_maps[0]._map.region_voxel_dimensions))
    for at in range(0, len(shared), chunk_regions):
        part = np.ascontiguousarray(shared[at:at + chunk_regions])
        n = len(part)
        voxels = n * voxels_per_region
        delta, obs, d_delta, d_obs = L._vp(), L._vp(), L._vp(), L._vp()
        L.check(L.lib.ohmhip_buffer_create(C.byref(delta), 4 * voxels, 3), "buffer_create")
        L.check(L.lib.ohmhip_buffer_create(C.byref(obs), voxels, 3), "buffer_create")
        L.check(L.lib.ohmhip_buffer_ptr(delta, C.byref(d_delta)))
        L.check(L.lib.ohmhip_buffer_ptr(obs, C.byref(d_obs)))
        d_sum = np.zeros(voxels, dtype=np.float32)
        o_max = np.zeros(voxels, dtype=np.uint8)
        h_d = np.zeros(voxels, dtype=np.float32)
        h_o = np.zeros(voxels, dtype=np.uint8)
        for gm in gpu_maps:
            L.check(L.lib.ohmhip_map_merge_pack(gm._handle, part.ctypes.data, n, d_delta, d_obs), "merge_pack")
            L.check(L.lib.ohmhip_buffer_read(delta, h_d.ctypes.data, 4 * voxels, 0, None, None, None))
            L.check(L.lib.ohmhip_buffer_read(obs, h_o.ctypes.data, voxels, 0, None, None, None))
            d_sum += h_d
            np.maximum(o_max, h_o, out=o_max)
        L.check(L.lib.ohmhip_buffer_write(delta, d_sum.ctypes.data, 4 * voxels, 0, None, None, None))
        L.check(L.lib.ohmhip_buffer_write(obs, o_max.ctypes.data, voxels, 0, None, None, None))
        for gm in gpu_maps:
            L.check(L.lib.ohmhip_map_merge_apply(gm._handle, part.ctypes.data, n, d_delta, d_obs), "merge_apply")
        L.lib.ohmhip_buffer_destroy(delta)
        L.lib.ohmhip_buffer_destroy(obs)
    for gm in gpu_maps:
        gm.wait()
    return shared, {"regions_pending": [len(p) for p in pend], "regions_union": len(counts),
                    "regions_shared": len(shared), "payload_bytes_per_rank": 5 * len(shared) * voxels_per_region}


def merge_deviation(merged_chunks, sequential_chunks, keys=None, rel=1e-5):
    """Replica-merged occupancy vs the SEQUENTIAL integration of the same rays (the additive rule is exact
    only while no clamp engages between the shards' updates -- 'must be stated with results').  Both arguments are
    {key: {'occupancy': float32[...]}} host maps; `keys` restricts the comparison (default: regions of both).
    Counts voxels whose observed-state differs, whose values differ at all (float summation order counts) and beyond
    `rel`, and the largest |difference|."""
    if keys is None:
        keys = sorted(set(merged_chunks) & set(sequential_chunks))
    out = {"regions_compared": 0, "voxels_compared": 0, "voxels_observed": 0, "voxels_state_differs": 0,
           "voxels_value_differs": 0, "voxels_beyond_rel": 0, "max_abs_delta": 0.0, "rel": rel}
    for key in keys:
        key = tuple(int(v) for v in key)
        a = merged_chunks[key]["occupancy"].reshape(-1)
        b = sequential_chunks[key]["occupancy"].reshape(-1)
        fa, fb = np.isfinite(a), np.isfinite(b)
        both = fa & fb
        d = np.abs(a[both].astype(np.float64) - b[both].astype(np.float64))
        out["regions_compared"] += 1
        out["voxels_compared"] += int(a.size)
        out["voxels_observed"] += int(fb.sum())
        out["voxels_state_differs"] += int((fa != fb).sum())
        out["voxels_value_differs"] += int((d > 0).sum())
        out["voxels_beyond_rel"] += int((d > rel * np.maximum(1.0, np.abs(b[both]))).sum())
        if d.size:
            out["max_abs_delta"] = max(out["max_abs_delta"], float(d.max()))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Partitioned map: territories, ray routing, all-to-all of the routed rays.
# ---------------------------------------------------------------------------------------------------------------------
class RegionPartition:
    """Who owns which region among `world_size` ranks (include/ohmhip.h: ohmhip_partition): blocks of 2^block_shift
    regions per axis dealt by a table over the block grid [grid_origin, grid_origin + table.shape) -- `table[x, y, z]`
    = owner rank, blocks outside the grid go to the nearest cell's owner -- or, with table=None, by the library's block
    hash.  Identical on every rank apart from `rank`."""

    def __init__(self, world_size, rank, block_shift=0, grid_origin=(0, 0, 0), table=None):
        self.world_size = int(world_size)
        self.rank = int(rank)
        self.block_shift = int(block_shift)
        self.grid_origin = tuple(int(v) for v in grid_origin)
        self.table = None if table is None else np.ascontiguousarray(table, dtype=np.uint8)
        if self.table is not None:
            assert self.table.ndim == 3 and self.table.size > 0 and int(self.table.max()) < self.world_size
            # the C ABI wants x fastest
            self._flat = np.ascontiguousarray(np.transpose(self.table, (2, 1, 0))).reshape(-1)
        else:
            self._flat = None

    def c_struct(self, self_rank=True):
        from . import _lib as L
        p = L.Partition()
        p.world_size = self.world_size
        p.rank = self.rank if self_rank else 0
        p.block_shift = self.block_shift
        for a in range(3):
            p.grid_origin[a] = self.grid_origin[a]
            p.grid_dims[a] = 0 if self.table is None else self.table.shape[a]
        p.owners = None if self._flat is None else self._flat.ctypes.data
        return p

    def owners(self, keys):
        """Owner rank of each int16 (n, 3) region key (the library's rule, evaluated on the host; no device)."""
        from . import _lib as L
        import ctypes as C
        keys = np.ascontiguousarray(keys, dtype=np.int16).reshape(-1, 3)
        out = np.zeros(len(keys), dtype=np.uint32)
        L.check(L.lib.ohmhip_partition_owners(C.byref(self.c_struct()), keys.ctypes.data, len(keys), out.ctypes.data),
                "partition_owners")
        return out

    def with_rank(self, rank):
        return RegionPartition(self.world_size, rank, self.block_shift, self.grid_origin, self.table)


def territories_from_origins(origins, world_size, rank, region_size, block_shift=1, margin=40.0, map_origin=(0, 0, 0)):
    """A RegionPartition that gives every block of 2^block_shift regions to the rank whose sensor origin is nearest to
    the block's centre (ties: the lower rank) -- spatially coherent territories, so only rays that reach into a
    neighbour's territory travel.  `origins`: one (x, y, z) per rank, or a list of points per rank; `region_size`: a
    region's edge in metres per axis (resolution x region voxel dimensions); the table covers the origins' bounding box
    grown by `margin` metres, everything beyond belongs to the nearest cell's owner.  Deterministic: every rank
    computes the same table."""
    pts, owner_of = [], []
    for r, o in enumerate(origins):
        o = np.asarray(o, dtype=np.float64).reshape(-1, 3)
        pts.append(o)
        owner_of += [r % world_size] * len(o)
    pts = np.concatenate(pts)
    owner_of = np.asarray(owner_of, dtype=np.int64)
    size = np.broadcast_to(np.asarray(region_size, dtype=np.float64), (3,))
    mo = np.asarray(map_origin, dtype=np.float64)
    block = size * (1 << block_shift)
    # region r spans [r - 1/2, r + 1/2) x size + map origin (ohm/OccupancyMap.h:757-778): block b = regions
    # [b * 2^shift, (b + 1) * 2^shift) spans [(b * 2^shift - 1/2) * size, ...)
    def block_of(p):
        return np.floor(((p - mo) / size + 0.5) / (1 << block_shift)).astype(np.int64)
    lo = block_of(pts.min(axis=0) - margin)
    hi = block_of(pts.max(axis=0) + margin)
    dims = (hi - lo + 1).astype(np.int64)
    ix, iy, iz = np.meshgrid(np.arange(dims[0]), np.arange(dims[1]), np.arange(dims[2]), indexing="ij")
    cells = np.stack([ix, iy, iz], axis=-1) + lo
    centres = ((cells + 0.5) * (1 << block_shift) - 0.5) * size + mo
    d2 = ((centres[..., None, :] - pts[None, None, None, :, :]) ** 2).sum(axis=-1)
    # nearest origin, ties to the lower rank: order the points by rank first (stable argmin picks the first minimum)
    order = np.argsort(owner_of, kind="stable")
    nearest = order[np.argmin(d2[..., order], axis=-1)]
    table = owner_of[nearest].astype(np.uint8)
    return RegionPartition(world_size, rank, block_shift, tuple(int(v) for v in lo), table)


def exchange_routed_rays(routed, counts, group=None):
    """All-to-all of routed rays over a torch.distributed group (RCCL for device tensors, gloo on the CPU): `routed` is a
    (sum(counts), 6) float64 tensor holding this rank's rays per destination, blocks back to back in rank order;
    returns (received (k, 6) tensor on the same device, rays per source rank) -- the rays addressed to this rank in
    (source rank, ray) order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    send = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=routed.device)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_counts = [int(c) for c in recv.tolist()]
    # (a rank whose local step failed sends -1 to EVERY rank -- PartitionedIntegrator._announce_failure -- so every rank
    # sees it here, in the same call, and none goes on to the payload exchange)
    if min(recv_counts) < 0:
        raise RuntimeError("partitioned step: a peer rank reported a local failure (OHMHIP_ERR_PEER); no rays exchanged")
    out = torch.empty((sum(recv_counts), 6), dtype=routed.dtype, device=routed.device)
    dist.all_to_all_single(out.view(-1), routed.reshape(-1, 6)[:int(sum(int(c) for c in counts))].reshape(-1),
                           [6 * c for c in recv_counts], [6 * int(c) for c in counts], group=group)
    assert len(recv_counts) == world
    return out, recv_counts


def exchange_routed_side(block, counts, recv_counts, group=None):
    """The all-to-all of exchange_routed_rays for a per-ray side array (time stamps: float64, intensities: float32): `block`
    is a 1-D tensor in routed order, `counts` / `recv_counts` the rays per destination / source the ray exchange used."""
    import torch
    import torch.distributed as dist
    out = torch.empty((int(sum(recv_counts)),), dtype=block.dtype, device=block.device)
    dist.all_to_all_single(out, block[:int(sum(int(c) for c in counts))].contiguous(), [int(c) for c in recv_counts],
                           [int(c) for c in counts], group=group)
    return out


class PartitionedIntegrator:
    """Exact multi-rank integration into a region-partitioned map.  Usage on every rank:
        part = territories_from_origins(all_origins, world, rank, region_size)
        integ = PartitionedIntegrator(gpu_map, part);  integ.integrateRays(local_rays_device_tensor)
    `gpu_map` must be empty; it ends up holding this rank's territory of the map ONE device would have built from
    rank 0's batch, then rank 1's, ... of every call.  `comm` (Communicator): the exchange runs inside the library over
    RCCL; otherwise over the torch.distributed group (RCCL for device tensors under the nccl backend, host-staged under
    gloo)."""

    def __init__(self, gpu_map, partition, group=None, comm=None):
        self.gpu_map = gpu_map
        self.partition = partition
        self.group = group
        self.comm = comm
        gpu_map.setRegionPartition(partition)
        self._routed = None
        # A batch reads its rays until it ends and the map keeps two batches in flight (include/ohmhip.h,
        # ohmhip_map_integrate_rays_device): three receive buffers used in turn are never written while a batch reads
        # them, so a step does not wait for the previous one -- the next batch is routed (the library routes on a stream
        # of its own) and exchanged while the previous one integrates.
        # The turn advances per LAUNCH, not per call: a step that receives no rays, or whose rays the map only
        # collects (batch coalescing), launches nothing, and the two batches launched last may both still be reading
        # their buffers.  Each buffer remembers the launch count its batch made; it is free once two more were launched.
        self._recv = [None, None, None]
        self._recv_batch = [0, 0, 0]
        # side arrays (time stamps, intensities) travel with the rays: routed-order staging and per-slot receive buffers
        self._index = None
        self._time_base_set = False
        self._side_send = {}
        self._side_recv = [{}, {}, {}]
        self._xstream = None
        self.last = {}

    def close(self):
        from . import _lib as L
        if self._xstream is not None:
            L.lib.ohmhip_stream_destroy(self._xstream)
            self._xstream = None

    def _exchange_stream(self):
        import ctypes as C
        from . import _lib as L
        if self._xstream is None:
            handle = L._vp()
            L.check(L.lib.ohmhip_stream_create(C.byref(handle)), "stream_create")
            self._xstream = handle
        return self._xstream

    def _ensure_routed(self, rays):
        import torch
        t = self._routed
        if t is None or t.shape[0] < rays:
            t = torch.empty((max(int(rays * 1.25) + 1024, 4096), 6), dtype=torch.float64, device="cuda")
            self._routed = t
        return t

    def _ensure_recv(self, slot, rays):
        import torch
        t = self._recv[slot]
        if t is None or t.shape[0] < rays:
            # (the batch that read the old buffer -- three calls ago -- has ended: it may go)
            t = torch.empty((max(int(rays * 1.25) + 1024, 4096), 6), dtype=torch.float64, device="cuda")
            self._recv[slot] = t
        return t

    def route(self, d_rays, n_rays, ray_update_flags=0, with_index=False):
        """Route `n_rays` rays at device pointer `d_rays`; returns (routed tensor, counts, visits).  with_index: the
        index of every routed ray in the input is kept in self._index (device int32 tensor) for the side arrays."""
        import torch
        routed = self._ensure_routed(2 * n_rays)
        while True:
            d_index = None
            if with_index:
                if self._index is None or self._index.shape[0] < routed.shape[0]:
                    self._index = torch.empty((routed.shape[0],), dtype=torch.int32, device="cuda")
                d_index = self._index.data_ptr()
            counts, visits, fits = self.gpu_map.routeRays(d_rays, n_rays, routed.data_ptr(), routed.shape[0],
                                                          ray_update_flags, d_index=d_index)
            if fits:
                return routed, counts, visits
            routed = self._ensure_routed(int(counts.sum()))

    def _announce_failure(self, counts):
        """Take part in the step's count exchange with the failure marker (every rank then leaves the step together)."""
        from . import _lib as L
        marker = np.full(len(counts), 0xffffffff, dtype=np.uint32)
        if self.comm is not None:
            got = np.zeros_like(marker)
            L.lib.ohmhip_comm_exchange_counts(self.comm._handle, marker.ctypes.data, got.ctypes.data,
                                              self._exchange_stream())
        else:
            import torch
            import torch.distributed as dist
            device = "cpu" if dist.get_backend(self.group) == "gloo" else "cuda"
            send = torch.full((len(counts),), -1, dtype=torch.int64, device=device)
            dist.all_to_all_single(torch.empty_like(send), send, group=self.group)

    def _agree_on_time_base(self, d_first_stamp):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib as L
        first = np.array([np.nan], dtype=np.float64)
        if d_first_stamp is not None:
            tmp = L._vp()
            L.check(L.lib.ohmhip_buffer_create(C.byref(tmp), 8, 3), "buffer_create")
            ptr = L._vp()
            L.check(L.lib.ohmhip_buffer_ptr(tmp, C.byref(ptr)))
            idx = torch.zeros((1,), dtype=torch.int32, device="cuda")
            torch.cuda.current_stream().synchronize()
            L.check(L.lib.ohmhip_gather_rows(d_first_stamp, idx.data_ptr(), 1, 8, ptr, None), "gather_rows")
            L.check(L.lib.ohmhip_device_synchronize(), "device_synchronize")
            L.check(L.lib.ohmhip_buffer_read(tmp, first.ctypes.data, 8, 0, None, None, None), "buffer_read")
            L.lib.ohmhip_buffer_destroy(tmp)
        # A base a map already holds (set by the caller, or taken from an earlier un-partitioned batch) is never overwritten
        # (ADVICE r5): it travels in the same all-gather, and where any rank holds one the lowest such rank's base is what
        # the ranks WITHOUT one adopt -- like integrate_partitioned_in_process, which only sets a base that is < 0.
        held = self.gpu_map.firstRayTime()
        pair = np.array([first[0], held if held >= 0 else np.nan], dtype=np.float64)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            device = "cpu" if dist.get_backend(self.group) == "gloo" else "cuda"
            mine = torch.tensor(pair, dtype=torch.float64, device=device)
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=self.group)
            pairs = [(float(t[0].item()), float(t[1].item())) for t in every]
        else:
            pairs = [(float(pair[0]), float(pair[1]))]
        bases = [b for _, b in pairs if b == b]
        known = [t for t, _ in pairs if t == t]  # (NaN: that rank has no rays in this batch)
        agreed = bases[0] if bases else (known[0] if known else None)
        if agreed is not None:
            if held < 0:
                self.gpu_map.setFirstRayTime(agreed)
            self._time_base_set = True

    def _side_buffer(self, store, name, rays, dtype):
        import torch
        t = store.get(name)
        if t is None or t.shape[0] < rays or t.dtype != dtype:
            t = torch.empty((max(int(rays * 1.25) + 1024, 4096),), dtype=dtype, device="cuda")
            store[name] = t
        return t

    def integrateRays(self, local_rays, ray_update_flags=0):
        """local_rays: (2N, 3) float64 CUDA tensor (origin, sample pairs).  Collective.  Returns the number of points
        (2 x rays, like GpuMap.integrateRays) this rank integrated: its own rays and received ones that pass the ray
        filter."""
        import torch
        local = local_rays.reshape(-1, 6)
        torch.cuda.current_stream().synchronize()  # the library works on its own HIP streams
        return self.integrateRaysDevice(local.data_ptr(), 2 * local.shape[0], ray_update_flags)

    def integrateRaysDevice(self, d_rays_ptr, element_count, ray_update_flags=0, d_timestamps=None, d_intensities=None):
        """The same for a raw device pointer to element_count dvec3 (complete when the call is made; free again when the
        call returns).  d_timestamps / d_intensities: device pointers to one double / float per local ray; they are put
        into routed order (ohmhip_gather_rows with the routing's index list), exchanged with the rays and handed to the
        map with what arrives -- a partitioned NDT-TM map gets its intensities, a touch-time layer its stamps, exactly as
        one map integrating the ranks' batches one after the other would.  The batch the call launches is left in flight:
        GpuMap.wait() / syncVoxels() settle it."""
        import torch
        from . import _lib as L
        gm = self.gpu_map
        n_local = int(element_count) // 2
        sides = [(name, ptr, dtype, size) for name, ptr, dtype, size in
                 (("timestamps", d_timestamps, torch.float64, 8), ("intensities", d_intensities, torch.float32, 4))
                 if ptr is not None]
        if d_timestamps is not None and not self._time_base_set:
            # One time base for the whole partitioned map (OccupancyMap::firstRayTime, the touch-time layer's zero): the
            # first stamp of the lowest rank that has rays -- what ONE map integrating rank 0's batch, then rank 1's, ...
            # would have taken.  Collective (an all-gather of two doubles per rank): EVERY rank must pass the side arrays
            # consistently -- a rank that omits timestamps while its peers pass them does not enter this collective.
            self._agree_on_time_base(d_timestamps if n_local else None)
        launched = gm.batchesLaunched()
        # (when the previous integrate call returned, every batch but the two launched last had ended)
        slot = next(i for i in range(3) if self._recv_batch[i] == 0 or self._recv_batch[i] + 2 <= launched)
        self._recv_batch[slot] = 0
        # A rank-local failure before the exchange (routing, buffers) must not leave the peers blocked in the collective:
        # the rank still takes part in the COUNT exchange, with the failure marker instead of counts, so that every rank
        # raises from this same call (include/ohmhip.h "FAILURE AGREEMENT").
        local_failure = None
        try:
            routed, counts, visits = self.route(d_rays_ptr, n_local, ray_update_flags, with_index=bool(sides))
        except Exception as exc:  # noqa: BLE001 -- re-raised below, after the peers have been told
            local_failure = exc
            world_size = self.partition.world_size
            routed, counts, visits = self._ensure_routed(1), np.zeros(world_size, dtype=np.uint32), 0
        if local_failure is not None:
            self._announce_failure(counts)
            raise local_failure
        sent = int(counts.sum())
        # side arrays into routed order (the library's gather kernel on the exchange stream)
        side_send = {}
        xs = self._exchange_stream() if (self.comm is not None or sides) else None
        for name, ptr, dtype, size in sides:
            buf = self._side_buffer(self._side_send, name, sent, dtype)
            L.check(L.lib.ohmhip_gather_rows(ptr, self._index.data_ptr(), sent, size, buf.data_ptr(), xs), "gather_rows")
            side_send[name] = buf
        if sides:
            L.check(L.lib.ohmhip_stream_finish(xs), "stream_finish")
        side_recv = {}
        if self.comm is not None:
            send_counts = np.ascontiguousarray(counts, dtype=np.uint32)
            recv_counts = np.zeros_like(send_counts)
            L.check(L.lib.ohmhip_comm_exchange_counts(self.comm._handle, send_counts.ctypes.data,
                                                      recv_counts.ctypes.data, xs), "exchange_counts")
            n_recv = int(recv_counts.sum())
            recv = self._ensure_recv(slot, n_recv)
            L.check(L.lib.ohmhip_comm_exchange_rays(self.comm._handle, routed.data_ptr(), send_counts.ctypes.data,
                                                    recv.data_ptr(), recv_counts.ctypes.data, xs), "exchange_rays")
            for name, ptr, dtype, size in sides:
                got = self._side_buffer(self._side_recv[slot], name, n_recv, dtype)
                L.check(L.lib.ohmhip_comm_exchange_side(self.comm._handle, side_send[name].data_ptr(),
                                                        send_counts.ctypes.data, got.data_ptr(),
                                                        recv_counts.ctypes.data, size, xs), "exchange_side")
                side_recv[name] = got
            L.check(L.lib.ohmhip_stream_finish(xs), "stream_finish")  # (not the device: batches stay in flight)
            recv_counts = [int(c) for c in recv_counts]
        else:
            import torch.distributed as dist
            block = routed[:sent]
            staged = dist.get_backend(self.group) == "gloo"
            if staged:
                got, recv_counts = exchange_routed_rays(block.cpu(), counts, self.group)
                n_recv = got.shape[0]
                recv = self._ensure_recv(slot, n_recv)
                recv[:n_recv].copy_(got)
            else:
                got, recv_counts = exchange_routed_rays(block, counts, self.group)
                n_recv = got.shape[0]
                recv = got
                self._recv[slot] = got  # stays alive while the batch reads it
            for name, ptr, dtype, size in sides:
                src = side_send[name][:sent]
                arrived = exchange_routed_side(src.cpu() if staged else src, counts, recv_counts, self.group)
                keep = self._side_buffer(self._side_recv[slot], name, n_recv, dtype)
                keep[:n_recv].copy_(arrived)
                side_recv[name] = keep
            torch.cuda.current_stream().synchronize()
        self.last = {"rays_local": n_local, "rays_routed": sent, "rays_kept": int(counts[self.partition.rank]),
                     "rays_received": n_recv, "recv_counts": recv_counts, "visits_local": visits}
        if n_recv == 0:
            return 0
        done = gm.integrateRaysDevice(recv.data_ptr(), 2 * n_recv, ray_update_flags,
                                      d_intensities=side_recv["intensities"].data_ptr() if "intensities" in side_recv else None,
                                      d_timestamps=side_recv["timestamps"].data_ptr() if "timestamps" in side_recv else None)
        after = gm.batchesLaunched()
        # (no launch: the call copied its rays behind the ones waiting and waited for the copy -- the buffer is free)
        self._recv_batch[slot] = after if after > launched else 0
        return done


def integrate_partitioned_in_process(gpu_maps, shards, ray_update_flags=0, timings=None, streams_out=None,
                                     timestamps=None, intensities=None):
    """The partitioned integration for several GpuMaps living in ONE process (stand-ins for ranks on a single GPU: tests,
    and bench.py's one-GPU C4 leg).  gpu_maps[r] carries rank r's partition (setRegionPartition); shards[r]: rank r's
    (2N_r, 3) float64 host rays; timestamps[r] / intensities[r] (optional): rank r's per-ray float64 / float32 side
    arrays.  Every rank's rays are routed by its own map (the library's kernels), its side arrays are put into routed
    order on the device (ohmhip_gather_rows with the routing's index list -- what PartitionedIntegrator does before the
    exchange), the blocks are re-assembled per destination in (source rank, ray) order -- what the all-to-all delivers --
    and integrated.  Returns a dict of counts: rays routed per (source, destination), rays received per rank.
    `streams_out` (a list) receives the stream each rank integrated, as (k, 6) host arrays."""
    import ctypes as C
    import time
    from . import _lib as L
    world = len(gpu_maps)
    blocks = [[None] * world for _ in range(world)]
    side_blocks = {"timestamps": [[None] * world for _ in range(world)], "intensities": [[None] * world for _ in range(world)]}
    side_in = {"timestamps": (timestamps, np.float64), "intensities": (intensities, np.float32)}
    matrix = np.zeros((world, world), dtype=np.int64)
    visits = []
    if timestamps is not None:
        # one time base for all ranks' maps: the first stamp of the lowest rank with rays (PartitionedIntegrator agrees on
        # it with an all-gather); only while none is set yet
        firsts = [float(np.asarray(t).reshape(-1)[0]) for t in timestamps if np.asarray(t).size]
        for gm in gpu_maps:
            if firsts and gm.firstRayTime() < 0:
                gm.setFirstRayTime(firsts[0])

    def device_copy(host):
        buf, ptr = L._vp(), L._vp()
        L.check(L.lib.ohmhip_buffer_create(C.byref(buf), max(host.nbytes, 48), 3), "buffer_create")
        if host.nbytes:
            L.check(L.lib.ohmhip_buffer_write(buf, host.ctypes.data, host.nbytes, 0, None, None, None), "buffer_write")
        L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(ptr)))
        return buf, ptr

    for r, (gm, rays) in enumerate(zip(gpu_maps, shards)):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        src, d_src = device_copy(rays)
        want_sides = [name for name, (arrays, _) in side_in.items() if arrays is not None]
        out = idx = None
        cap = max(2 * n, 1024)
        while True:
            out, d_out = L._vp(), L._vp()
            L.check(L.lib.ohmhip_buffer_create(C.byref(out), 48 * cap, 3), "buffer_create")
            L.check(L.lib.ohmhip_buffer_ptr(out, C.byref(d_out)))
            d_idx = None
            if want_sides:
                idx, d_idx = L._vp(), L._vp()
                L.check(L.lib.ohmhip_buffer_create(C.byref(idx), 4 * cap, 3), "buffer_create")
                L.check(L.lib.ohmhip_buffer_ptr(idx, C.byref(d_idx)))
            if timings is not None:
                gm.routeRays(d_src, n, d_out, cap, ray_update_flags)  # untimed: the map's routing buffers and stream
            t0 = time.perf_counter()
            counts, v, fits = gm.routeRays(d_src, n, d_out, cap, ray_update_flags, d_index=d_idx)
            if timings is not None and fits:
                timings.setdefault("route_ms", []).append(1e3 * (time.perf_counter() - t0))
            if fits:
                break
            L.lib.ohmhip_buffer_destroy(out)
            if idx is not None:
                L.lib.ohmhip_buffer_destroy(idx)
            cap = int(counts.sum()) + 1024
        visits.append(v)
        total = int(counts.sum())
        host = np.zeros((total, 6), dtype=np.float64)
        if total:
            L.check(L.lib.ohmhip_buffer_read(out, host.ctypes.data, host.nbytes, 0, None, None, None), "buffer_read")
        routed_sides = {}
        for name in want_sides:
            arrays, dtype = side_in[name]
            values = np.ascontiguousarray(arrays[r], dtype=dtype).reshape(-1)
            assert values.shape[0] == n, "one %s value per ray" % name
            sbuf, d_side = device_copy(values)
            gathered = np.zeros(total, dtype=dtype)
            gbuf, d_gathered = device_copy(gathered)
            L.check(L.lib.ohmhip_gather_rows(d_side, d_idx, total, values.itemsize, d_gathered, None), "gather_rows")
            L.check(L.lib.ohmhip_device_synchronize(), "device_synchronize")
            if total:
                L.check(L.lib.ohmhip_buffer_read(gbuf, gathered.ctypes.data, gathered.nbytes, 0, None, None, None))
            routed_sides[name] = gathered
            L.lib.ohmhip_buffer_destroy(sbuf)
            L.lib.ohmhip_buffer_destroy(gbuf)
        at = 0
        for d in range(world):
            blocks[r][d] = host[at:at + int(counts[d])]
            for name in want_sides:
                side_blocks[name][r][d] = routed_sides[name][at:at + int(counts[d])]
            matrix[r, d] = int(counts[d])
            at += int(counts[d])
        L.lib.ohmhip_buffer_destroy(src)
        L.lib.ohmhip_buffer_destroy(out)
        if idx is not None:
            L.lib.ohmhip_buffer_destroy(idx)
    integrated = []
    for d, gm in enumerate(gpu_maps):
        stream = np.concatenate([blocks[r][d] for r in range(world)]) if world else np.zeros((0, 6))
        if streams_out is not None:
            streams_out.append(stream)
        if stream.shape[0]:
            buf, ptr = device_copy(stream)
            extra_bufs, extra_ptrs = [], {}
            for name, (arrays, dtype) in side_in.items():
                if arrays is not None:
                    b, p_ = device_copy(np.ascontiguousarray(np.concatenate([side_blocks[name][r][d] for r in range(world)]),
                                                             dtype=dtype))
                    extra_bufs.append(b)
                    extra_ptrs[name] = p_
            gm.wait()
            t0 = time.perf_counter()
            integrated.append(gm.integrateRaysDevice(ptr, 2 * stream.shape[0], ray_update_flags,
                                                     d_intensities=extra_ptrs.get("intensities"),
                                                     d_timestamps=extra_ptrs.get("timestamps")))
            gm.wait()
            if timings is not None:
                timings.setdefault("integrate_ms", []).append(1e3 * (time.perf_counter() - t0))
            L.lib.ohmhip_buffer_destroy(buf)
            for b in extra_bufs:
                L.lib.ohmhip_buffer_destroy(b)
        else:
            integrated.append(0)
    return {"routed": matrix, "received": matrix.sum(axis=0), "integrated": integrated, "visits_local": visits}


def pipelined_rank_step_ms(gpu_map, shard, received, steps=8, warmup=2, ray_update_flags=0):
    """Steady-state time of ONE rank's step without the exchange, on this GPU (bench.py's one-GPU C4 leg): the rank's
    shard is routed (library kernels on the routing stream, host-synchronised as in PartitionedIntegrator) and the stream
    it receives -- `received`, (k, 6) host rays, the same every step here -- is integrated, steps back to back with the
    batches left in flight exactly as PartitionedIntegrator leaves them.  Returns milliseconds per step."""
    import ctypes as C
    import time
    from . import _lib as L
    shard = np.ascontiguousarray(shard, dtype=np.float64).reshape(-1, 6)
    received = np.ascontiguousarray(received, dtype=np.float64).reshape(-1, 6)
    bufs = []

    def device_copy(arr, extra=0):
        buf, ptr = L._vp(), L._vp()
        L.check(L.lib.ohmhip_buffer_create(C.byref(buf), max(arr.nbytes + extra, 48), 3), "buffer_create")
        if arr.nbytes:
            L.check(L.lib.ohmhip_buffer_write(buf, arr.ctypes.data, arr.nbytes, 0, None, None, None), "buffer_write")
        L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(ptr)), "buffer_ptr")
        bufs.append(buf)
        return ptr

    d_src = device_copy(shard)
    d_recv = device_copy(received)  # read-only here: no buffer rotation needed
    cap = 2 * shard.shape[0] + 1024
    d_out = device_copy(np.zeros((0, 6)), extra=48 * cap)
    try:
        t0 = 0.0
        for i in range(warmup + steps):
            if i == warmup:
                gpu_map.wait()
                t0 = time.perf_counter()
            _, _, fits = gpu_map.routeRays(d_src, shard.shape[0], d_out, cap, ray_update_flags)
            if not fits:
                raise RuntimeError("routing buffer too small")
            if received.shape[0]:
                gpu_map.integrateRaysDevice(d_recv, 2 * received.shape[0], ray_update_flags)
        gpu_map.wait()
        return 1e3 * (time.perf_counter() - t0) / steps
    finally:
        for b in bufs:
            L.lib.ohmhip_buffer_destroy(b)


# ---------------------------------------------------------------------------------------------------------------------
# Territories dealt by measured load (strong scaling of one sensor's stream).
# ---------------------------------------------------------------------------------------------------------------------
def estimate_region_loads(rays, region_size, map_origin=(0, 0, 0), ray_stride=16, samples_per_region=4):
    """Ray-region segments per region, estimated on the host from every `ray_stride`-th ray: points sampled along the ray
    every 1 / samples_per_region of a region edge, distinct (ray, region) pairs counted and scaled back.  A planning aid
    for `territories_by_load` -- the integration itself never uses it.  Returns {(rx, ry, rz): estimated segments}."""
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 6)[::max(1, int(ray_stride))]
    size = np.broadcast_to(np.asarray(region_size, dtype=np.float64), (3,))
    mo = np.asarray(map_origin, dtype=np.float64)
    start, end = rays[:, :3], rays[:, 3:]
    length = np.linalg.norm(end - start, axis=1)
    n_samples = int(np.ceil(length.max() / (size.min() / samples_per_region))) + 1
    t = np.linspace(0.0, 1.0, n_samples)
    loads = {}
    for at in range(0, len(rays), 8192):
        s, e = start[at:at + 8192], end[at:at + 8192]
        p = s[:, None, :] + (e - s)[:, None, :] * t[None, :, None]
        r = np.floor((p - mo) / size + 0.5).astype(np.int64)
        ray_id = np.broadcast_to(np.arange(len(s))[:, None], r.shape[:2])
        packed = (ray_id << 48) | ((r[..., 0] + 32768) << 32) | ((r[..., 1] + 32768) << 16) | (r[..., 2] + 32768)
        pairs = np.unique(packed.reshape(-1)) & ((1 << 48) - 1)
        keys, counts = np.unique(pairs, return_counts=True)
        for k, c in zip(keys.tolist(), counts.tolist()):
            key = ((k >> 32) - 32768, ((k >> 16) & 0xFFFF) - 32768, (k & 0xFFFF) - 32768)
            loads[key] = loads.get(key, 0.0) + float(c) * ray_stride
    return loads


def territories_by_load(loads, world_size, rank, centre, region_size, map_origin=(0, 0, 0), hub_radius=1):
    """A RegionPartition (region granularity) for ONE sensor's stream, dealt by measured load: the regions are ordered by
    the azimuth of their centre about `centre` (the sensor) and cut into `world_size` contiguous arcs of equal load -- a
    ray then crosses one or two territories, not all of them --, except the HUB, the regions within `hub_radius` regions
    of the sensor, which every ray crosses whatever its direction: those go one by one, heaviest first, to the rank with
    the least load so far.  Regions without a measured load fall into the arc their azimuth points at.  `loads`:
    {(rx, ry, rz): load} (estimate_region_loads, or segment counts from a pilot batch)."""
    size = np.broadcast_to(np.asarray(region_size, dtype=np.float64), (3,))
    mo = np.asarray(map_origin, dtype=np.float64)
    keys = np.array(sorted(loads), dtype=np.int64).reshape(-1, 3)
    weight = np.array([loads[tuple(k)] for k in keys.tolist()], dtype=np.float64)
    c_region = np.floor((np.asarray(centre, dtype=np.float64) - mo) / size + 0.5).astype(np.int64)
    lo = np.minimum(keys.min(axis=0), c_region) - 1
    hi = np.maximum(keys.max(axis=0), c_region) + 1
    dims = (hi - lo + 1).astype(np.int64)

    def azimuth(region_keys):
        centres = region_keys * size + mo
        d = centres - np.asarray(centre, dtype=np.float64)
        return np.arctan2(d[..., 1], d[..., 0])

    hub = np.abs(keys - c_region).max(axis=1) <= hub_radius
    rim = ~hub
    # arcs of equal rim load; the hub's load is dealt afterwards to whoever has least
    order = np.argsort(azimuth(keys[rim]), kind="stable")
    rim_weight = weight[rim][order]
    cum = np.cumsum(rim_weight)
    total = float(cum[-1]) if len(cum) else 0.0
    hub_total = float(weight[hub].sum())
    share = (total + hub_total) / world_size
    # boundaries by azimuth such that rank r's arc carries ~ share minus what it will receive from the hub: first deal
    # the hub (LPT) against equal arcs, then re-cut the arcs to even the sums out
    rank_load = np.zeros(world_size)
    hub_owner = {}
    for i in np.argsort(-weight[hub], kind="stable"):
        r = int(np.argmin(rank_load))
        hub_owner[tuple(keys[hub][i].tolist())] = r
        rank_load[r] += weight[hub][i]
    want = np.maximum(share - rank_load, 0.0)
    want *= total / max(want.sum(), 1e-30)
    bounds = np.cumsum(want)
    rim_owner = np.minimum(np.searchsorted(bounds, cum - 0.5 * rim_weight), world_size - 1)
    az_sorted = azimuth(keys[rim])[order]
    # arc limits in azimuth: the last region of each arc
    limits = []
    for r in range(world_size - 1):
        idx = np.flatnonzero(rim_owner <= r)
        limits.append(az_sorted[idx[-1]] if len(idx) else -np.pi)
    limits = np.array(limits)
    ix, iy, iz = np.meshgrid(np.arange(dims[0]), np.arange(dims[1]), np.arange(dims[2]), indexing="ij")
    cells = np.stack([ix, iy, iz], axis=-1) + lo
    table = np.searchsorted(limits, azimuth(cells), side="left").astype(np.uint8)
    table = np.minimum(table, world_size - 1).astype(np.uint8)
    for key, r in hub_owner.items():
        c = np.asarray(key) - lo
        table[c[0], c[1], c[2]] = r
    return RegionPartition(world_size, rank, 0, tuple(int(v) for v in lo), table)
