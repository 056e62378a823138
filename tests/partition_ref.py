"""Reference routing for the partitioned-map tests (TEST INFRASTRUCTURE): which ranks a ray has to reach, derived from the
CPU oracle's line walk -- the regions of the voxels the walk visits plus the sample's region -- and the partition's
ownership rule.  The library's routing kernel (ohmhip_map_route_rays) is held to this in tests/test_gpu_partitioned.py."""
import numpy as np


def ray_destinations(om, part, rays, include_end=False):
    """Per ray: sorted list of owner ranks.  `include_end`: the end voxel is walked (kRfEndPointAsFree / TSDF) instead of
    receiving a sample -- the set of regions is the same either way."""
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 3)
    dests = []
    for i in range(rays.shape[0] // 2):
        keys, _, _ = om.walk(rays[2 * i], rays[2 * i + 1], flags=2)  # ORACLE_WALK_EXCLUDE_END
        regions = {k[0] for k in keys}
        end = om.voxel_key(rays[2 * i + 1])
        if end is not None:
            regions.add(end[0])
        owners = part.owners(np.array(sorted(regions), dtype=np.int16)) if regions else []
        dests.append(sorted(set(int(o) for o in owners)))
    return dests


def route_reference(om, part, rays, world, with_index=False):
    """(routed (k, 6) rays: destination blocks back to back, rays in order inside a block; counts per destination[; the
    index of every routed ray in the input -- what ohmhip_map_route_rays returns in d_routed_index and side arrays are
    put into routed order with])."""
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 3)
    dests = ray_destinations(om, part, rays)
    blocks = [[] for _ in range(world)]
    index = [[] for _ in range(world)]
    for i, ds in enumerate(dests):
        for d in ds:
            blocks[d].append(rays[2 * i:2 * i + 2].reshape(6))
            index[d].append(i)
    counts = [len(b) for b in blocks]
    flat = [r for b in blocks for r in b]
    routed = np.array(flat, dtype=np.float64).reshape(-1, 6)
    if with_index:
        return routed, counts, np.array([i for b in index for i in b], dtype=np.int64)
    return routed, counts
