"""-m gpu: regions larger than 32768 voxels (ohm/OccupancyMap.h:287 takes any glm::u8vec3 region size; VERDICT r3 missing
3).  Inside the library such a region is cut into tiles of at most 2^15 voxels (ohm_amd/csrc/tiling_impl.h: z slabs of
whole layers, or y strips of single layers when one layer alone is too large); the C ABI keeps speaking the caller's
region keys and MapChunk blocks.  Bar as everywhere: region sets and integer fields bit exact against the CPU oracle
run with the SAME region dimensions, occupancy / mean / TSDF bit exact, NDT within 1e-5."""
import numpy as np
import pytest

from ohm_amd import GpuMap, GpuNdtMap, GpuTsdfMap, OccupancyMap, RayFlag, synth

from parity import assert_parity, compare_maps, make_oracle

pytestmark = pytest.mark.gpu

DIMS = [(64, 64, 32), (48, 48, 48), (64, 64, 64), (255, 255, 2), (40, 36, 37), (16, 128, 128)]


def _rays(res, seed):
    scale = res / 0.1
    a = synth.rays_c1(n=9000, max_range=14.0 * scale, seed=seed)
    b = synth.random_rays(3000, extent=9.0 * scale, seed=seed + 1, origin_spread=4.0 * scale)
    b[1::2, 2] *= 0.6
    return np.concatenate([a, b])


@pytest.mark.parametrize("dims", DIMS)
def test_occupancy_and_mean_with_large_regions(gpu, dims):
    layers = ("occupancy", "mean")
    res = 0.1
    map_ = OccupancyMap(res, dims, layers=layers)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    rays = _rays(res, 300 + dims[0])
    total = 0
    for i, flags in ((0, 0), (1, int(RayFlag.kRfEndPointAsFree)), (2, 0)):
        part = rays[i * 8000:(i + 1) * 8000]
        total += gm.integrateRays(part, ray_update_flags=flags)
        om.integrate_occupancy(part, flags=flags)
    assert total == rays.shape[0]
    gm.syncVoxels()
    expect = om.chunks()
    assert_parity(compare_maps(expect, map_.chunks, list(layers), exact_float=True))
    assert set(map_.chunks) == set(expect) and len(expect) > 1
    # the listing speaks region keys: as many regions as the oracle holds, each once
    keys = gm.regionKeys()
    assert len(keys) == len(expect) == len({tuple(int(v) for v in k) for k in keys})
    gm.close()


def test_region_blocks_round_trip_and_removal(gpu):
    """write_regions / read_regions move whole MapChunk blocks of the CALLER's region size (upload of a CPU-side map,
    sync back), tiles no ray has reached read as cleared, remove_regions drops a region with all its tiles."""
    dims = (64, 64, 64)
    layers = ("occupancy", "mean")
    rays = _rays(0.1, 77)
    first, second = rays[:12000], rays[12000:]
    # CPU-side map integrated on the CPU, then handed to the GPU (gpumap::enableGpu + upload) for the second half
    cpu = OccupancyMap(0.1, dims, layers=layers)
    om = make_oracle(cpu)
    om.integrate_occupancy(first)
    cpu.chunks = {k: {n: v.copy() for n, v in c.items()} for k, c in om.chunks().items()}
    gm = GpuMap(cpu)                      # uploads every chunk the host map holds
    assert gm.integrateRays(second) == second.shape[0]
    om.integrate_occupancy(second)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), cpu.chunks, list(layers), exact_float=True))
    # a sparse region: one short ray touches one tile of eight -> the other seven read as cleared
    lone = np.array([[40.05, 40.05, 40.05], [40.35, 40.05, 40.05]])
    gm.integrateRays(lone)
    om.integrate_occupancy(lone)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), cpu.chunks, list(layers), exact_float=True))
    far_key = tuple(int(v) for v in np.floor(lone[0] / 6.4 + 0.5))
    block = cpu.chunks[far_key]["occupancy"]
    assert block.size == 64 ** 3 and np.isinf(block).sum() > block.size - 16
    # removal by region key
    before = len(gm.regionKeys())
    assert gm.removeRegions([far_key]) == 1
    assert len(gm.regionKeys()) == before - 1 and far_key not in {tuple(int(v) for v in k) for k in gm.regionKeys()}
    gm.close()


def test_tsdf_with_large_regions(gpu):
    dims = (64, 64, 32)
    map_ = OccupancyMap(0.1, dims, layers=("tsdf",))
    gm = GpuTsdfMap(map_, default_truncation_distance=0.1)
    om = make_oracle(map_)
    opts = gm.tsdf_options
    om.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
    rays = synth.rays_c2(n=14000, seed=5)
    for i in range(0, rays.shape[0], 12000):
        gm.integrateRays(rays[i:i + 12000])
        om.integrate_tsdf(rays[i:i + 12000])
    gm.syncVoxels()
    gm.close()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["tsdf"], exact_float=True))


def test_ndt_with_large_regions(gpu):
    dims = (48, 48, 48)
    map_ = OccupancyMap(0.2, dims, layers=("occupancy",))
    gm = GpuNdtMap(map_)
    om = make_oracle(map_)
    om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold, adaptation_rate=gm.adaptation_rate,
               reinit_threshold=gm.reinitialise_covariance_threshold,
               reinit_count=gm.reinitialise_covariance_point_count, ndt_tm=False)
    rays = synth.rays_c2(n=30000, seed=9)
    for i in range(0, rays.shape[0], 24000):
        gm.integrateRays(rays[i:i + 24000])
        om.integrate_ndt(rays[i:i + 24000])
    gm.syncVoxels()
    gm.close()
    assert_parity(compare_maps(om.chunks(), map_.chunks, list(map_.layers), rel=1e-5))


def test_line_keys_speak_region_keys(gpu):
    """The LineKeysQueryGpu counterpart reports the caller's region key + voxel inside the region, not tiles."""
    dims = (64, 64, 64)
    map_ = OccupancyMap(0.1, dims)
    gm = GpuMap(map_)
    om = make_oracle(map_)
    lines = synth.random_rays(300, extent=15.0, seed=31, origin_spread=3.0)
    regions, voxels, counts = gm.lineKeys(lines, max_keys_per_line=1024)
    for i in range(lines.shape[0] // 2):
        keys, _, _ = om.walk(lines[2 * i], lines[2 * i + 1], flags=0)
        assert counts[i] == len(keys)
        for j, (region, local) in enumerate(keys):
            assert tuple(regions[i, j]) == tuple(region) and tuple(voxels[i, j]) == tuple(local)
    gm.close()


def test_partitioned_map_with_large_regions(gpu):
    """Ownership is a property of the caller's regions: all tiles of a region go to one rank."""
    from ohm_amd import distributed as D
    dims = (64, 64, 64)
    origins = [(0.05, 0.05, 0.05), (12.85, 0.05, 0.05)]
    part0 = D.territories_from_origins(origins, 2, 0, 6.4, block_shift=0, margin=20.0)
    maps = [OccupancyMap(0.1, dims, layers=("occupancy", "mean")) for _ in range(2)]
    gms = [GpuMap(m) for m in maps]
    for r, g in enumerate(gms):
        g.setRegionPartition(part0.with_rank(r))
    om = make_oracle(maps[0])
    # (rank 0's sweep points along +x, rank 1's -- half a revolution later -- along -x: each reaches into the other's)
    shards = [synth.rays_c1(n=8000, origin=origins[r], max_range=14.0, seed=40 + r, first=496000 * r) for r in range(2)]
    info = D.integrate_partitioned_in_process(gms, shards)
    assert info["routed"][0, 1] > 0 and info["routed"][1, 0] > 0
    for s in shards:
        om.integrate_occupancy(s)
    union = {}
    for r, (m, g) in enumerate(zip(maps, gms)):
        g.syncVoxels()
        g.close()
        keys = np.array(sorted(m.chunks), dtype=np.int16).reshape(-1, 3)
        assert np.all(part0.owners(keys) == r)
        union.update(m.chunks)
    assert_parity(compare_maps(om.chunks(), union, ["occupancy", "mean"], exact_float=True))


def test_spill_to_host_with_large_regions(gpu):
    """The residency limit counts tiles; evicted tiles stay part of their region (listed, synced from the host store) and
    come back when rays reach them again.  Same track and sensor as tests/test_gpu_spill.py, 64 x 64 x 32 regions."""
    from test_gpu_spill import sensor_rays, track
    dims = (64, 64, 32)
    map_ = OccupancyMap(0.1, dims, layers=("occupancy", "mean"))
    gm = GpuMap(map_, region_capacity=64)
    gm.setMemoryLimit(100 * gm.cacheStats()["bytes_per_region"])  # 100 tiles of 64 x 64 x 8 voxels
    gm.setSpillToHost(True)
    om = make_oracle(map_)
    for k, origin in enumerate(track(6)):
        rays = sensor_rays(origin, 6000, seed=700 + k)
        assert gm.integrateRays(rays) == rays.shape[0]
        om.integrate_occupancy(rays)
    st = gm.cacheStats()
    assert st["evictions"] > 0 and st["readmissions"] > 0 and st["regions_spilled"] > 0
    assert st["regions_resident"] <= 100
    assert len(gm.regionKeys()) == len(om.chunks())
    gm.syncVoxels()
    gm.close()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean"], exact_float=True))


def test_tile_key_range_boundary_is_pinned_and_counted(gpu):
    """include/ohmhip.h "LARGE REGIONS": tile coordinates share the packed key's 16-bit fields, so a 64^3 region (8 z
    slabs) is addressable for region z in [-4096, 4095] only, where the reference addresses +-32767
    (ohm/MapRegion.cpp:32-69).  Pinned here: the last addressable region behaves like any other (parity with the oracle,
    nothing counted), a ray one region further is neither walked nor sampled, and ohmhip_map_rays_beyond_tiles counts it
    (such rays must not vanish silently)."""
    dims = (64, 64, 64)
    edge = 6.4
    map_ = OccupancyMap(0.1, dims, layers=("occupancy", "mean"))
    gm = GpuMap(map_)
    om = make_oracle(map_)
    z_in = 4095 * edge          # centre of the last addressable region along z
    z_out = 4096 * edge         # one region further: region 4096 x 8 tiles = 32768 > int16
    inside = np.array([[0.05, 0.05, z_in + 0.05], [0.35, 0.05, z_in + 0.25]])
    assert gm.integrateRays(inside) == 2
    om.integrate_occupancy(inside)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean"], exact_float=True))
    assert (0, 0, 4095) in map_.chunks and gm.raysBeyondTiles() == 0
    # the same at the negative end: region -4096 is the last one there
    low = np.array([[0.05, 0.05, -4096 * edge + 0.05], [0.35, 0.05, -4096 * edge + 0.25]])
    assert gm.integrateRays(low) == 2
    om.integrate_occupancy(low)
    gm.syncVoxels()
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean"], exact_float=True))
    assert (0, 0, -4096) in map_.chunks and gm.raysBeyondTiles() == 0
    regions_before = len(gm.regionKeys())
    # entirely beyond: passes the filter (counted as integrated, like the reference's upload count), changes nothing
    beyond = np.array([[0.05, 0.05, z_out + 0.05], [0.35, 0.05, z_out + 0.25]])
    assert gm.integrateRays(beyond) == 2
    assert gm.raysBeyondTiles() == 1
    # crossing the limit: start in region 4095, sample in region 4096 -> not walked, sample dropped
    crossing = np.array([[0.05, 0.05, z_in + 3.15], [0.05, 0.05, z_in + 3.35]])
    assert gm.integrateRays(crossing) == 2
    assert gm.raysBeyondTiles() == 2
    below = np.array([[0.05, 0.05, -4097 * edge + 0.05], [0.35, 0.05, -4097 * edge + 0.25]])
    assert gm.integrateRays(below) == 2
    assert gm.raysBeyondTiles() == 3
    gm.syncVoxels()
    assert len(gm.regionKeys()) == regions_before
    assert_parity(compare_maps(om.chunks(), map_.chunks, ["occupancy", "mean"], exact_float=True))
    gm.close()
    # ordinary regions: the reference's own range applies, nothing is ever counted
    m32 = OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    g32 = GpuMap(m32)
    far = np.array([[0.05, 0.05, 32767 * 3.2 + 0.05], [0.35, 0.05, 32767 * 3.2 + 0.25]])
    assert g32.integrateRays(far) == 2
    g32.syncVoxels()
    assert (0, 0, 32767) in m32.chunks and g32.raysBeyondTiles() == 0
    g32.close()
