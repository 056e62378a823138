#!/usr/bin/env python
"""Feature combinations against the CPU oracle on the GPU box (not a pytest: a wider sweep than the suite affords).
  1. large regions (tiles) + NDT + residency limit + spill to host (+ background write-back) on a sensor track
  2. region partition (3 maps in process) + spill on every map + TSDF
  3. large regions + region partition + mean layer + ray flags, ragged shards over several rounds
  4. region partition by table + host-side ray filter (clip box)
Usage: python scripts/stress_combos.py [seed]"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np  # noqa: E402
from ohm_amd import GpuMap, GpuNdtMap, GpuTsdfMap, OccupancyMap, RayFlag, synth  # noqa: E402
from ohm_amd import distributed as D  # noqa: E402
from parity import compare_maps, make_oracle  # noqa: E402
from test_gpu_spill import sensor_rays, track  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
bad_total = 0


def report(name, bad, extra=""):
    global bad_total
    n_bad = sum(v for k, v in bad.items() if k.startswith(("diff_", "missing", "extra")))
    if bad.get("regions_cpu") != bad.get("regions_gpu"):
        n_bad += 1
    bad_total += n_bad
    print("%-58s %s %s" % (name, "ok (%d regions)" % bad.get("regions_cpu", 0) if n_bad == 0 else "MISMATCH %r" % (bad,), extra))


# 1. tiles + NDT + spill (+ write-back)
for writeback in (False, True):
    dims = [(64, 64, 32), (48, 48, 48)][int(rng.integers(2))]
    map_ = OccupancyMap(0.2, dims, layers=("occupancy",))
    gm = GpuNdtMap(map_, region_capacity=64)
    gm.setMemoryLimit(int(rng.integers(60, 120)) * gm.cacheStats()["bytes_per_region"])
    gm.setSpillToHost(True)
    gm.setSpillWriteback(writeback)
    om = make_oracle(map_)
    om.set_ndt(sensor_noise=gm.sensor_noise, sample_threshold=gm.sample_threshold, adaptation_rate=gm.adaptation_rate,
               reinit_threshold=gm.reinitialise_covariance_threshold,
               reinit_count=gm.reinitialise_covariance_point_count, ndt_tm=False)
    for k, origin in enumerate(track(4, spacing=18.0)):
        rays = sensor_rays(origin, 5000, seed=seed * 100 + k, min_range=2.0, max_range=9.0)
        rays[1::2] = np.round(rays[1::2] / 0.6) * 0.6 + 0.07 * rng.normal(size=(rays.shape[0] // 2, 3))
        assert gm.integrateRays(rays) == rays.shape[0]
        om.integrate_ndt(rays)
    st = gm.cacheStats()
    gm.syncVoxels()
    gm.close()
    report("tiles %s + NDT + spill, write-back %s" % (dims, writeback),
           compare_maps(om.chunks(), map_.chunks, list(map_.layers), rel=1e-5),
           "evictions %d readmissions %d write-back hits %d" % (st["evictions"], st["readmissions"], st["writeback_hits"]))

# 2. partition + spill + TSDF: three sensors moving out and back, each batch fits the 60-region pool, the tracks do not
world = 3
starts = [(0.05, 0.05 + 8.0 * r, 0.05) for r in range(world)]
part0 = D.territories_from_origins(starts, world, 0, 3.2, block_shift=0, margin=25.0)
maps = [OccupancyMap(0.1, (32, 32, 32), layers=("tsdf",)) for _ in range(world)]
gms = [GpuTsdfMap(m, default_truncation_distance=0.1, region_capacity=32) for m in maps]
for r, g in enumerate(gms):
    g.setRegionPartition(part0.with_rank(r))
    g.setMemoryLimit(60 * g.cacheStats()["bytes_per_region"])
    g.setSpillToHost(True)
om = make_oracle(maps[0])
opts = gms[0].tsdf_options
om.set_tsdf(max_weight=opts[0], trunc=opts[1], dropoff=opts[2], sparsity=opts[3])
for rnd, stop in enumerate(track(4, spacing=7.0)):
    shards = [sensor_rays((starts[r][0] + stop[0], starts[r][1] + stop[1], 0.0), int(rng.integers(2000, 5000)),
                          seed=seed * 10 + rnd * world + r, min_range=1.5, max_range=5.0) for r in range(world)]
    D.integrate_partitioned_in_process(gms, shards)
    for s in shards:
        om.integrate_tsdf(s)
union = {}
ev = 0
for m, g in zip(maps, gms):
    ev += g.cacheStats()["evictions"]
    g.syncVoxels()
    g.close()
    union.update(m.chunks)
report("partition x3 + spill + TSDF", compare_maps(om.chunks(), union, ["tsdf"], exact_float=True), "evictions %d" % ev)

# 3. tiles + partition + mean + flags, ragged shards
dims = (64, 64, 64)
world = 2
origins = [(0.05, 0.05, 0.05), (11.0, 3.0, 0.05)]
part0 = D.territories_from_origins(origins, world, 0, 6.4, block_shift=0, margin=30.0)
maps = [OccupancyMap(0.1, dims, layers=("occupancy", "mean")) for _ in range(world)]
gms = [GpuMap(m) for m in maps]
for r, g in enumerate(gms):
    g.setRegionPartition(part0.with_rank(r))
om = make_oracle(maps[0])
for rnd, flags in enumerate((0, int(RayFlag.kRfEndPointAsFree), int(RayFlag.kRfExcludeOrigin), int(RayFlag.kRfExcludeSample))):
    shards = [synth.rays_c1(n=int(rng.integers(2000, 12000)), origin=origins[r], max_range=16.0, seed=seed * 7 + rnd * 2 + r,
                            first=int(rng.integers(0, 900000))) for r in range(world)]
    D.integrate_partitioned_in_process(gms, shards, flags)
    for s in shards:
        om.integrate_occupancy(s, flags=flags)
union = {}
for m, g in zip(maps, gms):
    g.syncVoxels()
    g.close()
    union.update(m.chunks)
report("tiles 64^3 + partition x2 + mean + flags", compare_maps(om.chunks(), union, ["occupancy", "mean"], exact_float=True))

# 4. partition + ray filter (max range clip) on every map
world = 4
origins = [(0.05 + 9.0 * (r % 2), 0.05 + 9.0 * (r // 2), 0.05) for r in range(world)]
part0 = D.territories_from_origins(origins, world, 0, 3.2, block_shift=1, margin=30.0)
maps = [OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",)) for _ in range(world)]
for m in maps:
    m.ray_filter = ("clip", 6.0)  # the map's own filter: rays longer than 6 m are clipped, on the device
gms = [GpuMap(m) for m in maps]
for r, g in enumerate(gms):
    g.setRegionPartition(part0.with_rank(r))
om = make_oracle(maps[0])
for rnd in range(2):
    shards = [synth.rays_c1(n=6000, origin=origins[r], max_range=14.0, seed=seed * 3 + rnd * world + r,
                            first=int(rng.integers(0, 900000))) for r in range(world)]
    D.integrate_partitioned_in_process(gms, shards)
    for s in shards:
        om.integrate_occupancy(s)
union = {}
for m, g in zip(maps, gms):
    g.syncVoxels()
    g.close()
    union.update(m.chunks)
report("partition x4 (blocks of 2^3) + range-clipping ray filter", compare_maps(om.chunks(), union, ["occupancy"], exact_float=True))
print("STRESS_COMBOS_OK" if bad_total == 0 else "STRESS_COMBOS_FAILED %d" % bad_total)
