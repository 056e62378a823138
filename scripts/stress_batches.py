import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ohm_amd import GpuMap, OccupancyMap, RayFlag, synth
from parity import compare_maps, make_oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
layers = ("occupancy",) if rng.integers(2) else ("occupancy", "mean")
map_ = OccupancyMap(0.1, (32, 32, 32), layers=layers)
gm = GpuMap(map_, region_capacity=32 if "spill" in sys.argv[2:] else 64)   # tiny pool: grows repeatedly
if "spill" in sys.argv[2:]:
    # bounded pool: the least recently used regions move to the host store and back (one sensor position: every batch
    # touches most regions, so this mostly exercises eviction + immediate re-admission)
    gm.setMemoryLimit(40 * gm.cacheStats()["bytes_per_region"])
    gm.setSpillToHost(True)
if "async" in sys.argv[2:]:
    gm.setAsyncLaunch(True)   # large host batches return once staged; their launch runs on the map's thread
om = make_oracle(map_)
n_total = 1_500_000 if "big" in sys.argv[2:] else 400_000
if "spill" in sys.argv[2:]:
    # sensor positions 9 m apart, 8 m range: one position fits the 40-region budget, the track does not, and the order of
    # the batches walks back and forth over it (eviction, re-admission of regions evicted earlier)
    legs = [0, 1, 2, 1, 0, 2]
    all_rays = np.concatenate([synth.rays_c1(n=n_total // len(legs), max_range=8.0,
                                             origin=(0.05 + 9.0 * k, 0.05 + 0.3 * k, 0.05), seed=11 + i)
                               for i, k in enumerate(legs)])
else:
    all_rays = np.concatenate([synth.rays_c1(n=n_total // 3, max_range=20.0, origin=(0.05 + 1.3 * k, 0.05 - 0.7 * k, 0.05),
                                             seed=11 + k) for k in range(3)])
pos = 0
n_batches = 0
t0 = time.time()
while pos < all_rays.shape[0] // 2 and n_batches < 200:
    n = int(rng.choice([500, 3000, 4096, 20000, 65536, 120000] + ([131072, 140001, 300000] if "big" in sys.argv[2:] else [])))
    part = all_rays[2 * pos:2 * (pos + n)]
    if part.shape[0] == 0:
        break
    pos += n
    flags = int(rng.choice([0, 0, 0, int(RayFlag.kRfEndPointAsFree), int(RayFlag.kRfExcludeOrigin)]))
    got = gm.integrateRays(part, ray_update_flags=flags)
    assert got == part.shape[0], (got, part.shape)
    om.integrate_occupancy(part, flags=flags)
    r = rng.integers(6)
    if r == 0:
        gm.syncVoxels()
    elif r == 1:
        gm.stats()
    elif r == 2:
        gm.wait()
    n_batches += 1
gm.syncVoxels()
stats = compare_maps(om.chunks(), map_.chunks, list(layers), exact_float=True)
bad = {k: v for k, v in stats.items() if (k.startswith("diff_") or k.endswith("_on_gpu") or k.endswith("_on_cpu")) and v}
print(gm.cacheStats()["evictions"], gm.cacheStats()["readmissions"], end=" ")
print("seed", sys.argv[1:], "layers", layers, "batches", n_batches, "rays", pos, "regions", len(map_.chunks), "bad", bad, "%.1fs" % (time.time() - t0))
assert not bad
