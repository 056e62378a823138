"""Development probe: host-pointer C1 batches (48 B/ray over PCIe), back to back."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ohm_amd
from ohm_amd import synth

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rays = synth.rays_c1(n=n_rays)
m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
g = ohm_amd.GpuMap(m, gpu_mem_size=8 << 30)
if len(sys.argv) > 2 and sys.argv[2] == "async":
    g.setAsyncLaunch(True)
for _ in range(3):
    assert g.integrateRays(rays) == rays.shape[0]
g.wait()
groups = []
n = 20
for _ in range(6):
    t = time.perf_counter()
    for _ in range(n):
        g.integrateRays(rays)
    g.wait()
    groups.append((time.perf_counter() - t) / n)
dt = min(groups)
dt_med = sorted(groups)[len(groups) // 2]
t = time.perf_counter()
g.syncVoxels()
dts = time.perf_counter() - t
n_bytes = sum(c["occupancy"].nbytes for c in m.chunks.values())
print("first syncVoxels (creates the host chunks): %.3f ms for %.1f MB" % (dts * 1e3, n_bytes / 1e6))
for _ in range(3):
    g.integrateRays(rays)
    g.wait()
    t = time.perf_counter()
    g.syncVoxels()
    dts = time.perf_counter() - t
    print("syncVoxels: %.3f ms for %.1f MB (%.1f GB/s)" % (dts * 1e3, n_bytes / 1e6, n_bytes / dts / 1e9))
total = sum(float(c["occupancy"][np.isfinite(c["occupancy"])].sum()) for c in m.chunks.values())
print("host batches: best %.3f ms per call (%.3e rays/s), median %.3f ms of 6 x 20 calls, checksum %.3f" %
      (dt * 1e3, n_rays / dt, dt_med * 1e3, total))
