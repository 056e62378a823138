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
for _ in range(3):
    assert g.integrateRays(rays) == rays.shape[0]
g.wait()
t = time.perf_counter()
n = 10
for _ in range(n):
    g.integrateRays(rays)
g.wait()
dt = (time.perf_counter() - t) / n
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
print("host batches: %.3f ms per call, %.3e rays/s, checksum %.3f" % (dt * 1e3, n_rays / dt, total))
