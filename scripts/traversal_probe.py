"""Development probe: C1 rays into a map with a traversal layer; prints ms per batch and a checksum of the layer.
OHMHIP_DEBUG_FLAGS=1024 selects the round-2 scheme (global atomic per visit) for comparison."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ohm_amd
from ohm_amd import _lib as L
from ohm_amd import synth

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rays = synth.rays_c1(n=n_rays)
mt = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy", "traversal"))
gt = ohm_amd.GpuMap(mt, gpu_mem_size=8 << 30)
h = L._vp()
L.check(L.lib.ohmhip_buffer_create(C.byref(h), rays.nbytes, 3), "buffer_create")
L.check(L.lib.ohmhip_buffer_write(h, rays.ctypes.data, rays.nbytes, 0, None, None, None), "buffer_write")
dptr = L._vp()
L.check(L.lib.ohmhip_buffer_ptr(h, C.byref(dptr)), "buffer_ptr")
for _ in range(2):
    gt.integrateRaysDevice(dptr, rays.shape[0])
gt.wait()
t1 = time.perf_counter()
for _ in range(5):
    gt.integrateRaysDevice(dptr, rays.shape[0])
gt.wait()
dt = (time.perf_counter() - t1) / 5
gt.syncVoxels()
total = 0.0
peak = 0.0
for key, chunk in mt.chunks.items():
    t = chunk["traversal"].astype(np.float64)
    total += float(t.sum())
    peak = max(peak, float(t.max()))
print("traversal: %.3f ms per batch, walk %.3f ms, layer sum %.6f m, max voxel %.6f m" %
      (dt * 1e3, float(gt.stats()["ms_walk"]), total, peak))
