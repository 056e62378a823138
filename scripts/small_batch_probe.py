"""Development aid: time N device-resident batches of 4096 rays (the reference tools' batch size) -- run under
rocprofv3 --kernel-trace --stats to see which launches the per-batch latency is made of."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ohm_amd
from ohm_amd import _lib as L, synth

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
small = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rays = synth.rays_c1(n=max(small * 64, 1 << 20))
m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
g = ohm_amd.GpuMap(m, gpu_mem_size=4 << 30)
buf = L._vp()
L.check(L.lib.ohmhip_buffer_create(C.byref(buf), rays.nbytes, 3))
L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, rays.nbytes, 0, None, None, None))
ptr = L._vp()
L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(ptr)))
g.integrateRaysDevice(ptr, rays.shape[0])
g.wait()
stride = 2 * small * 24
t0 = time.perf_counter()
for b in range(n_batches):
    g.integrateRaysDevice(C.c_void_p(ptr.value + (b % (rays.shape[0] // (2 * small))) * stride), 2 * small)
g.wait()
dt = time.perf_counter() - t0
print(f"{n_batches} batches of {small} rays: {dt * 1e3 / n_batches:.4f} ms per batch, {n_batches * small / dt:.3e} rays/s")
print(g.batchTimings())
