"""Development probe: C0 (100 000 uniform 10 m rays from one origin), device-resident, back to back."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import ohm_amd
from ohm_amd import _lib as L
from ohm_amd import synth

rays = synth.rays_c0()
m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
g = ohm_amd.GpuMap(m, gpu_mem_size=8 << 30)
h = L._vp()
L.check(L.lib.ohmhip_buffer_create(C.byref(h), rays.nbytes, 3), "buffer_create")
L.check(L.lib.ohmhip_buffer_write(h, rays.ctypes.data, rays.nbytes, 0, None, None, None), "buffer_write")
p = L._vp()
L.check(L.lib.ohmhip_buffer_ptr(h, C.byref(p)), "buffer_ptr")
for _ in range(5):
    g.integrateRaysDevice(p, rays.shape[0])
g.wait()
best = 1e9
for _ in range(5):
    t = time.perf_counter()
    for _ in range(40):
        g.integrateRaysDevice(p, rays.shape[0])
    g.wait()
    best = min(best, (time.perf_counter() - t) / 40)
print("C0: %.4f ms per batch, %.3e rays/s" % (best * 1e3, rays.shape[0] / 2 / best))
