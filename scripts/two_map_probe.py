"""Development aid: upper bound of what overlapping batches buys.  Two independent maps (each with its own stream) are
fed the same 1 M-ray C1 batch alternately, so one map's front half can run under the other's walk tail / apply kernels;
compared with one map doing all the batches."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import ohm_amd
from ohm_amd import _lib as L, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rays = synth.rays_c1(n=1_000_000)
buf = L._vp()
L.check(L.lib.ohmhip_buffer_create(C.byref(buf), rays.nbytes, 3))
L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, rays.nbytes, 0, None, None, None))
ptr = L._vp()
L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(ptr)))
maps = []
for i in range(2):
    m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    g = ohm_amd.GpuMap(m, gpu_mem_size=8 << 30)
    for _ in range(3):
        g.integrateRaysDevice(ptr, rays.shape[0])
    g.wait()
    maps.append((m, g))
g0 = maps[0][1]
t0 = time.perf_counter()
for _ in range(2 * steps):
    g0.integrateRaysDevice(ptr, rays.shape[0])
g0.wait()
one = (time.perf_counter() - t0) / (2 * steps)
t0 = time.perf_counter()
for _ in range(steps):
    for _, g in maps:
        g.integrateRaysDevice(ptr, rays.shape[0])
for _, g in maps:
    g.wait()
two = (time.perf_counter() - t0) / (2 * steps)
print(f"one map: {one * 1e3:.4f} ms per batch;  two maps alternating: {two * 1e3:.4f} ms per batch  (x{one / two:.3f})")
