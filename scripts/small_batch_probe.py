"""Where a small device batch's time goes: 4096-ray calls, coalescing off, one device batch per call (the reference
tools' call size, ohmapp/OhmAppCpu.h:52).  OHMHIP_DEBUG_FLAGS=256 prints the phase timeline of the last batches."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import ohm_amd
from ohm_amd import _lib as L, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rays = synth.rays_c1(n=1_000_000)
buf = L._vp(); L.check(L.lib.ohmhip_buffer_create(C.byref(buf), rays.nbytes, 3)); L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, rays.nbytes, 0, None, None, None))
p = L._vp(); L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(p)))
m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
g = ohm_amd.GpuMap(m, gpu_mem_size=8 << 30)
g.setBatchCoalescing(0)
g.integrateRaysDevice(p, rays.shape[0]); g.wait()   # the map exists: steady state
for k in range(40):
    g.integrateRaysDevice(C.c_void_p(p.value + k * n * 48), 2 * n)
g.wait()
t0 = time.perf_counter()
calls = 200
for k in range(calls):
    g.integrateRaysDevice(C.c_void_p(p.value + (k % 200) * n * 48), 2 * n)
g.wait()
dt = (time.perf_counter() - t0) / calls
bt = g.batchTimings(0)
print("%d rays per call: %.1f us per call (%.3e rays/s); last batch device: total %.1f setup %.1f walk %.1f apply %.1f us" %
      (n, dt * 1e6, n / dt, bt["ms_total"] * 1e3, bt["ms_setup"] * 1e3, bt["ms_walk"] * 1e3, bt["ms_apply"] * 1e3))
g.close()
