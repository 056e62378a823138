import sys, time, ctypes as C
sys.path.insert(0, '.')
import ohm_amd
from ohm_amd import _lib as L, synth
rays = synth.rays_c1(n=1_000_000)
buf = L._vp(); L.check(L.lib.ohmhip_buffer_create(C.byref(buf), rays.nbytes, 3)); L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, rays.nbytes, 0, None, None, None))
p = L._vp(); L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(p)))
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1  # one variant only (traces)
for rep in range(2):
    if only in (0, 1) and rep != only:
        continue
    m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
    g = ohm_amd.GpuMap(m, expected_element_count=(rays.shape[0] if rep else 2048), gpu_mem_size=8 << 30)
    g.setPhaseTiming(only < 0)
    g.wait()
    for k in range(3):
        t0 = time.perf_counter(); g.integrateRaysDevice(p, rays.shape[0]); g.wait(); dt = time.perf_counter() - t0
        bt = g.batchTimings(0)
        print("rep %d pass %d  host %.3f ms  device total %.3f setup %.3f walk %.3f apply %.3f" % (rep, k, dt * 1e3, bt["ms_total"], bt["ms_setup"], bt["ms_walk"], bt["ms_apply"]))
    g.close()
