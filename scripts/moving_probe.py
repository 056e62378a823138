import sys, time, ctypes as C
sys.path.insert(0,'.')
import ohm_amd
from ohm_amd import _lib as L, synth
mm = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
g = ohm_amd.GpuMap(mm, gpu_mem_size=8 << 30)
bufs=[]
for b in range(9):
    rb = synth.rays_c1(n=1000000, origin=(0.05 + 0.4 * b, 0.05 + 0.1 * b, 0.05), first=b * 997)
    hb = L._vp(); L.check(L.lib.ohmhip_buffer_create(C.byref(hb), rb.nbytes, 3)); L.check(L.lib.ohmhip_buffer_write(hb, rb.ctypes.data, rb.nbytes, 0, None, None, None))
    pb = L._vp(); L.check(L.lib.ohmhip_buffer_ptr(hb, C.byref(pb))); bufs.append((pb, rb.shape[0]))
g.integrateRaysDevice(*bufs[0]); g.wait()
t=time.perf_counter()
for pb,c in bufs[1:]:
    g.integrateRaysDevice(pb,c)
g.wait()
print("ms/step", (time.perf_counter()-t)/8*1e3)
for back in range(8):
    print(back, g.batchTimings(back))
print(g.stats())
