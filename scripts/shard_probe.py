"""Time one C4 shard (sensor origin on the voxel lattice) like bench.py times C1: device-resident rays, 10 steps."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import ohm_amd
from ohm_amd import _lib as L, synth
shard = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rays = synth.rays_c4_shard(shard) if shard >= 0 else synth.rays_c1()
m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
g = ohm_amd.GpuMap(m, gpu_mem_size=8 << 30)
hb = L._vp(); L.check(L.lib.ohmhip_buffer_create(C.byref(hb), rays.nbytes, 3)); L.check(L.lib.ohmhip_buffer_write(hb, rays.ctypes.data, rays.nbytes, 0, None, None, None))
pb = L._vp(); L.check(L.lib.ohmhip_buffer_ptr(hb, C.byref(pb)))
for _ in range(3):
    g.integrateRaysDevice(pb, rays.shape[0])
g.wait()
t = time.perf_counter()
for _ in range(10):
    g.integrateRaysDevice(pb, rays.shape[0])
g.wait()
print("shard", shard, "ms/step %.4f" % ((time.perf_counter() - t) / 10 * 1e3), g.batchTimings(0))
