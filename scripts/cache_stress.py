#!/usr/bin/env python
"""The C3 cache-stress leg of bench.py on its own: GpuTsdfMap, 0.05 m, 4 M rays presented as 45-degree
sectors against a pool bounded at the reference's default 1 GiB, least recently used regions spilling to the host store.
OHMHIP_DEBUG_FLAGS=512 prints where the spill path spends its time."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import ohm_amd  # noqa: E402
from ohm_amd import _lib as L  # noqa: E402
from ohm_amd import synth  # noqa: E402

n_rays = 1_000_000
limit = int(float(sys.argv[1]) * (1 << 20)) if len(sys.argv) > 1 else (1 << 30)  # MiB on the command line
m4 = ohm_amd.OccupancyMap(0.05, (32, 32, 32), layers=("tsdf",))
g4 = ohm_amd.GpuTsdfMap(m4, region_capacity=256 if limit < (1 << 29) else 1024)
g4.setMemoryLimit(limit)
g4.setSpillToHost(True)
r4 = synth.rays_c3(n=4 * n_rays)
b4 = L._vp()
L.check(L.lib.ohmhip_buffer_create(C.byref(b4), r4.nbytes, 3), "buffer_create")
L.check(L.lib.ohmhip_buffer_write(b4, r4.ctypes.data, r4.nbytes, 0, None, None, None), "buffer_write")
p4 = L._vp()
L.check(L.lib.ohmhip_buffer_ptr(b4, C.byref(p4)), "buffer_ptr")
sectors = int(sys.argv[2]) if len(sys.argv) > 2 else 32
per = (r4.shape[0] // 2) // sectors
t1 = time.perf_counter()
done4 = 0
for k in range(sectors):
    done4 += g4.integrateRaysDevice(C.c_void_p(p4.value + k * per * 48), 2 * per)
g4.wait()
dt4 = time.perf_counter() - t1
cs = g4.cacheStats()
print("limit %d MiB  rays/s %.3e  seconds %.3f  evictions %d readmissions %d resident %d stored %d  write-backs %d hits %d "
      "stale %d" % (limit >> 20, (done4 // 2) / dt4, dt4, cs["evictions"], cs["readmissions"], cs["regions_resident"],
                   cs["regions_spilled"], cs["writebacks"], cs["writeback_hits"], cs["writeback_stale"]))
g4.close()
