#!/usr/bin/env python
"""Spill to host under plain LRU pressure: a sensor moving in a straight line through new space (GpuTsdfMap, 0.05 m,
1 GiB pool), one 125 k-ray revolution per stop.  The regions left behind are never touched again: the case the
background write-back is made for (OHMHIP_WRITEBACK=0 / 1 to compare)."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import ohm_amd  # noqa: E402
from ohm_amd import _lib as L  # noqa: E402
from ohm_amd import synth  # noqa: E402

stops = int(sys.argv[1]) if len(sys.argv) > 1 else 48
m = ohm_amd.OccupancyMap(0.05, (32, 32, 32), layers=("tsdf",))
g = ohm_amd.GpuTsdfMap(m, region_capacity=1024)
g.setMemoryLimit(1 << 30)
g.setSpillToHost(True)
if len(sys.argv) > 2:
    L.check(L.lib.ohmhip_map_set_spill_writeback(g._handle, int(sys.argv[2])))
bufs = []
for k in range(stops):
    idx = np.arange(125_000) * 8  # every eighth ray of a revolution: a whole sweep per stop
    d, _ = synth.lidar_directions(1_000_000)
    d = d[idx]
    r = synth._room_range(d, half=8.0, max_range=12.0)
    o = np.array([0.05 + 1.5 * k, 0.05, 0.05])
    rays = synth._pairs(o, o + d * r[:, None])
    b, p = L._vp(), L._vp()
    L.check(L.lib.ohmhip_buffer_create(C.byref(b), rays.nbytes, 3))
    L.check(L.lib.ohmhip_buffer_write(b, rays.ctypes.data, rays.nbytes, 0, None, None, None))
    L.check(L.lib.ohmhip_buffer_ptr(b, C.byref(p)))
    bufs.append((b, p, rays.shape[0]))
g.wait()
t0 = time.perf_counter()
for b, p, n in bufs:
    g.integrateRaysDevice(p, n)
g.wait()
dt = time.perf_counter() - t0
cs = g.cacheStats()
print("stops %d  rays/s %.3e  seconds %.3f  evictions %d readmissions %d resident %d stored %d  write-backs %d hits %d stale %d" %
      (stops, stops * 125_000 / dt, dt, cs["evictions"], cs["readmissions"], cs["regions_resident"], cs["regions_spilled"],
       cs["writebacks"], cs["writeback_hits"], cs["writeback_stale"]))
g.close()
