"""syncVoxels of C1's map (1243 regions x 128 KiB occupancy): the library call alone against the Python mirror's
bookkeeping around it (5.9 ms = 27.6 GB/s in BENCH r04)."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import ohm_amd
from ohm_amd import _lib as L, synth
from ohm_amd.gpumap import LAYERS

rays = synth.rays_c1(n=1_000_000)
m = ohm_amd.OccupancyMap(0.1, (32, 32, 32), layers=("occupancy",))
g = ohm_amd.GpuMap(m, gpu_mem_size=8 << 30)
g.integrateRays(rays)
g.syncVoxels()
for rep in range(3):
    g.integrateRays(rays)
    g.wait()
    t0 = time.perf_counter()
    keys = g.regionKeys(dirty_only=True)
    t1 = time.perf_counter()
    lid, dtype, comps = LAYERS["occupancy"]
    blocks = [m.chunks[(int(k[0]), int(k[1]), int(k[2]))]["occupancy"] for k in keys]
    ptrs = (C.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
    t2 = time.perf_counter()
    L.check(L.lib.ohmhip_map_read_regions(g._handle, lid, keys.ctypes.data, len(blocks), ptrs))
    t3 = time.perf_counter()
    nbytes = sum(b.nbytes for b in blocks)
    print("regions %d  %.1f MB: dirty keys %.3f ms, python pointer list %.3f ms, read_regions %.3f ms = %.1f GB/s" %
          (len(blocks), nbytes / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, nbytes / (t3 - t2) / 1e9))
    t4 = time.perf_counter()
    g.syncVoxels()
    t5 = time.perf_counter()
    print("   GpuMap.syncVoxels() of the same state: %.3f ms" % ((t5 - t4) * 1e3))
g.close()
