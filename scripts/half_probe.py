"""Round 6: C1 rays into maps of different region shapes -- ms per back-to-back batch and the walk kernel's share.
usage: python scripts/half_probe.py 32,32,32 32,32,16 [...]   (OHMHIP_WALK_HALF=0 keeps the full walk shape for small regions)"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import ohm_amd
from ohm_amd import _lib as L
from ohm_amd import synth

rays = synth.rays_c1(n=1_000_000)
buf = L._vp()
L.check(L.lib.ohmhip_buffer_create(C.byref(buf), rays.nbytes, 3), "buffer_create")
L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, rays.nbytes, 0, None, None, None), "buffer_write")
ptr = L._vp()
L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(ptr)), "buffer_ptr")
for spec in sys.argv[1:]:
    dims = tuple(int(v) for v in spec.split(","))
    m = ohm_amd.OccupancyMap(0.1, dims, layers=("occupancy",))
    g = ohm_amd.GpuMap(m, gpu_mem_size=8 << 30)
    for _ in range(6):
        g.integrateRaysDevice(ptr, rays.shape[0])
    g.wait()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        g.integrateRaysDevice(ptr, rays.shape[0])
    g.wait()
    dt = (time.perf_counter() - t0) / steps
    walk = sum(g.batchTimings(b)["ms_walk"] for b in range(steps)) / steps
    st = g.stats()
    print("regions %-10s  %.4f ms per batch  walk %.4f ms  segments %d  regions touched %d" %
          (spec, dt * 1e3, walk, st["ray_region_segments"], st["regions_touched"]))
    g.close()
