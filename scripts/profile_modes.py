"""Integrate the C2 (NDT, 1 M rays at 0.2 m) or C3 (TSDF, 4 M rays at 0.05 m in one call) batch of bench.py a few
times: target for `rocprofv3 --kernel-trace --stats` and the separate `--pmc` passes (scripts/profile_modes.sh)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ohm_amd
from ohm_amd import _lib as L
from ohm_amd import synth

mode = sys.argv[1] if len(sys.argv) > 1 else "ndt"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if mode == "ndt":
    cls, res, layers = ohm_amd.GpuNdtMap, 0.2, ("occupancy",)
    rays = synth.rays_c2(n=1_000_000)
else:
    cls, res, layers = ohm_amd.GpuTsdfMap, 0.05, ("tsdf",)
    rays = synth.rays_c3(n=4_000_000)
m = ohm_amd.OccupancyMap(res, (32, 32, 32), layers=layers)
g = cls(m, gpu_mem_size=24 << 30)
buf = L._vp()
L.check(L.lib.ohmhip_buffer_create(C.byref(buf), rays.nbytes, 3), "buffer_create")
L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, rays.nbytes, 0, None, None, None), "buffer_write")
ptr = L._vp()
L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(ptr)), "buffer_ptr")
for _ in range(steps):
    g.integrateRaysDevice(ptr, rays.shape[0])
g.wait()
print(g.stats())
