#!/bin/bash
# C2 / C3 batch time of the loaded library (scripts/profile_modes.py prints stats; here: wall time per batch over 6 batches)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import ctypes as C, sys, time
sys.path.insert(0, '.')
import ohm_amd
from ohm_amd import _lib as L, synth
for mode, cls, res, layers, rays in (("C2 ndt", ohm_amd.GpuNdtMap, 0.2, ("occupancy",), synth.rays_c2(n=1_000_000)),
                                     ("C3 tsdf", ohm_amd.GpuTsdfMap, 0.05, ("tsdf",), synth.rays_c3(n=4_000_000))):
    m = ohm_amd.OccupancyMap(res, (32, 32, 32), layers=layers)
    g = cls(m, gpu_mem_size=24 << 30)
    buf = L._vp(); L.check(L.lib.ohmhip_buffer_create(C.byref(buf), rays.nbytes, 3)); L.check(L.lib.ohmhip_buffer_write(buf, rays.ctypes.data, rays.nbytes, 0, None, None, None))
    p = L._vp(); L.check(L.lib.ohmhip_buffer_ptr(buf, C.byref(p)))
    for _ in range(3):
        g.integrateRaysDevice(p, rays.shape[0])
    g.wait()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(6):
            g.integrateRaysDevice(p, rays.shape[0])
        g.wait()
        best = min(best, (time.perf_counter() - t0) / 6)
    print("%s: %.4f ms per batch" % (mode, best * 1e3))
    g.close(); L.lib.ohmhip_buffer_destroy(buf)
PY
