import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ohm_amd
from ohm_amd import GpuMap, OccupancyMap, synth
from parity import make_oracle

rays = np.array([[0.3, 0, 0], [1.1, 0, 0], [-5, 0, 0], [0.11, 0, 0]], dtype=np.float64)
map_ = OccupancyMap(0.1, (32, 32, 32))
gm = GpuMap(map_)
print("integrated", gm.integrateRays(rays))
gm.syncVoxels()
print(gm.stats())
om = make_oracle(map_)
om.integrate_occupancy(rays)
cpu = om.chunks()
for key in sorted(cpu):
    c = cpu[key]["occupancy"]; g = map_.chunks.get(key, {}).get("occupancy")
    ci = np.nonzero(np.isfinite(c))[0]
    gi = np.nonzero(np.isfinite(g))[0] if g is not None else []
    print(key, "cpu finite", len(ci), "gpu finite", len(gi))
    print("  cpu idx", ci[:20], c[ci[:20]])
    print("  gpu idx", gi[:20], g[gi[:20]] if g is not None else None)
