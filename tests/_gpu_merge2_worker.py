"""Worker for tests/test_gpu_merge.py::test_two_process_gloo_merge_on_device_tiles: one of two ranks (gloo), each with
its own GpuMap on the one test GPU, merging through ohm_amd.distributed.ReplicaMerger (library pack / apply kernels on
device tiles, payload all-reduce over the process group).  torch is imported before ohm_amd on purpose (see
tests/_gpu_merge_worker.py)."""
import os
import sys

import torch  # noqa: F401,E402
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ohm_amd import GpuMap, OccupancyMap, synth  # noqa: E402
from ohm_amd import distributed as D  # noqa: E402
from oracle.oracle import OracleMap  # noqa: E402


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        map_ = OccupancyMap(0.1)
        gm = GpuMap(map_)
        merger = D.ReplicaMerger(gm)  # no RCCL communicator: the steps run over the gloo group
        rays = synth.rays_c0(n=4000, origin=(0.05 + 4.0 * rank, 0.05, 0.05), length=5.0, seed=800 + rank)
        gm.integrateRays(rays)
        st = merger.merge()
        assert st["regions_shared"] > 0 and st["regions_union"] >= st["regions_local"]
        # what only this rank touched stays pending on its shared base; the exchanged regions are settled
        assert len(merger.local_keys()) == st["regions_local"] - st["regions_shared"]
        gm.syncVoxels()
        om = OracleMap(0.1)
        om.integrate_occupancy(rays)
        own = om.chunks()
        # shared = regions both ranks touched: gather the key sets again from the oracle maps
        keys = sorted(own.keys())
        gathered = [None] * world
        dist.all_gather_object(gathered, keys)
        shared = sorted(set(gathered[0]) & set(gathered[1]))
        inf = np.float32(np.inf)
        np.save(os.path.join(out, f"shared_{rank}.npy"), np.array(shared, dtype=np.int16))
        np.save(os.path.join(out, f"tiles_{rank}.npy"), np.stack([map_.chunks[k]["occupancy"] for k in shared]))
        np.save(os.path.join(out, f"own_{rank}.npy"),
                np.stack([own[k]["occupancy"] if k in own else np.full(32 ** 3, inf, np.float32) for k in shared]))
        print("MERGE2_OK", st)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
