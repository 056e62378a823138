"""Worker for tests/test_gpu_distributed.py: runs in its own process because torch must be imported BEFORE ohm_amd when
both are used (torch bundles its own HIP runtime; loading /opt/rocm's first makes torch lose the GPU)."""
import os
import socket
import sys

import torch  # noqa: E402  (first, on purpose)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ohm_amd import GpuMap, OccupancyMap, synth  # noqa: E402
from ohm_amd import distributed as D  # noqa: E402
from oracle.oracle import OracleMap  # noqa: E402


def main():
    assert torch.cuda.is_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        map_ = OccupancyMap(0.1)
        gm = GpuMap(map_)
        comm = D.Communicator()  # RCCL communicator inside libohmhip.so, unique id broadcast over the torch group
        merger = D.ReplicaMerger(gm, comm=comm)
        rays = synth.rays_c0(n=5000, length=4.0)
        gm.integrateRays(rays)
        view = D.occupancy_tensor(gm)
        assert view.is_cuda and view.shape[1] == 32 ** 3
        st = merger.merge()
        n = st["regions_local"]
        assert n == len(gm.regionKeys()) and st["regions_union"] == n and st["regions_shared"] == 0
        # with one rank nothing is shared: every region stays pending on its (unobserved) shared base ...
        assert len(merger.local_keys()) == n
        gm.integrateRays(rays[:2000])
        st = merger.merge()
        assert st["regions_local"] == n and st["payload_bytes"] == 0
        # ... until a full-union merge exchanges it (with itself here: merged = clamp(base + own delta) = own value)
        from ohm_amd import _lib as LL
        LL.check(LL.lib.ohmhip_map_set_merge_mode(gm._handle, LL.MERGE_FULL_UNION), "merge_mode")
        st = merger.merge()
        assert st["regions_shared"] == n and st["payload_bytes"] == 5 * n * 32 ** 3
        assert len(merger.local_keys()) == 0
        comm.close()
        gm.syncVoxels()
        om = OracleMap(0.1)
        om.integrate_occupancy(rays)
        om.integrate_occupancy(rays[:2000])
        checked = 0
        for key, layers in om.chunks().items():
            got = map_.chunks[key]["occupancy"]
            exp = layers["occupancy"]
            assert np.array_equal(np.isinf(got), np.isinf(exp))
            fin = np.isfinite(exp)
            assert np.allclose(got[fin], exp[fin], rtol=1e-6, atol=1e-6)
            checked += int(fin.sum())
        # Owner-computes integrator on device tensors: RCCL all-gather of the ray batch, then the map reads the gathered
        # stream straight from the torch allocation (world 1 here: the filter is off, the plumbing is what runs).
        map2 = OccupancyMap(0.1)
        gm2 = GpuMap(map2)
        integ = D.OwnerComputesIntegrator(gm2)
        d_rays = torch.from_numpy(rays).to("cuda")
        assert integ.integrateRays(d_rays) == rays.shape[0]
        assert integ.integrateRays(d_rays[:4000]) == 4000
        gm2.syncVoxels()
        om2 = OracleMap(0.1)
        om2.integrate_occupancy(rays)
        om2.integrate_occupancy(rays[:4000])
        for key, layers in om2.chunks().items():
            assert np.array_equal(map2.chunks[key]["occupancy"].view(np.uint32), layers["occupancy"].view(np.uint32))
        # Partitioned integrator, both transports, world 1: the routing kernels, the library's RCCL exchange (count
        # all-gather + the self block) and torch's all_to_all_single on device tensors all run; every ray comes back.
        for use_comm in (True, False):
            map3 = OccupancyMap(0.1)
            gm3 = GpuMap(map3)
            comm3 = D.Communicator() if use_comm else None
            part = D.territories_from_origins([(0.05, 0.05, 0.05)], 1, 0, 3.2)
            pinteg = D.PartitionedIntegrator(gm3, part, comm=comm3)
            assert pinteg.integrateRays(d_rays) == rays.shape[0]
            assert pinteg.last["rays_received"] == rays.shape[0] // 2 == pinteg.last["rays_kept"]
            assert pinteg.integrateRays(d_rays[:4000]) == 4000
            gm3.syncVoxels()
            for key, layers in om2.chunks().items():
                assert np.array_equal(map3.chunks[key]["occupancy"].view(np.uint32), layers["occupancy"].view(np.uint32))
            # Batches left in flight (no wait between steps; every step its own device batch with rays of its own): the
            # three receive buffers used in turn are never written under a batch that still reads them.
            map4, map5 = OccupancyMap(0.1), OccupancyMap(0.1)
            gm4, gm5 = GpuMap(map4), GpuMap(map5)
            pin4 = D.PartitionedIntegrator(gm4, part, comm=comm3)
            steps = [torch.from_numpy(synth.rays_c1(n=90000 + 7000 * k, seed=50 + k, first=40000 * k)).cuda()
                     for k in range(7)]
            # (the receive ring advances per LAUNCHED batch.  Steps 2 and 5 are small -- their rays are only
            # collected, no batch is launched -- and must leave the ring where it is: the buffers of the two batches
            # still in flight stay untouched, and the small step's own buffer is free when its call returns.)
            small = {2: torch.from_numpy(synth.rays_c1(n=3000, seed=70)).cuda(),
                     5: torch.from_numpy(synth.rays_c1(n=2500, seed=71)).cuda()}
            order = []
            for k, t in enumerate(steps):
                if k in small:
                    launched = gm4.batchesLaunched()
                    in_flight = sorted(v for v in pin4._recv_batch if v and v + 2 > launched)
                    assert len(in_flight) == 2  # the two batches launched last
                    assert pin4.integrateRays(small[k]) == small[k].shape[0]
                    assert gm4.batchesLaunched() == launched
                    assert sorted(v for v in pin4._recv_batch if v and v + 2 > launched) == in_flight
                    order.append(small[k])
                launched = gm4.batchesLaunched()
                assert pin4.integrateRays(t) == t.shape[0]
                assert gm4.batchesLaunched() > launched
                order.append(t)
            for t in order:
                gm5.integrateRays(t.cpu().numpy())
            gm4.syncVoxels()
            gm5.syncVoxels()
            assert set(map4.chunks) == set(map5.chunks)
            for key, layers in map5.chunks.items():
                assert np.array_equal(map4.chunks[key]["occupancy"].view(np.uint32), layers["occupancy"].view(np.uint32))
            pin4.close()
            pinteg.close()
            gm4.close()
            gm5.close()
            if comm3 is not None:
                comm3.close()
            gm3.close()
        print("MERGE_OK", n, checked)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
