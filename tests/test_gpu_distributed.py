"""-m gpu: the multi-GPU paths on real device memory with RCCL (backend "nccl").  World_size 1 on the single test GPU:
the replica merge (zero-copy torch view of the resident occupancy layer, ensure_regions / mark_dirty, the all-reduce on
HIP memory) and the partitioned integrator over both of its transports (the library's RCCL exchange, torch's
all_to_all_single).  World_size 2 over RCCL runs when two devices are visible and is reported as SKIPPED otherwise --
the pool's test boxes have one GPU; the multi-rank protocol itself is covered by the gloo tests
(tests/test_distributed_cpu.py, tests/test_partition_cpu.py) and by two gloo ranks sharing the GPU
(tests/test_gpu_full_configs.py)."""
import pytest

pytestmark = pytest.mark.gpu


def test_replica_merge_single_rank_rccl(gpu):
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_merge_worker.py")
    res = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "MERGE_OK" in res.stdout, (res.returncode, res.stdout[-2000:], res.stderr[-4000:])


def test_two_rank_rccl_partitioned_map(gpu):
    """`bench.py --gpus 2` with one rank per GPU: RCCL communicator inside the library, routed rays exchanged by
    ncclSend / ncclRecv, result bit-identical to sequential integration.  Needs two devices."""
    import json
    import os
    import subprocess
    import sys
    import ohm_amd
    if ohm_amd.device_count() < 2:
        pytest.skip("RCCL with more than one rank needs two GPUs; this box has %d" % ohm_amd.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for mode in ("partitioned", "replica-merge"):
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
                              "1", "--rays", "200000", "--multi-gpu-mode", mode], env=env, capture_output=True,
                             text=True, timeout=900)
        assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
        line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
        assert line["n_gpus"] == 2 and line["backend"] == "RCCL" and line["rccl_ranks"] == 2
        if mode == "partitioned":
            dev = line["multi_gpu"]["deviation"]
            assert "error" not in dev and dev["regions_compared"] == dev["regions_sequential"] > 0
            assert dev["voxels_value_differs"] == 0 and dev["regions_missing"] == 0
        else:
            assert "error" not in line["merge"] and line["merge"]["deviation"]["voxels_state_differs"] == 0
