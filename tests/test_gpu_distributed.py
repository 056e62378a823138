"""-m gpu: the replica merge on real device memory with RCCL (backend "nccl"), world_size 1 on the single test GPU:
exercises the zero-copy torch view of the resident occupancy layer, ensure_regions / mark_dirty and the all-reduce on
HIP memory.  The multi-rank protocol itself is covered by the gloo tests (tests/test_distributed_cpu.py)."""
import pytest

pytestmark = pytest.mark.gpu


def test_replica_merge_single_rank_rccl(gpu):
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_merge_worker.py")
    res = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "MERGE_OK" in res.stdout, (res.returncode, res.stdout[-2000:], res.stderr[-4000:])
