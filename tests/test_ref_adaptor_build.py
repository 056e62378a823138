"""f1: the reference's own GpuMap / GpuNdtMap / GpuTsdfMap / GpuCache member definitions on this backend
(ohm_amd/host/ref_adaptor) are BUILT against the reference checkout whenever glm -- the one thing they need that this
image lacks -- is present; otherwise the build reports `skipped: glm absent`.  No stand-in for glm is ever written
(scripts/build_ref_adaptor.sh).  The gputil half of the adaptor needs no glm and is built and run regardless
(__graft_entry__.build() -> gputil_hip_check, tests/test_gpu_cpp_host.py)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ref_adaptor_builds_when_glm_is_present(tmp_path):
    script = os.path.join(ROOT, "scripts", "build_ref_adaptor.sh")
    res = subprocess.run(["bash", script, "/root/reference", str(tmp_path / "ref_adaptor")], capture_output=True,
                         text=True, timeout=900)
    if res.returncode == 77:
        reason = [ln for ln in res.stdout.splitlines() if ln.startswith("SKIPPED")]
        assert reason, res.stdout
        pytest.skip(reason[0])
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-6000:])
    assert "COMPILED: 9 objects" in res.stdout
    for name in ("GpuMap.o", "GpuNdtMap.o", "GpuTsdfMap.o", "GpuCache.o", "private_HipMapBinding.o"):
        assert os.path.getsize(tmp_path / "ref_adaptor" / "obj" / name) > 0
