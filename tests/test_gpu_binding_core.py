"""-m gpu: the glm-free core of the Level-2 adaptor (ohm_amd/host/ref_adaptor/private/HipBindingCore.*) -- the logic the
ohm::GpuMap / GpuCache member definitions of that directory delegate to -- driven by ohm_amd/lib/binding_core_driver
(plain g++, built by __graft_entry__.build()): create -> integrate in batches (with and without the host ray-filter
pass) -> stamp-checked download -> destroy -> create -> upload of every host region by stamp -> integrate -> download ->
one CPU-side edit uploads exactly one region.  The driver asserts the stamp protocol itself (GpuLayerCache.cpp:462-502,
670-700); here its final host blocks are compared with the CPU oracle integrating all rays.  What tests the reference's
own harness against the same calls: tests/ohmtestgpu/GpuMapTest.cpp:68-205 (restated in tests/test_gpu_reference_suite.py)."""
import os

import numpy as np
import pytest

import ohm_amd
from ohm_amd import synth
from oracle.oracle import OracleMap

from parity import assert_parity, compare_maps
from test_gpu_cpp_host import run_driver

pytestmark = pytest.mark.gpu

DRIVER = os.path.join(os.path.dirname(ohm_amd.LIB_PATH), "binding_core_driver")


@pytest.mark.parametrize("mode,layers,res", [("occ", ("occupancy",), 0.1), ("occmean", ("occupancy", "mean"), 0.1),
                                             ("ndt", ("occupancy", "mean", "covariance"), 0.2),
                                             ("tsdf", ("tsdf",), 0.1)])
def test_binding_core_upload_integrate_download(gpu, mode, layers, res):
    assert os.path.exists(DRIVER), "binding_core_driver missing: run __graft_entry__.build()"
    rays = synth.rays_c2(n=12000)
    batch = 2048
    gpu_chunks = run_driver(mode, res, batch, rays, len(layers), driver=DRIVER)
    om = OracleMap(res, layers=layers)
    if mode == "ndt":
        om.set_ndt()
    n_points = rays.shape[0]
    half = (n_points // 4) * 2
    for first, end in ((0, half), (half, n_points)):  # the driver's two phases: batches restart at the phase boundary
        for i in range(first, end, 2 * batch):
            chunk = rays[i:min(i + 2 * batch, end)]
            if mode == "ndt":
                om.integrate_ndt(chunk)
            elif mode == "tsdf":
                om.integrate_tsdf(chunk)
            else:
                om.integrate_occupancy(chunk)
    stats = compare_maps(om.chunks(), gpu_chunks, list(layers), rel=1e-5, exact_float=(mode != "ndt"))
    assert_parity(stats)
