"""-m gpu: the C++14 host mirror (ohm_amd/host/OhmGpuMap.h: ohm::GpuMap / GpuNdtMap / GpuTsdfMap over the C ABI),
driven by ohm_amd/lib/gpumap_driver (built by __graft_entry__.build() with plain g++), checked against the oracle.
Mirrors the batching harness of tests/ohmtestgpu/GpuMapTest.cpp:68-205."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import ohm_amd
from ohm_amd import LAYERS, synth
from oracle.oracle import OracleMap

from parity import assert_parity, compare_maps

pytestmark = pytest.mark.gpu

DRIVER = os.path.join(os.path.dirname(ohm_amd.LIB_PATH), "gpumap_driver")
ID_TO_NAME = {v[0]: k for k, v in LAYERS.items()}


def run_driver(mode, resolution, batch, rays, n_layers, driver=None):
    driver = driver or DRIVER
    assert os.path.exists(driver), "%s missing: run __graft_entry__.build()" % os.path.basename(driver)
    with tempfile.TemporaryDirectory() as tmp:
        rp, op = os.path.join(tmp, "rays.bin"), os.path.join(tmp, "out.bin")
        with open(rp, "wb") as f:
            f.write(struct.pack("<Q", rays.shape[0]))
            f.write(np.ascontiguousarray(rays, dtype=np.float64).tobytes())
        res = subprocess.run([driver, mode, repr(resolution), str(batch), rp, op], capture_output=True, text=True,
                             timeout=300)
        assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
        data = open(op, "rb").read()
    off = 0
    (n_regions,) = struct.unpack_from("<Q", data, off)
    off += 8
    chunks = {}
    for _ in range(n_regions):
        key = struct.unpack_from("<3h", data, off)
        off += 6
        layers = {}
        for _l in range(n_layers):
            lid, nbytes = struct.unpack_from("<IQ", data, off)
            off += 12
            name = ID_TO_NAME[lid]
            dtype = np.dtype(LAYERS[name][1])
            layers[name] = np.frombuffer(data, dtype=dtype, count=nbytes // dtype.itemsize, offset=off).copy()
            off += nbytes
        chunks[tuple(key)] = layers
    assert off == len(data)
    return chunks


@pytest.mark.parametrize("mode,layers,res", [("occ", ("occupancy",), 0.1), ("occmean", ("occupancy", "mean"), 0.1),
                                             ("ndt", ("occupancy", "mean", "covariance"), 0.2),
                                             ("tsdf", ("tsdf",), 0.1)])
def test_cpp_host_mirror_matches_oracle(gpu, mode, layers, res):
    rays = synth.rays_c2(n=12000)
    # the C++ OccupancyMap keeps its default occupancy layer next to the TSDF layer (as MapFlag::kDefault does)
    gpu_chunks = run_driver(mode, res, 4096, rays, 2 if mode == "tsdf" else len(layers))
    om = OracleMap(res, layers=layers)
    for i in range(0, rays.shape[0], 2 * 4096):
        chunk = rays[i:i + 2 * 4096]
        if mode == "ndt":
            om.set_ndt() if i == 0 else None
            om.integrate_ndt(chunk)
        elif mode == "tsdf":
            om.integrate_tsdf(chunk)
        else:
            om.integrate_occupancy(chunk)
    stats = compare_maps(om.chunks(), gpu_chunks, list(layers), rel=1e-5, exact_float=(mode != "ndt"))
    assert_parity(stats)


def test_cpp_line_keys_query_gpu(gpu):
    """ohm::LineKeysQueryGpu of the C++ mirror (setRays / executeAsync / wait / resultIndices / resultCounts /
    intersectedVoxels, ohm/LineKeysQuery.h:47-101): every ray's keys equal the oracle's CPU walk."""
    lines = synth.random_rays(1500, extent=4.0, seed=77, origin_spread=2.0)
    assert os.path.exists(DRIVER), "gpumap_driver missing: run __graft_entry__.build()"
    with tempfile.TemporaryDirectory() as tmp:
        rp, op = os.path.join(tmp, "rays.bin"), os.path.join(tmp, "out.bin")
        with open(rp, "wb") as f:
            f.write(struct.pack("<Q", lines.shape[0]))
            f.write(np.ascontiguousarray(lines, dtype=np.float64).tobytes())
        res = subprocess.run([DRIVER, "linekeys", "0.1", "0", rp, op], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
        data = open(op, "rb").read()
    (n,) = struct.unpack_from("<Q", data, 0)
    assert n == lines.shape[0] // 2
    table = np.frombuffer(data, dtype="<u8", count=2 * n, offset=8).reshape(n, 2)
    keys_off = 8 + 16 * n
    total = int(table[-1, 0] + table[-1, 1])
    assert len(data) == keys_off + 9 * total
    om = OracleMap(0.1)
    pairs = lines.reshape(-1, 6)
    for i in range(n):
        keys, _, _ = om.walk(pairs[i, :3], pairs[i, 3:], 0)
        index, count = int(table[i, 0]), int(table[i, 1])
        assert count == len(keys)
        for j, (region, local) in enumerate(keys):
            off = keys_off + 9 * (index + j)
            assert struct.unpack_from("<3h", data, off) == tuple(region)
            assert struct.unpack_from("<3B", data, off + 6) == tuple(local)


def test_cpp_transform_samples_feeds_device_integration(gpu):
    # ohm::GpuTransformSamples -> device buffer -> GpuMap::integrateRays(Buffer): with a static identity trajectory the
    # result must equal integrating rays from the origin to the same sample points.
    rays = synth.rays_c2(n=8000)
    gpu_chunks = run_driver("occdev", 0.1, 4096, rays, 1)
    from_origin = rays.copy()
    from_origin[0::2] = 0.0
    om = OracleMap(0.1, layers=("occupancy",))
    for i in range(0, rays.shape[0], 2 * 4096):
        om.integrate_occupancy(from_origin[i:i + 2 * 4096])
    stats = compare_maps(om.chunks(), gpu_chunks, ["occupancy"], exact_float=True)
    assert_parity(stats)


def test_cpp_batch_coalescing_and_region_ownership(gpu):
    from ohm_amd import distributed as D
    rays = synth.rays_c1(n=20000, max_range=12.0)
    om = OracleMap(0.1, layers=("occupancy",))
    om.integrate_occupancy(rays)
    expect = om.chunks()
    # GpuMap::setBatchCoalescing: 1000-ray calls, a device batch every fourth call, the rest at syncVoxels()
    merged = run_driver("occcoalesce", 0.1, 1000, rays, 1)
    assert_parity(compare_maps(expect, merged, ["occupancy"], exact_float=True))
    # GpuMap::setRegionOwnership(2, 1): exactly the regions rank 1 owns, with the values the whole map has there
    owned = run_driver("occowner", 0.1, 4096, rays, 1)
    keys = np.array(sorted(expect.keys()), dtype=np.int16).reshape(-1, 3)
    mine = {tuple(int(v) for v in k) for k, o in zip(keys, D.region_owner(keys, 2, 0)) if o == 1}
    assert set(owned.keys()) == mine and len(mine) > 0
    assert_parity(compare_maps({k: expect[k] for k in mine}, owned, ["occupancy"], exact_float=True))


def test_cpp_region_partition_and_ray_routing(gpu):
    """GpuMap::setRegionPartition + GpuMap::routeRays through the C++ mirror: rank 1 of a two-way table partition routes
    every batch on the device and integrates the block addressed to it -> exactly the regions the table gives rank 1,
    with the values the whole map has there."""
    from ohm_amd import distributed as D
    rays = synth.rays_c1(n=20000, max_range=12.0)
    om = OracleMap(0.1, layers=("occupancy",))
    for i in range(0, rays.shape[0], 2 * 4096):
        om.integrate_occupancy(rays[i:i + 2 * 4096])
    expect = om.chunks()
    owned = run_driver("occpart", 0.1, 4096, rays, 1)
    part = D.RegionPartition(2, 1, 1, (0, 0, 0), np.array([0, 1], dtype=np.uint8).reshape(2, 1, 1))
    keys = np.array(sorted(expect.keys()), dtype=np.int16).reshape(-1, 3)
    mine = {tuple(int(v) for v in k) for k, o in zip(keys, part.owners(keys)) if o == 1}
    assert all(k[0] >= 2 for k in mine) and 0 < len(mine) < len(expect)
    assert set(owned.keys()) == mine
    assert_parity(compare_maps({k: expect[k] for k in mine}, owned, ["occupancy"], exact_float=True))


def test_cpp_partitioned_integrator_keeps_batches_in_flight(gpu):
    """ohm::PartitionedIntegrator + ohm::RayCommunicator (ohm_amd/host/OhmGpuMap.h) at world size 1: routing kernels, the
    library's RCCL exchange and the integration of what arrives, batch after batch without a wait -- the caller's ray
    buffer is rewritten as soon as a call returns, the receive buffers rotate.  Batches of 70 000 rays: each is a device
    batch of its own (above the coalescing threshold), so two are in flight while the third is being routed."""
    rays = synth.rays_c1(n=420000, max_range=14.0, seed=61)
    om = OracleMap(0.1, layers=("occupancy",))
    for i in range(0, rays.shape[0], 2 * 70000):
        om.integrate_occupancy(rays[i:i + 2 * 70000])
    got = run_driver("occpartint", 0.1, 70000, rays, 1)
    assert_parity(compare_maps(om.chunks(), got, ["occupancy"], exact_float=True))


def test_cpp_set_ray_filter_clip_box(gpu):
    """GpuMap::setRayFilter with a RayFilterFunction wrapping clipBounded, as GpuMap.ClipBox does
    (tests/ohmtestgpu/GpuMapTest.cpp:633-647): the C++ mirror's Aabb / clipBounded against the numpy restatement."""
    from ohm_amd import rayfilter as RF
    rays = synth.random_rays(6000, extent=4.0, seed=12, origin_spread=2.0)
    filt = RF.clip_bounded(RF.Aabb((-1.0, -1.0, -1.0), (2.0, 2.0, 2.0)))
    gpu_chunks = run_driver("occclipbox", 0.1, 2048, rays, 1)
    om = OracleMap(0.1, layers=("occupancy",))
    for i in range(0, rays.shape[0], 2 * 2048):
        chunk = rays[i:i + 2 * 2048]
        keep, starts, ends, flags = filt(chunk[0::2].copy(), chunk[1::2].copy())
        kept = np.empty((2 * int(keep.sum()), 3))
        kept[0::2] = starts[keep]
        kept[1::2] = ends[keep]
        om.integrate_occupancy(kept, filter_flags=flags[keep])
    assert_parity(compare_maps(om.chunks(), gpu_chunks, ["occupancy"], exact_float=True))


def test_gputil_hip_backend_through_the_reference_headers(gpu):
    """ohm_amd/host/ref_adaptor/gputil_hip: gputil::Device / Queue / Event as the REFERENCE's headers declare them,
    implemented over the C ABI.  The check program is compiled (in the build container, where the reference checkout
    is) against those headers in place and runs here: device enumeration and selection, queue creation, event
    marking, reference counting, waits and callbacks."""
    binary = os.path.join(os.path.dirname(DRIVER), "gputil_hip_check")
    assert os.path.exists(binary), "gputil_hip_check missing: run __graft_entry__.build() where /root/reference exists"
    res = subprocess.run([binary], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "GPUTIL_HIP_OK" in res.stdout, (res.returncode, res.stdout, res.stderr)
