"""Shared helpers for the CPU-oracle vs HIP parity tests (mirrors compareMaps in the reference's
tests/ohmtestgpu/GpuMapTest.cpp:207-310, but with the north_star bar: identical voxel sets, bit-exact integer
fields, values within 1e-5 relative)."""
import numpy as np

from oracle.oracle import OracleMap


def make_oracle(map_):
    """Build an OracleMap with the same parameters as an ohm_amd.OccupancyMap."""
    om = OracleMap(map_.resolution, map_.region_voxel_dimensions, layers=[n for n in map_.layers])
    om.set_origin(map_.origin)
    mode, rng = map_.ray_filter if map_.ray_filter else ("none", 0.0)
    om.set_ray_filter(mode, rng)
    # non-default probabilities / clamps / saturation travel too
    from oracle.oracle import lib as _olib
    _olib.oracle_map_set_hit_value(om.handle, float(map_.hit_value))
    _olib.oracle_map_set_miss_value(om.handle, float(map_.miss_value))
    _olib.oracle_map_set_min_max(om.handle, float(map_.min_voxel_value), float(map_.max_voxel_value))
    _olib.oracle_map_set_saturation(om.handle, int(map_.saturate_at_min_value), int(map_.saturate_at_max_value))
    return om


def compare_layer(name, cpu, gpu, rel=1e-5, exact=False):
    """cpu, gpu: flat numpy arrays of one region's layer.  Returns the number of differing elements."""
    if cpu.dtype.kind in "ui" or exact:
        return int(np.count_nonzero(cpu.view(np.uint32) != gpu.view(np.uint32)))
    both_inf = np.isinf(cpu) & np.isinf(gpu) & (np.sign(cpu) == np.sign(gpu))
    with np.errstate(invalid="ignore"):
        ok = both_inf | (np.abs(cpu - gpu) <= rel * np.maximum(np.abs(cpu), np.abs(gpu)))
    ok |= (cpu == gpu)
    return int(np.count_nonzero(~ok))


def compare_maps(cpu_chunks, gpu_chunks, layers, rel=1e-5, exact_float=False):
    """Both are {region key: {layer: array}}.  Returns a dict of mismatch statistics; all zeros == parity."""
    stats = {"regions_cpu": len(cpu_chunks), "regions_gpu": len(gpu_chunks), "missing_on_gpu": 0, "extra_on_gpu": 0}
    for name in layers:
        stats["diff_" + name] = 0
    for key, cpu_layers in cpu_chunks.items():
        g = gpu_chunks.get(key)
        if g is None:
            stats["missing_on_gpu"] += 1
            continue
        for name in layers:
            stats["diff_" + name] += compare_layer(name, cpu_layers[name], g[name], rel, exact_float)
    for key, g in gpu_chunks.items():
        if key not in cpu_chunks:
            # A region only the GPU knows is fine iff it is untouched (all clear values).
            occ = g.get("occupancy")
            touched = occ is not None and np.isfinite(occ).any()
            tsdf = g.get("tsdf")
            touched = touched or (tsdf is not None and np.any(tsdf != 0))
            if touched:
                stats["extra_on_gpu"] += 1
    return stats


def assert_parity(stats):
    bad = {k: v for k, v in stats.items() if (k.startswith("diff_") or k in ("missing_on_gpu", "extra_on_gpu")) and v}
    assert not bad, f"CPU/GPU parity failure: {bad} (all stats: {stats})"
