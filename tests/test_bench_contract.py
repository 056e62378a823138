"""CPU: the committed default bench line (profiles/r05_bench_default.json, produced by `python bench.py` on an MI355X in
the same gpurun call as profiles/r05_profile_final.txt and profiles/r05_traffic.json: scripts/profile_r05.sh) carries every
field the bench contract names, with consistent arithmetic."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as fh:
        line = json.load(fh)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "f64"
    assert "workload" in line["config"] and "model" not in line["config"]
    rays = line["config"]["rays_per_step_per_gpu"]
    assert abs(line["value"] - rays / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_lower"):
        assert key in roof, key
    with open(os.path.join(ROOT, "profiles", "r05_traffic.json")) as fh:
        traffic = json.load(fh)
    walk = traffic["kernels"]["k_region_walk"]
    assert roof["traffic"] == walk["bytes_upper"] and roof["traffic_lower"] == walk["bytes_lower"]
    assert roof["pipeline_traffic"] == traffic["batch_bytes_upper"]
    assert roof["pipeline_traffic_lower"] == traffic["batch_bytes_lower"]
    assert 2000.0 < roof["peak_measured_copy"] < roof["peak"]  # a device-to-device copy, GB/s read + write
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # Basis since round 5: `achieved` / `frac` charge the algorithmic bytes to the whole batch interval
    # on the device (every kernel of integrateRays) -- the same basis the C2 / C3 blocks use; the dominant kernel alone is
    # kernel_achieved / kernel_frac.
    step_s = line["ms_per_step"] * 1e-3
    assert abs(roof["pipeline_ms"] * 1e-3 - step_s) / step_s < 0.02
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["pipeline_ms"] * 1e-3) / 1e9) < 1e-3 * roof["achieved"]
    assert roof["frac"] == roof["pipeline_frac"] and 0.30 <= roof["frac"] < roof["kernel_frac"]
    # the dominant kernel: algorithmic bytes per launch / the kernel's average duration (HIP stop events)
    assert abs(roof["kernel_achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms"] * 1e-3) / 1e9) < 1e-3 * roof["kernel_achieved"]
    assert abs(roof["kernel_frac"] - roof["kernel_achieved"] / roof["peak"]) < 1e-9 and roof["kernel_frac"] >= 0.40
    assert roof["kernel_ms"] < roof["pipeline_ms"]
    # the two other C1 workloads the review asks for, top level: a fresh map's first pass and the moving sensor
    for key in ("first_pass", "moving_sensor"):
        assert line[key]["ms_per_step"] > line["ms_per_step"] and 0.0 < line[key]["pipeline_frac"] < roof["frac"], key
    visits = line["config"]["voxel_visits_per_step"]
    assert roof["algorithmic_bytes_per_launch"] == 44 * rays + 8 * visits
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] == 1 and cpu["unit"] == "rays/s"
    # the other CPU legs SURVEY 8d asks for: C0 on its own rays, and replicas over all physical cores
    assert cpu["c0"]["cores"] == 1 and cpu["c0"]["value"] > 0 and "C0" in cpu["c0"]["sample"]
    assert cpu["all_cores"]["cores"] > 1 and cpu["all_cores"]["value"] > cpu["value"]
    # every BASELINE config has a line: C0, C2, C3 (+ cache stress), C4 (one-GPU stand-in with the merge deviation)
    other = line["other_configs"]
    for key in ("C0_100k_rays_10m", "C2_ndt_1M_rays_0.2m", "C3_tsdf_4M_rays_0.05m", "C3_tsdf_cache_stress_1GiB",
                "C4_8_shards_one_gpu_replica_merge", "C1_4096_ray_batches"):
        assert key in other and "error" not in other[key], key
    dev = other["C4_8_shards_one_gpu_replica_merge"]["deviation_vs_sequential"]
    assert dev["voxels_state_differs"] == 0 and dev["regions_compared"] > 1000
    assert "walk_kernel_frac" not in other["C2_ndt_1M_rays_0.2m"]["roofline"]


def test_traffic_file_matches_the_profile_it_cites():
    with open(os.path.join(ROOT, "profiles", "r05_traffic.json")) as fh:
        traffic = json.load(fh)
    walk = traffic["kernels"]["k_region_walk"]
    assert abs(walk["bytes_upper"] - (2.0 * walk["fetch_size_kb"] + walk["write_size_kb"]) * 1024.0) < 1024.0
    assert abs(walk["bytes_lower"] - (walk["fetch_size_kb"] + walk["write_size_kb"]) * 1024.0) < 1024.0
    total = sum(k["bytes_upper"] * k["launches_per_batch"] for k in traffic["kernels"].values())
    assert abs(total - traffic["batch_bytes_upper"]) < 1e-6 * total
    summary = open(os.path.join(ROOT, "profiles", "r05_profile_final.txt")).read()
    assert "k_region_walk" in summary and "FETCH_SIZE" in summary and "WRITE_SIZE" in summary
    # the profile's average duration of the dominant kernel agrees with the bench line's HIP-event figure
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as fh:
        line = json.load(fh)
    row = [ln for ln in summary.splitlines() if ln.strip().startswith("k_region_walk")][0].split()
    avg_us = float(row[3])
    assert abs(avg_us * 1e-3 - line["roofline"]["kernel_ms"]) / line["roofline"]["kernel_ms"] < 0.05
    assert f"{walk['fetch_size_kb']:.1f}" in summary and f"{walk['write_size_kb']:.1f}" in summary


def test_gpus_n_without_a_launcher_environment_starts_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks itself (VERDICT r2 missing 2).  With every GPU
    hidden the product path cannot run -- there is no CPU fallback -- so the ranks only rendezvous (gloo, 127.0.0.1),
    count each other and rank 0 prints a line with n_gpus = 2, value = null and the reason."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo"
    assert line["value"] is None and "no HIP device" in line["error"]
