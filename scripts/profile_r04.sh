#!/bin/bash
# Round-4 profile set, one gpurun call: C1 trace + counter passes, C2 / C3 traces + traffic passes, traffic JSONs, the
# VALU / LDS issue probe, the default bench line.  Everything lands under gpurun_out/ and is copied into profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
bash scripts/profile_bench.sh r04final pmc > gpurun_out/prof_r04final_stdout.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r04final "python bench.py --steps 20 --warmup 5 --no-cpu --no-extra" > gpurun_out/r04_traffic.json
bash scripts/profile_modes.sh r04 ndt 4 > gpurun_out/prof_r04_ndt_stdout.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r04_ndt "python scripts/profile_modes.py ndt 4" > gpurun_out/r04_traffic_c2_ndt.json
bash scripts/profile_modes.sh r04 tsdf 3 > gpurun_out/prof_r04_tsdf_stdout.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r04_tsdf "python scripts/profile_modes.py tsdf 3" > gpurun_out/r04_traffic_c3_tsdf.json
timeout 300 scripts/valu_probe > gpurun_out/r04_valu_probe.txt 2>&1
# the traffic JSONs have to sit in profiles/ for bench.py to quote them: this run's copies (same build id)
cp gpurun_out/r04_traffic.json gpurun_out/r04_traffic_c2_ndt.json gpurun_out/r04_traffic_c3_tsdf.json profiles/
timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
tail -c 400 gpurun_out/r04_bench_default.err
head -30 gpurun_out/prof_r04final/summary.txt
