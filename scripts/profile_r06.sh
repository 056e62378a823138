#!/bin/bash
# Round-6 profile set, one gpurun call: C1 trace + counter passes + steady-state timeline, C2 / C3 traces + traffic passes,
# traffic JSONs, the event-cost probe, the default bench line.  Everything lands under gpurun_out/ and is copied into
# profiles/ by hand (the traffic JSONs at once: bench.py quotes them by build id).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
bash scripts/profile_bench.sh r06final pmc > gpurun_out/prof_r06final_stdout.txt 2>&1
python scripts/timeline.py gpurun_out/prof_r06final 3 > gpurun_out/prof_r06final/timeline.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r06final "python bench.py --steps 20 --warmup 5 --no-cpu --no-extra" > gpurun_out/r06_traffic.json
bash scripts/profile_modes.sh r06 ndt 4 > gpurun_out/prof_r06_ndt_stdout.txt 2>&1
python scripts/timeline.py gpurun_out/prof_r06_ndt 1 > gpurun_out/prof_r06_ndt/timeline.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r06_ndt "python scripts/profile_modes.py ndt 4" > gpurun_out/r06_traffic_c2_ndt.json
bash scripts/profile_modes.sh r06 tsdf 3 > gpurun_out/prof_r06_tsdf_stdout.txt 2>&1
python scripts/timeline.py gpurun_out/prof_r06_tsdf 1 > gpurun_out/prof_r06_tsdf/timeline.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r06_tsdf "python scripts/profile_modes.py tsdf 3" > gpurun_out/r06_traffic_c3_tsdf.json
cp gpurun_out/r06_traffic.json gpurun_out/r06_traffic_c2_ndt.json gpurun_out/r06_traffic_c3_tsdf.json profiles/
timeout 1200 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -c 400 gpurun_out/r06_bench_default.err
head -30 gpurun_out/prof_r06final/summary.txt
head -30 gpurun_out/prof_r06final/timeline.txt
