#!/bin/bash
# Round-5 profile set, one gpurun call: C1 trace + counter passes + steady-state timeline, C2 / C3 traces + traffic passes,
# traffic JSONs, the event-cost probe, the default bench line.  Everything lands under gpurun_out/ and is copied into
# profiles/ by hand (the traffic JSONs at once: bench.py quotes them by build id).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
bash scripts/profile_bench.sh r05final pmc > gpurun_out/prof_r05final_stdout.txt 2>&1
python scripts/timeline.py gpurun_out/prof_r05final 3 > gpurun_out/prof_r05final/timeline.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r05final "python bench.py --steps 20 --warmup 5 --no-cpu --no-extra" > gpurun_out/r05_traffic.json
bash scripts/profile_modes.sh r05 ndt 4 > gpurun_out/prof_r05_ndt_stdout.txt 2>&1
python scripts/timeline.py gpurun_out/prof_r05_ndt 1 > gpurun_out/prof_r05_ndt/timeline.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r05_ndt "python scripts/profile_modes.py ndt 4" > gpurun_out/r05_traffic_c2_ndt.json
bash scripts/profile_modes.sh r05 tsdf 3 > gpurun_out/prof_r05_tsdf_stdout.txt 2>&1
python scripts/timeline.py gpurun_out/prof_r05_tsdf 1 > gpurun_out/prof_r05_tsdf/timeline.txt 2>&1
python scripts/traffic_json.py gpurun_out/prof_r05_tsdf "python scripts/profile_modes.py tsdf 3" > gpurun_out/r05_traffic_c3_tsdf.json
timeout 120 scripts/probes/event_probe > gpurun_out/r05_event_probe.txt 2>&1
cp gpurun_out/r05_traffic.json gpurun_out/r05_traffic_c2_ndt.json gpurun_out/r05_traffic_c3_tsdf.json profiles/
timeout 1200 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 400 gpurun_out/r05_bench_default.err
head -30 gpurun_out/prof_r05final/summary.txt
head -30 gpurun_out/prof_r05final/timeline.txt
