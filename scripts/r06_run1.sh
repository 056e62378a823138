#!/bin/bash
# Round-6 first call: GPU tests, A/B of the concurrent dense sort (default vs -DOHMHIP_SORT_CONCURRENT=0), timeline of the
# default build, per-chunk walk trace, kernel trace of 4096-ray device batches, traversal probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06/run1_pytest.txt 2>&1; tail -3 gpurun_out/r06/run1_pytest.txt
bash scripts/ab_bench.sh default seq default seq > gpurun_out/r06/run1_ab.txt 2>&1; cat gpurun_out/r06/run1_ab.txt
bash scripts/r05_trace.sh r06a > gpurun_out/r06/run1_trace_stdout.txt 2>&1; tail -30 gpurun_out/prof_r06a/timeline.txt
OHMHIP_DEBUG_FLAGS=64 OHMHIP_DEBUG_TRACE=/tmp/walk_trace.txt timeout 120 python bench.py --steps 6 --warmup 3 --no-cpu --no-extra > /dev/null 2>&1
python scripts/analyse_trace.py /tmp/walk_trace.txt > gpurun_out/r06/run1_walk_trace.txt 2>&1; cat gpurun_out/r06/run1_walk_trace.txt
OUT=gpurun_out/prof_r06small; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python scripts/small_batch_probe.py 4096 > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python scripts/summarise_prof.py $OUT 2>&1 | head -24
python - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_r06small/**/*kernel_trace.csv', recursive=True))[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:], r.get('Queue_Id','?')) for r in csv.DictReader(open(f))))
tail = rows[-40:]
t0 = tail[0][0]
for s, e, n, q in tail:
    print('%9.1f %8.1f q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
timeout 120 python scripts/traversal_probe.py 2>&1 | tail -1
