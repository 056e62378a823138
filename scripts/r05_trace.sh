#!/bin/bash
# Kernel trace of the default bench -> steady-state batch timeline (scripts/timeline.py) under gpurun_out/prof_<tag>.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/bench_trace.log 2>&1
python scripts/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
python scripts/timeline.py $OUT 2 > $OUT/timeline.txt 2>&1
head -16 $OUT/summary.txt
tail -14 $OUT/timeline.txt
