#!/bin/bash
# Kernel trace + timeline of the moving-sensor leg (scripts/moving_probe.py): gpurun_out/prof_<tag>_moving
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_${1:-r05}_moving
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python scripts/moving_probe.py > $OUT/trace.log 2>&1
python scripts/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
python scripts/timeline.py $OUT 2 > $OUT/timeline.txt 2>&1
tail -5 $OUT/trace.log
head -18 $OUT/summary.txt
cat $OUT/timeline.txt | head -60
