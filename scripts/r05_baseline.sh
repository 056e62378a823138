#!/bin/bash
# Round-5 first call: kernel trace of the default bench at HEAD (timeline of a steady-state batch) + the bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_r05base
mkdir -p $OUT
CMD="python bench.py --steps 20 --warmup 5 --no-cpu --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
python scripts/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
python scripts/timeline.py $OUT 2 > $OUT/timeline.txt 2>&1
head -16 $OUT/summary.txt
cat $OUT/timeline.txt | head -80
bash scripts/ab_bench.sh default
