#!/bin/bash
# Kernel trace of the first passes over a fresh map (scripts/first_pass_probe.py): every kernel in time order
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_${1:-r05}_first
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python scripts/first_pass_probe.py ${2:-1} > $OUT/trace.log 2>&1
tail -8 $OUT/trace.log | grep "^rep"
cd scripts && python trace_list.py ../$OUT 200 > ../$OUT/list.txt 2>&1; cd ..
cat $OUT/list.txt
