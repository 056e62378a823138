#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/d1
mkdir -p $OUT
OHMHIP_DEBUG_FLAGS=64 OHMHIP_DEBUG_TRACE=$OUT/trace64.txt timeout 200 python bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $OUT/trace_bench.txt 2> $OUT/trace_stderr.txt
python scripts/analyse_trace.py $OUT/trace64.txt > $OUT/trace64_summary.txt 2>&1
cat $OUT/trace64_summary.txt
