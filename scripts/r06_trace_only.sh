#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for MODE in 1 0; do
OHMHIP_CLAMP_MASK=$MODE OHMHIP_DEBUG_FLAGS=64 OHMHIP_DEBUG_TRACE=/tmp/walk_trace.txt timeout 120 python bench.py --steps 6 --warmup 3 --no-cpu --no-extra > /dev/null 2>&1
python scripts/analyse_trace.py /tmp/walk_trace.txt > gpurun_out/r06/${1:-run3}_walk_trace_clamp$MODE.txt 2>&1; echo "== OHMHIP_CLAMP_MASK=$MODE"; head -24 gpurun_out/r06/${1:-run3}_walk_trace_clamp$MODE.txt
done
