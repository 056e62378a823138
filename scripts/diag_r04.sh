#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/d1
mkdir -p $OUT
timeout 300 scripts/valu_probe > $OUT/valu_probe.txt 2>&1
cat $OUT/valu_probe.txt
