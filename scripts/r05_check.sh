#!/bin/bash
# Round-5 check call: the GPU parity suite (or the part named by $1) + the default bench line per library variant.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
SEL=${1:-tests}
shift
timeout 1500 python -m pytest $SEL -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_tests.txt
cat gpurun_out/r05_tests.txt
bash scripts/ab_bench.sh default "$@" 2>&1 | tee gpurun_out/r05_ab.txt
