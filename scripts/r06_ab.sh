#!/bin/bash
# Usage (GPU box): scripts/r06_ab.sh [pytest] <variant tag|default> ...   -- alternating default-bench lines, two rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
if [ "$1" = "pytest" ]; then shift; timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06/ab_pytest.txt 2>&1; tail -3 gpurun_out/r06/ab_pytest.txt; fi
line() { timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
dm = {k: round(v, 4) for k, v in d['device_ms'].items() if k != 'note'}
print('ms_per_step %.4f  kernel_ms %.4f  pipeline_frac %.4f  device_ms %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['pipeline_frac'], dm))"; }
for round in 1 2; do
  for TAG in "$@"; do
    if [ "$TAG" = "default" ]; then unset OHMHIP_LIB; else export OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_$TAG.so; fi
    echo "== $TAG"; line
  done
done
