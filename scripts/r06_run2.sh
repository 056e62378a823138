#!/bin/bash
# Round-6 call 2: min-clamp mask + unordered small chunks.  GPU tests, then A/B in one call:
#   seq      = the library before the change (variant built from the previous commit)
#   off      = this build with OHMHIP_CLAMP_MASK=0 (same kernels, mask not used)
#   default  = this build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06/run2_pytest.txt 2>&1; tail -3 gpurun_out/r06/run2_pytest.txt
line() { timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
dm = {k: round(v, 4) for k, v in d['device_ms'].items() if k != 'note'}
print('ms_per_step %.4f  kernel_ms %.4f  pipeline_frac %.4f  device_ms %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['pipeline_frac'], dm))"; }
for round in 1 2; do
  echo "== seq";     OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_seq.so line
  echo "== off";     OHMHIP_CLAMP_MASK=0 line
  echo "== default"; line
  echo "== noorder0"; OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_ord0.so line
done 2>&1 | tee gpurun_out/r06/run2_ab.txt
OHMHIP_DEBUG_FLAGS=64 OHMHIP_DEBUG_TRACE=/tmp/walk_trace.txt timeout 120 python bench.py --steps 6 --warmup 3 --no-cpu --no-extra > /dev/null 2>&1
python scripts/analyse_trace.py /tmp/walk_trace.txt > gpurun_out/r06/run2_walk_trace.txt 2>&1; cat gpurun_out/r06/run2_walk_trace.txt | head -22
timeout 600 python bench.py --no-cpu > gpurun_out/r06/run2_bench_full.json 2> gpurun_out/r06/run2_bench_full.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06/run2_bench_full.json').read().strip().splitlines()[-1])
print('full: ms_per_step', d['ms_per_step'], 'first_pass', d.get('first_pass', {}).get('ms_per_step'), 'moving', d.get('moving_sensor', {}).get('ms_per_step'))
PY
