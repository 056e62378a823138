#!/bin/bash
# Usage (on the GPU box): scripts/ab_bench.sh <tag> [<tag> ...]   -- default bench line per library variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for TAG in "$@"; do
  if [ "$TAG" = "default" ]; then unset OHMHIP_LIB; else export OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_$TAG.so; fi
  echo "== $TAG"
  timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
dm = {k: round(v, 4) for k, v in d['device_ms'].items() if k != 'note'}
print('ms_per_step %.4f  kernel_ms %.4f  frac %.4f  pipeline_frac %.4f  device_ms %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['pipeline_frac'], dm))"
done
