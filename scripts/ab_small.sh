#!/bin/bash
# Usage (on the GPU box): scripts/ab_small.sh <tag> ...  -- bench line at 65 536 / 100 000 / 1 000 000 rays per batch per library variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for TAG in "$@"; do
  if [ "$TAG" = "default" ]; then unset OHMHIP_LIB; else export OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_$TAG.so; fi
  for N in 65536 100000 1000000; do
    timeout 120 python bench.py --rays $N --steps 40 --warmup 10 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
dm = {k: round(v, 4) for k, v in d['device_ms'].items() if k != 'note'}
print('$TAG rays $N: ms_per_step %.4f  %.3e rays/s  device_ms %s' % (d['ms_per_step'], d['value'], dm))"
  done
done
