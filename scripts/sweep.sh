#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in 4096 8192 16384; do for r in 4 8 12 16 24; do
  echo -n "chunk=$c refill=$r : "
  OHMHIP_CHUNK_SEGMENTS=$c OHMHIP_REFILL_MIN_IDLE=$r python bench.py --steps 5 --warmup 2 --no-cpu --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['device_ms']['walk'], d['device_ms']['total'])"
done; done
