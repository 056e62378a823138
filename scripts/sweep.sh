#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_occupancy.py -m gpu -x -q 2>&1 | tail -3
for d in 0 15; do for c in 4096 16384; do
  echo -n "dbg=$d chunk=$c: "
  OHMHIP_CHUNK_SEGMENTS=$c OHMHIP_DEBUG_FLAGS=$d python bench.py --steps 5 --warmup 2 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['device_ms'])"
done; done
