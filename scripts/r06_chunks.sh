#!/bin/bash
# Round 6: chunks of several rounds (kMaxChunkTotal).  A/B by environment on ONE build + the previous library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
line() { timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
dm = {k: round(v, 4) for k, v in d['device_ms'].items() if k != 'note'}
print('ms_per_step %.4f  kernel_ms %.4f  pipeline_frac %.4f  device_ms %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['pipeline_frac'], dm))"; }
for round in 1 2; do
  echo "== seq (previous library)"; OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_seq.so line
  for C in 8192 12288 16384 24576; do echo "== OHMHIP_CHUNK_SEGMENTS=$C"; OHMHIP_CHUNK_SEGMENTS=$C line; done
done 2>&1 | tee gpurun_out/r06/chunks_ab.txt
for C in 16384 24576; do
  echo "== parity with OHMHIP_CHUNK_SEGMENTS=$C"
  OHMHIP_CHUNK_SEGMENTS=$C timeout 900 python -m pytest tests -x -q -m gpu -k "occupancy or full_configs or ndt_tsdf or secondary or host_batches or partitioned" 2>&1 | tail -3
done 2>&1 | tee gpurun_out/r06/chunks_parity.txt
