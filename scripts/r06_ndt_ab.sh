#!/bin/bash
# k_replay_ndt at 3 / 4 / 5 waves per SIMD (launch-bounds variants): C2 batch time + the kernel's time from the trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for TAG in default ndtw4 ndtw5 default ndtw4 ndtw5; do
  if [ "$TAG" = "default" ]; then unset OHMHIP_LIB; else export OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_$TAG.so; fi
  echo "== $TAG"; bash scripts/r06_modes_ab.sh 2>&1 | grep C2
done
for TAG in default ndtw4 ndtw5; do
  if [ "$TAG" = "default" ]; then unset OHMHIP_LIB; else export OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_$TAG.so; fi
  OUT=/tmp/ndtprof_$TAG; rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python scripts/profile_modes.py ndt 4 > /dev/null 2>&1
  echo "== $TAG"; python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_replay_ndt' in r['Name']:
            print('   k_replay_ndt avg %.1f us' % (float(r['AverageNs']) / 1000.0))
PY
done
