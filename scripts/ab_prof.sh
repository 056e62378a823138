#!/bin/bash
# Usage (on the GPU box): scripts/ab_prof.sh <tag> [<tag> ...]   -- per-kernel average times (rocprofv3 --kernel-trace
# --stats of the bench command) per library variant built with scripts/build_variant.sh; "default" = the product build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for TAG in "$@"; do
  if [ "$TAG" = "default" ]; then unset OHMHIP_LIB; else export OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_$TAG.so; fi
  OUT=/tmp/abprof_$TAG
  rm -rf $OUT
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python bench.py --steps 20 --warmup 5 --no-cpu --no-extra > /tmp/abprof_$TAG.log 2>&1
  echo "== $TAG"
  python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    out = []
    for r in rows:
        name = r['Name'].split('(')[0].split('::')[-1]
        out.append('%s %.1f' % (name[:24], float(r['AverageNs']) / 1000.0))
    print('   ' + ' | '.join(out[:9]))
PY
done
