#!/bin/bash
# Usage (on the GPU box): scripts/prof_cmd.sh <python script> [args...]  -- per-kernel average times of one command
# (rocprofv3 --kernel-trace --stats), run from the repo root, bounded by a timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=/tmp/prof_cmd
rm -rf $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python "$@" > /tmp/prof_cmd.log 2>&1
tail -2 /tmp/prof_cmd.log | cut -c1-200
python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        name = r['Name'].split('(')[0].split('::')[-1]
        print('   %-40s calls %6s avg %10.1f us  %5s %%' % (name[:40], r['Calls'], float(r['AverageNs']) / 1000.0, r['Percentage'][:5]))
PY
