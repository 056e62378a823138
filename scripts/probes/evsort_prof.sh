cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf /tmp/evp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/evp -o t -- scripts/probes/evsort_probe > /tmp/evp.log 2>&1
tail -3 /tmp/evp.log | cut -c1-160
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/evp/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
k = [r for r in rows if 'evsort' in r['Kernel_Name']][:12]
for r in k:
    print('%7.1f us  grid %8s  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'], r['Kernel_Name'].split('(')[0][-34:]))
PY
