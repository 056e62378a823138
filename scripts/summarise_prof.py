"""Summarise rocprofv3 outputs (kernel stats + PMC counter CSVs) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 14:
                print("  ", ", ".join(row[:8]))

for f in find("*counter_collection.csv"):
    print("== counters:", os.path.relpath(f, out))
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    with open(f) as fh:
        rd = csv.DictReader(fh)
        for row in rd:
            k = row.get("Kernel_Name", "?")[:60]
            c = row.get("Counter_Name")
            v = float(row.get("Counter_Value", 0) or 0)
            agg[k][c] += v
            if c == list(agg[k].keys())[0]:
                cnt[k] += 1
    for k, cs in agg.items():
        n = max(cnt[k], 1)
        print("  ", k, "launches", n)
        for c, v in cs.items():
            print(f"      {c:28s} per-launch {v / n:18.1f}")
