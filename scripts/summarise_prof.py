"""Summarise rocprofv3 CSV outputs (kernel stats + PMC counter CSVs) into a small text table."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


def short(name):
    m = re.search(r"(k_[a-z_]+|radix_sort[a-z_]*|merge_sort[a-z_]*|fillBuffer\w*|copyBuffer\w*)", name)
    return m.group(1) if m else name[:50]


for f in find("*kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    print(f"   {'kernel':34s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows[:16]:
        print(f"   {short(r['Name']):34s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e3:12.1f} "
              f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['MinNs']) / 1e3:10.1f} {float(r['MaxNs']) / 1e3:10.1f} "
              f"{float(r['Percentage']):6.2f}")

for f in find("*counter_collection.csv"):
    print("== counters:", os.path.relpath(f, out))
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Grid_Size") and row.get("Grid_Size") == row.get("Workgroup_Size") and "k_region_walk" in row.get("Kernel_Name", ""):
                continue  # the one-workgroup launches of map creation (queue scratch warm-up), not batches
            k = short(row.get("Kernel_Name", "?"))
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
            launches[k].add(row.get("Dispatch_Id"))
    for k, cs in agg.items():
        if not k.startswith("k_"):
            continue
        n = max(len(launches[k]), 1)
        print(f"   {k}  launches={n}")
        for c, v in cs.items():
            print(f"      {c:26s} per-launch {v / n:18.1f}")
