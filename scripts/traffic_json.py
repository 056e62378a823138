#!/usr/bin/env python
"""Per-kernel HBM counter traffic of a profile directory (scripts/profile_bench.sh <tag> pmc, or profile_modes.sh) as
JSON: for every kernel the per-launch FETCH_SIZE / WRITE_SIZE (KiB, separate --pmc passes) and the byte figures bench.py
reports -- upper = 2 x FETCH + WRITE (the guide's correction for coalesced streams), lower = FETCH + WRITE (gathers of
<= 64 bytes: profiles/r02_fetch_calibration.txt) -- plus the sum over the kernels of one batch.
usage: python scripts/traffic_json.py gpurun_out/prof_<tag> "<command the profile was taken with>" > profiles/<name>.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
command = sys.argv[2] if len(sys.argv) > 2 else ""


def short(name):
    m = re.search(r"(k_[a-z_0-9]+|radix_sort[a-z_]*|fillBuffer\w*|copyBuffer\w*)", name)
    return m.group(1) if m else name[:40]


def counters(pattern, counter):
    agg, launches = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(out, "**", pattern), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                if row.get("Grid_Size") and row.get("Grid_Size") == row.get("Workgroup_Size") and "k_region_walk" in row.get("Kernel_Name", ""):
                    continue  # the one-workgroup launches of map creation (queue scratch warm-up), not batches
                k = short(row.get("Kernel_Name", "?"))
                agg[k] += float(row["Counter_Value"] or 0)
                launches[k].add(row.get("Dispatch_Id"))
    return {k: (v / max(len(launches[k]), 1), len(launches[k])) for k, v in agg.items()}


fetch = counters("*counter_collection.csv", "FETCH_SIZE")
write = counters("*counter_collection.csv", "WRITE_SIZE")
# launches per batch: relative to the walk kernel (one per batch)
walk_launches = fetch.get("k_region_walk", (0, 1))[1] or 1
kernels = {}
batch_upper = batch_lower = 0.0
for k in sorted(set(fetch) | set(write)):
    f_kb, n = fetch.get(k, (0.0, 0))
    w_kb, n2 = write.get(k, (0.0, 0))
    n = max(n, n2)
    per_batch = n / walk_launches
    if per_batch < 0.5 or k.startswith(("fillBuffer", "copyBuffer", "k_fill_u")):
        continue  # start-up work (pool initialisation dominates the runtime's fill / copy kernels), not part of a batch
    upper = (2.0 * f_kb + w_kb) * 1024.0
    lower = (f_kb + w_kb) * 1024.0
    kernels[k] = {"launches_per_batch": round(per_batch, 2), "fetch_size_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
                  "bytes_upper": upper, "bytes_lower": lower}
    batch_upper += upper * per_batch
    batch_lower += lower * per_batch
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ohm_amd import _lib as _L
    build_id = _L.lib.ohmhip_build_id().decode()
except Exception:
    build_id = None
doc = {"source": command, "library_build_id": build_id, "unit_note": "FETCH_SIZE / WRITE_SIZE in KiB per launch; bytes_upper = 2 x FETCH + WRITE (coalesced "
       "streams move 128 B per counted 64 B request: MI355X_MICROARCH.md), bytes_lower = FETCH + WRITE (<= 64 B gathers are "
       "one request each: profiles/r02_fetch_calibration.txt)",
       "kernels": kernels, "batch_bytes_upper": batch_upper, "batch_bytes_lower": batch_lower}
print(json.dumps(doc, indent=1))
