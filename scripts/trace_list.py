"""Every kernel of a rocprofv3 kernel trace in time order: queue, start (relative to the first kernel, or to the
previous k_ray_setup with `--per-pass`), duration, gap to the previous kernel's end on the device.
Usage: python scripts/trace_list.py <dir-or-csv> [max rows]"""
import sys
from timeline import load

rows = load(sys.argv[1])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = rows[0][0]
prev_end = None
for s, e, n, q in rows[:limit]:
    gap = "" if prev_end is None else "%8.1f" % ((s - prev_end) / 1e3)
    print("%-28s q%-3s start %10.1f  dur %8.1f  gap(dev) %8s" % (n, q, (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end = e if prev_end is None else max(prev_end, e)
