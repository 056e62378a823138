"""Steady-state batch timeline from a rocprofv3 kernel-trace CSV (…_kernel_trace.csv).

Usage: python scripts/timeline.py <dir-or-csv> [batches-from-the-end]

For the last few batches of the run (a batch = the kernels between two k_region_walk starts) prints every kernel with
its stream (queue), start relative to the batch's first kernel, duration, and the gap to the previous kernel's end on
the same queue and on the device as a whole; then the mean interval between walk starts and the mean idle time of the
device per batch (time inside the interval during which no kernel of the process runs)."""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_0-9]+|radix_sort[a-z_]*|onesweep[a-z_]*|fillBuffer\w*|copyBuffer\w*)", name)
    return m.group(1) if m else name[:40]


def load(path):
    if os.path.isdir(path):
        found = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
        if not found:
            raise SystemExit("no *kernel_trace.csv under " + path)
        path = found[0]
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def main():
    rows = load(sys.argv[1])
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    walks = [i for i, r in enumerate(rows) if r[2] == "k_region_walk"]
    if len(walks) < last + 2:
        raise SystemExit("too few walk launches: %d" % len(walks))
    starts = [rows[i][0] for i in walks]
    intervals = [(b - a) / 1e3 for a, b in zip(starts[:-1], starts[1:])]
    tail = intervals[-(last + 8):]
    print("walk-to-walk interval, last %d: mean %.1f us  min %.1f  max %.1f" %
          (len(tail), sum(tail) / len(tail), min(tail), max(tail)))
    # device idle time inside the last intervals
    for k in range(len(walks) - last - 1, len(walks) - 1):
        lo, hi = rows[walks[k]][0], rows[walks[k + 1]][0]
        inside = [r for r in rows if r[1] > lo and r[0] < hi]
        # union of busy spans
        busy = 0
        cur_s, cur_e = None, None
        for s, e, _, _ in sorted(inside):
            s, e = max(s, lo), min(e, hi)
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        print("\n== batch interval %d: %.1f us, device idle %.1f us" % (k, (hi - lo) / 1e3, (hi - lo - busy) / 1e3))
        prev_end_dev = None
        prev_end_q = {}
        for s, e, n, q in sorted(inside):
            gd = "" if prev_end_dev is None else "%7.1f" % ((s - prev_end_dev) / 1e3)
            gq = "" if q not in prev_end_q else "%7.1f" % ((s - prev_end_q[q]) / 1e3)
            print("  %-22s q%-3s start %8.1f  dur %7.1f  gap(dev) %7s  gap(queue) %7s" %
                  (n, q, (s - lo) / 1e3, (e - s) / 1e3, gd, gq))
            prev_end_dev = e if prev_end_dev is None else max(prev_end_dev, e)
            prev_end_q[q] = e


if __name__ == "__main__":
    main()
