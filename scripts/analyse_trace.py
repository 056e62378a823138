"""Summarise the per-chunk debug trace of k_region_walk (OHMHIP_DEBUG_FLAGS=64 OHMHIP_DEBUG_TRACE=<file>).
Record: chunk n_seg|single<<32 loop_start wave_end[15] hwid clk_start clk_end p0 p1 p2 p3 epilogue_start (10 ns ticks)."""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], dtype=np.uint64)
n_seg = (rows[:, 1] & np.uint64(0xffffffff)).astype(np.int64)
single = ((rows[:, 1] >> np.uint64(32)) & np.uint64(1)).astype(bool)
loop_start = rows[:, 2].astype(np.int64)
ends = rows[:, 3:18].astype(np.int64)
hw = rows[:, 18]
start = rows[:, 19].astype(np.int64)
fin = rows[:, 20].astype(np.int64)
p = rows[:, 21:25].astype(np.int64)
epi_start = rows[:, 25].astype(np.int64)
t0 = start.min()
span = fin.max() - t0
print("chunks", len(rows), "span us", span / 100.0)
tot = fin - start
phases = {
    "loads issued": p[:, 0] - start,
    "tile init": p[:, 1] - p[:, 0],
    "sync1": p[:, 2] - p[:, 1],
    "hist+hits": p[:, 3] - p[:, 2],
    "scan+scatter": loop_start - p[:, 3],
    "loop (mean wave)": ends.mean(axis=1) - loop_start,
    "loop tail (max-mean)": ends.max(axis=1) - ends.mean(axis=1),
    "flush queues+sync": epi_start - ends.max(axis=1),
    "epilogue": fin - epi_start,
}
print("sum of chunk times us", tot.sum() / 100.0, " = CU-equivalents", tot.sum() / span)
for k, v in phases.items():
    print(f"  {k:22s} {v.sum() / tot.sum() * 100:6.2f}%   mean {v.mean() / 100.0:8.2f} us")
for lo, hi in [(0, 256), (256, 1024), (1024, 2048), (2048, 3072), (3072, 1 << 30)]:
    m = (n_seg > lo) & (n_seg <= hi)
    if m.any():
        print(f"n_seg ({lo},{hi}]: chunks {m.sum():5d} segs {n_seg[m].sum():9d} mean us {tot[m].mean() / 100:8.2f} "
              f"loop {(ends.max(axis=1) - loop_start)[m].mean() / 100:8.2f} epi {(fin - epi_start)[m].mean() / 100:7.2f} "
              f"single {single[m].mean():.2f}")
if rows.shape[1] >= 29:
    # finer epilogue stamps (kTrace builds since round 6): end of each direct-apply half, end of the sample replay
    h0 = rows[:, 26].astype(np.int64); h1 = rows[:, 27].astype(np.int64); ih = rows[:, 28].astype(np.int64)
    ok = single & (h0 > 0) & (h1 > 0)
    if ok.any():
        has_ih = ok & (ih > 0)
        print("direct-apply epilogue of single-chunk regions (thread 0's wave), mean us: half0 %.2f  half1 %.2f  sample replay %.2f (%d regions)  rest %.2f" %
              ((h0 - epi_start)[ok].mean() / 100.0, (h1 - h0)[ok].mean() / 100.0,
               ((ih - h1)[has_ih].mean() / 100.0) if has_ih.any() else 0.0, has_ih.sum(),
               (fin - np.where(ih > 0, ih, h1))[ok].mean() / 100.0))
        for lo, hi in [(0, 256), (256, 1024), (1024, 3072), (3072, 1 << 30)]:
            m = ok & (n_seg > lo) & (n_seg <= hi)
            if m.any():
                mi = m & (ih > 0)
                print("   n_seg (%d,%d]: half0 %.2f half1 %.2f replay %.2f" % (lo, hi, (h0 - epi_start)[m].mean() / 100.0, (h1 - h0)[m].mean() / 100.0, ((ih - h1)[mi].mean() / 100.0) if mi.any() else 0.0))
xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
unit = (xcc * 8 + se) * 16 + cu
gaps = []
for u in np.unique(unit):
    m = unit == u
    o = np.argsort(start[m])
    gaps.extend((start[m][o][1:] - fin[m][o][:-1]).tolist())
gaps = np.array(gaps)
print("CUs", len(np.unique(unit)), "gap between chunks on a CU: mean us", gaps.mean() / 100.0)
print("per-XCD: chunks, mean chunk us, last finish us")
for x in np.unique(xcc):
    m = xcc == x
    print("  xcc", x, m.sum(), round(tot[m].mean() / 100.0, 2), (fin[m].max() - t0) / 100.0)
