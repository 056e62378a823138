#!/usr/bin/env python
"""CPU simulation of k_region_walk's lane -> tile-address pattern on C1 (development tool, numpy only).

For a few regions at different ranges it rebuilds the ray-region segments (float DDA -- good enough for statistics),
splits them into chunks of <= 8192 segments, orders the segments of a chunk in several candidate ways, plays the
kernel's schedule (16 waves, lanes refilled from a shared cursor once >= 20 are idle) and counts, per wave-step:
  lanes     active lanes
  distinct  distinct tile words among them (what a perfect in-wave combine would issue)
  runs      maximal runs of ADJACENT active lanes with the same voxel (what a DPP neighbour combine would issue)
  worst     largest number of lanes on one tile word (same-address serialisation of the returning add)
"""
import sys
import numpy as np

sys.path.insert(0, ".")
from ohm_amd import synth  # noqa: E402

RES, DIM = 0.1, 32
RSIZE = RES * DIM


def segments_in_region(rays, key):
    """Rays clipped to the region box: entry voxel + voxel sequences (float DDA)."""
    o = rays[0::2]
    e = rays[1::2]
    d = e - o
    lo = (np.array(key) - 0.5) * RSIZE
    hi = lo + RSIZE
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - o) / d
        t1 = (hi - o) / d
    tmin = np.nanmax(np.minimum(t0, t1), axis=1)
    tmax = np.nanmin(np.maximum(t0, t1), axis=1)
    tmin = np.maximum(tmin, 0.0)
    tmax = np.minimum(tmax, 1.0)
    hit = tmax > tmin + 1e-9
    idx = np.nonzero(hit)[0]
    seqs = []
    for i in idx:
        a = o[i] + d[i] * (tmin[i] + 1e-9)
        b = o[i] + d[i] * (tmax[i] - 1e-9)
        va = np.clip(np.floor((a - lo) / RES).astype(int), 0, DIM - 1)
        vb = np.clip(np.floor((b - lo) / RES).astype(int), 0, DIM - 1)
        n = int(np.abs(vb - va).sum()) + 1
        # DDA
        step = np.sign(d[i]).astype(int)
        with np.errstate(divide="ignore"):
            tdelta = np.where(d[i] != 0, RES / np.abs(d[i]), np.inf)
            nxt = np.where(step > 0, (lo + (va + 1) * RES - o[i]) / d[i], np.where(step < 0, (lo + va * RES - o[i]) / d[i], np.inf))
        v = va.copy()
        seq = np.empty(n, dtype=np.int32)
        for k in range(n):
            seq[k] = v[0] + DIM * (v[1] + DIM * v[2])
            ax = 0
            if nxt[1] <= nxt[ax]:
                ax = 1
            if nxt[2] <= nxt[ax]:
                ax = 2
            v[ax] += step[ax]
            nxt[ax] += tdelta[ax]
            if v[ax] < 0 or v[ax] >= DIM:
                seq = seq[:k + 1]
                break
        seqs.append(seq)
    return idx, seqs


def tile_word(vi):
    w = vi >> 1
    return w ^ (((w >> 5) ^ (w >> 10)) & 31)


def simulate(seqs, order, waves=16, refill_idle=20):
    n = len(order)
    cursor = 0
    lane_seg = -np.ones((waves, 64), dtype=np.int64)
    lane_pos = np.zeros((waves, 64), dtype=np.int64)
    lane_len = np.zeros((waves, 64), dtype=np.int64)
    exhausted = False
    stats = np.zeros(5, dtype=np.int64)  # wave_steps, lanes, distinct, runs, worst-sum
    bank_cycles = np.zeros(4, dtype=np.int64)
    alive = [True] * waves
    while any(alive):
        for w in range(waves):
            if not alive[w]:
                continue
            active = lane_pos[w] < lane_len[w]
            idle = ~active
            n_idle = int(idle.sum())
            thr = 64 if exhausted else refill_idle
            if n_idle >= thr:
                if cursor >= n:
                    exhausted = True
                    if n_idle == 64:
                        alive[w] = False
                        continue
                else:
                    take = min(n_idle, n - cursor)
                    lanes = np.nonzero(idle)[0][:take]
                    for j, ln in enumerate(lanes):
                        s = order[cursor + j]
                        lane_seg[w, ln] = s
                        lane_pos[w, ln] = 0
                        lane_len[w, ln] = len(seqs[s])
                    cursor += take
                    if cursor >= n:
                        exhausted = True
                    active = lane_pos[w] < lane_len[w]
            if not active.any():
                if exhausted:
                    alive[w] = False
                continue
            lanes = np.nonzero(active)[0]
            vox = np.array([seqs[lane_seg[w, ln]][lane_pos[w, ln]] for ln in lanes])
            words = tile_word(vox)
            uw, counts = np.unique(words, return_counts=True)
            # runs of adjacent active lanes (lane indices consecutive) with the same voxel
            brk = (np.diff(lanes) != 1) | (np.diff(vox) != 0)
            runs = 1 + int(brk.sum())
            # LDS cycles of the returning add ~ the most loaded bank, every lane an operation (same-address atomics
            # serialise like any other bank conflict): as issued today, ...
            cyc_now = int(np.bincount(words & 31, minlength=32).max())
            # ... with quad combining (lanes of an aligned quad equal to the quad's first lane fold into it), ...
            quad_first = lanes - (lanes & 3)
            first_vox = {}
            for ln, v in zip(lanes, vox):
                if (ln & 3) == 0:
                    first_vox[ln] = v
            keep = np.array([not ((ln & 3) != 0 and first_vox.get(q, -1) == v) for ln, q, v in zip(lanes, quad_first, vox)])
            cyc_quad = int(np.bincount(words[keep] & 31, minlength=32).max())
            n_quad = int(keep.sum())
            # ... and with a perfect combine per voxel
            uv = np.unique(vox)
            cyc_perfect = int(np.bincount(tile_word(uv) & 31, minlength=32).max())
            bank_cycles += np.array([cyc_now, cyc_quad, cyc_perfect, n_quad])
            stats += (1, len(lanes), len(uw), runs, int(counts.max()))
            lane_pos[w, lanes] += 1
    return stats, bank_cycles


def dealt(order, cls_sorted, stride):
    """Inside every length class, hand the (entry-voxel sorted) items out with a large stride."""
    out = order.copy()
    start = 0
    n = len(order)
    while start < n:
        end = start
        while end < n and cls_sorted[end] == cls_sorted[start]:
            end += 1
        m = end - start
        st = stride % m if m > 1 else 1
        while m > 1 and np.gcd(st, m) != 1:
            st += 1
        out[start:end] = order[start + (np.arange(m) * st) % m]
        start = end
    return out


def main():
    rays = synth.rays_c1()
    keys = [(0, 0, 0), (2, 1, 0), (4, 3, 0), (6, 2, -1), (8, 1, 0)]
    if len(sys.argv) > 1:
        keys = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for key in keys:
        idx, seqs = segments_in_region(rays, key)
        if not len(idx):
            continue
        lens = np.array([len(s) for s in seqs])
        entry = np.array([s[0] for s in seqs])
        n = min(len(idx), 8192)
        # one chunk: the first 8192 in ray order (k_ray_bin's bucket order is roughly ray order)
        pick = np.arange(n)
        cls = np.minimum(lens[pick], 127)
        beam = idx[pick] % 64
        az = idx[pick] // 64
        orders = {
            "len,ray (today, approx)": np.lexsort((idx[pick], -cls)),
            "len,entry voxel": np.lexsort((idx[pick], entry[pick], -cls)),
            "len/4,entry voxel": np.lexsort((idx[pick], entry[pick], -(cls // 4))),
            "entry voxel,len": np.lexsort((idx[pick], -cls, entry[pick])),
            "len,beam,azimuth": np.lexsort((az, beam, -cls)),
            "random within len": np.lexsort((np.random.default_rng(1).permutation(n), -cls)),
        }
        base = orders["len,entry voxel"]
        stride = 1031  # coprime with everything in sight: consecutive lanes come from far-apart places of the sorted list
        orders["len,entry voxel, dealt"] = dealt(base, cls[base], stride)
        orders["entry-sorted, dealt (no len)"] = np.lexsort((idx[pick], entry[pick]))[(np.arange(n) * stride) % n]
        print(f"region {key}: {len(idx)} segments, mean len {lens.mean():.1f}, chunk of {n}")
        for name, order in orders.items():
            st, bank = simulate(seqs, pick[order])
            ws, lanes, distinct, runs, worst = st
            print(f"  {name:30s} steps {ws:5d} lanes {lanes / ws:5.1f} distinct {distinct / ws:5.1f} runs {runs / ws:5.1f} "
                  f"worst {worst / ws:5.2f} | LDS cycles/step: now {bank[0] / ws:5.2f} quad {bank[1] / ws:5.2f} "
                  f"(adds {bank[3] / ws:4.1f}) perfect {bank[2] / ws:5.2f}")


if __name__ == "__main__":
    main()
