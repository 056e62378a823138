"""Ray filters for GpuMap.setRayFilter -- the reference's `RayFilterFunction`s (ohm/RayFilter.h:45-100,
ohm/RayFilter.cpp:12-93), vectorised over a batch.

A filter here is a callable `f(starts, ends) -> (keep, starts, ends, flags)`: `starts` / `ends` are (N, 3) float64
arrays, `keep` a bool mask of the rays that survive, `flags` the per-ray `RayFilterFlag` bits (uint8).  The reference
calls its filter once per ray with pointers (`bool(dvec3 *start, dvec3 *end, unsigned *filter_flags)`); the arithmetic
below follows the same statements ray by ray, so results are those of the per-ray form.
"""
import numpy as np

kRffInvalid = 1  # ohm/RayFilter.h:21-29
kRffClippedStart = 2
kRffClippedEnd = 4


def _finite(v):
    return np.all(np.isfinite(v), axis=1)


def good_ray_filter(max_range=0.0):
    """ohm::goodRayFilter (ohm/RayFilter.cpp:12-34): reject NaN / inf rays and rays longer than max_range (> 0)."""
    def f(starts, ends):
        ray = ends - starts
        len2 = (ray[:, 0] * ray[:, 0] + ray[:, 1] * ray[:, 1]) + ray[:, 2] * ray[:, 2]
        keep = _finite(starts) & _finite(ends)
        if max_range > 0:
            with np.errstate(invalid="ignore"):
                keep &= len2 <= max_range * max_range
        return keep, starts, ends, np.zeros(len(starts), dtype=np.uint8)
    return f


def clip_ray_filter(max_length):
    """ohm::clipRayFilter (ohm/RayFilter.cpp:37-58): shorten rays longer than max_length, flagging the clipped end."""
    def f(starts, ends):
        keep = _finite(starts) & _finite(ends)
        ray = ends - starts
        len2 = (ray[:, 0] * ray[:, 0] + ray[:, 1] * ray[:, 1]) + ray[:, 2] * ray[:, 2]
        with np.errstate(invalid="ignore"):
            clip = keep & (max_length > 0) & (len2 > max_length * max_length)
        ends = ends.copy()
        flags = np.zeros(len(starts), dtype=np.uint8)
        if clip.any():
            unit = ray[clip] / np.sqrt(len2[clip])[:, None]
            ends[clip] = starts[clip] + unit * max_length
            flags[clip] |= kRffClippedEnd
        return keep, starts, ends, flags
    return f


class Aabb:
    """The parts of ohm::Aabb (ohm/Aabb.h) the ray filters use."""

    def __init__(self, min_ext, max_ext):
        self.min = np.asarray(min_ext, dtype=np.float64).reshape(3)
        self.max = np.asarray(max_ext, dtype=np.float64).reshape(3)

    def contains(self, points):
        """Aabb::contains (ohm/Aabb.h:301-312, epsilon 0): closed box."""
        return np.all((points >= self.min) & (points <= self.max), axis=1)

    def _ray_intersect(self, origin, direction):
        """Aabb::rayIntersect (ohm/Aabb.h:330-383): slab test; returns (hit, t_entry, t_exit)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / direction
            sign = (direction < 0.0).astype(np.int64)
            corners = np.stack([self.min, self.max])  # [2, 3]
            near = corners[sign, np.arange(3)]        # (N, 3): corners_[sign[a]][a]
            far = corners[1 - sign, np.arange(3)]
            t0 = (near[:, 0] - origin[:, 0]) * inv[:, 0]
            t1 = (far[:, 0] - origin[:, 0]) * inv[:, 0]
            miss = np.zeros(len(origin), dtype=bool)
            for a in (1, 2):
                tmin = (near[:, a] - origin[:, a]) * inv[:, a]
                tmax = (far[:, a] - origin[:, a]) * inv[:, a]
                miss |= (t0 > tmax) | (tmin > t1)
                t0 = np.where((tmin > t0) | np.isnan(t0), tmin, t0)
                t1 = np.where((tmax < t1) | np.isnan(t1), tmax, t1)
        return ~miss, t0, t1

    def clip_line(self, starts, ends):
        """Aabb::clipLine (ohm/Aabb.h:386-446, allow_clamp false): returns (clipped_any, starts, ends, clip_flags) with
        clip flag 1 = start moved, 2 = end moved."""
        origin = starts
        direction = ends - starts
        d2 = (direction[:, 0] * direction[:, 0] + direction[:, 1] * direction[:, 1]) + direction[:, 2] * direction[:, 2]
        live = ~(d2 < 1e-9)  # degenerate rays are returned untouched
        with np.errstate(divide="ignore", invalid="ignore"):
            length = np.sqrt(d2)
            unit = direction / length[:, None]
        hit, t0, t1 = self._ray_intersect(origin, unit)
        hit &= live
        with np.errstate(invalid="ignore"):
            move_start = hit & (t0 > 0) & (t0 < length)
            move_end = hit & (t1 > 0) & (t1 < length)
        new_starts = np.where(move_start[:, None], origin + unit * t0[:, None], starts)
        new_ends = np.where(move_end[:, None], origin + unit * t1[:, None], ends)
        flags = move_start.astype(np.uint8) * 1 + move_end.astype(np.uint8) * 2
        return move_start | move_end, new_starts, new_ends, flags


def clip_bounded(box):
    """ohm::clipBounded (ohm/RayFilter.cpp:61-78): clip each ray to `box`; a ray that was clipped and still has neither
    end inside the box is rejected."""
    def f(starts, ends):
        clipped, new_starts, new_ends, clip_flags = box.clip_line(starts, ends)
        reject = clipped & ~box.contains(new_starts) & ~box.contains(new_ends)
        flags = ((clip_flags & 1) != 0).astype(np.uint8) * kRffClippedStart + \
            ((clip_flags & 2) != 0).astype(np.uint8) * kRffClippedEnd
        return ~reject, new_starts, new_ends, flags
    return f


def clip_to_bounds(box):
    """ohm::clipToBounds (ohm/RayFilter.cpp:81-93): samples inside `box` are marked as clipped ends (no hit)."""
    def f(starts, ends):
        flags = box.contains(ends).astype(np.uint8) * kRffClippedEnd
        return np.ones(len(starts), dtype=bool), starts, ends, flags
    return f
