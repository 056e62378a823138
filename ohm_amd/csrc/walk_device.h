// walk_device.h -- fp64 device maths shared by the gfx950 kernels: voxel keys, voxel centres, line-walk set-up and
// the closed-form "resume the walk at step (axis, j)" that lets a region workgroup pick a ray up mid-walk.
//
// Everything here follows the CPU instantiation of the reference's shared compute headers (WalkReal = double) with
// the same operation order; the library is compiled with -ffp-contract=off so no FMA is formed, and fp64 divide /
// sqrt are IEEE on gfx950.  Citations are reference file:line.
#ifndef OHMHIP_WALK_DEVICE_H
#define OHMHIP_WALK_DEVICE_H

#include "ohmhip_internal.h"

namespace ohmhip
{
__device__ inline double dInf()
{
  return __longlong_as_double(0x7ff0000000000000ll);
}

/// double -> int the way the host the reference runs on does it (x86 cvttsd2si): values outside the int range and
/// NaN give INT_MIN, where the device conversion would saturate.  Only matters for absurd inputs (points 10^5 km from
/// the map origin), but those still have to produce the same keys as the CPU mapper.
__device__ inline int hostInt(double v)
{
  return (v >= -2147483648.0 && v < 2147483648.0) ? int(v) : int(0x80000000u);
}

/// ohm/MapCoord.h:85-93
__device__ inline int pointToRegionCoord(double coord, double resolution)
{
  return hostInt(floor(coord / resolution + 0.5));
}

/// ohm/MapCoord.h:45-80
__device__ inline int pointToRegionVoxel(double coord, double voxel_resolution, double region_resolution)
{
  const double epsilon = double(1e-6f);
  if (-epsilon <= coord && coord < 0)
  {
    coord = 0;
  }
  else if (coord >= region_resolution && coord - epsilon < region_resolution)
  {
    coord -= epsilon;
  }
  return hostInt(floor(coord / voxel_resolution));
}

/// ohm/OccupancyMap.cpp:859-886 -> ohm/MapRegion.cpp:32-69.  Returns false when the point is not addressable (the
/// reference then hands out Key::kNull).  See keyIsNull() for the one addressable key that still reads as null.
/// @param[out] region Region coordinate per axis.
/// @param[out] local Local voxel coordinate per axis.
/// @param[out] beyond_tiles Optional: set when the point is addressable in the reference's sense and only the tile
///   coordinates of a map cut into tiles leave the key's range.
__device__ inline bool voxelKey(const MapConst &mc, const double p[3], int region[3], int local[3],
                                bool *beyond_tiles = nullptr)
{
  bool ok = true;
  bool tiles_ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    const int coord = pointToRegionCoord(p[a] - mc.origin[a], mc.region_dim[a]);
    // The reference stores the region coordinate in an int16: anything outside wraps to a far-away region, whose local
    // coordinate is then out of range, i.e. the key is null (ohm/MapRegion.cpp:32-69).
    ok = ok && coord >= -32768 && coord <= 32767;
    const double centre = coord * mc.region_dim[a];
    const double region_min = centre - 0.5 * mc.region_dim[a];
    const double pl = p[a] - mc.origin[a] - region_min;
    const int q = pointToRegionVoxel(pl, mc.resolution, mc.region_dim[a]);
    ok = ok && 0 <= q && q < mc.kdim[a];
    // (regions cut into tiles: the tile coordinate coord * tile_split + j has to fit the 16-bit field of the packed
    // key too -- include/ohmhip.h, "LARGE REGIONS")
    tiles_ok = tiles_ok && coord * mc.tile_split[a] >= -32768 && coord * mc.tile_split[a] + (mc.tile_split[a] - 1) <= 32767;
    region[a] = coord;
    local[a] = q;
  }
  if (beyond_tiles)
  {
    *beyond_tiles = ok && !tiles_ok;
  }
  return ok && tiles_ok;
}

/// Key::isNull() is "all three region coordinates == int16 lowest" (ohm/Key.h:206): that one corner region reads as
/// null -- for the line walk -- even when the point is addressable.
__device__ inline bool keyIsNull(bool addressable, const int region[3])
{
  return !addressable || (region[0] == -32768 && region[1] == -32768 && region[2] == -32768);
}

/// ohm/OccupancyMap.h:757-778 (one axis).
__device__ inline double voxelCentreAxis(const MapConst &mc, int a, int region, int local)
{
  double v = double(float(region));
  v *= mc.region_dim[a];
  v -= 0.5 * mc.region_dim[a];
  v += mc.origin[a];
  v += double(local) * mc.resolution;
  v += 0.5 * mc.resolution;
  return v;
}

/// floor division / modulo for global voxel coordinate -> (region, local), or -> (tile, local) with the tile edge.
__device__ inline void splitGlobal(int g, int dim, int &region, int &local)
{
  if ((dim & (dim - 1)) == 0)
  {
    // power-of-two region edge (the default 32): arithmetic shift == floor division, mask == floor modulo
    region = g >> (__ffs(dim) - 1);
    local = g & (dim - 1);
    return;
  }
  int q = g / dim;
  int r = g - q * dim;
  if (r < 0)
  {
    r += dim;
    --q;
  }
  region = q;
  local = r;
}

/// Local voxel coordinate of a global voxel coordinate (floor modulo); a mask when the region edge is a power of two.
__device__ inline int localCoord(int g, int dim)
{
  if ((dim & (dim - 1)) == 0)
  {
    return g & (dim - 1);
  }
  int r = g % dim;
  return (r < 0) ? r + dim : r;
}

/// Voxel centre along one axis from the GLOBAL voxel coordinate, evaluated as the reference evaluates it: from the
/// caller's region coordinate and the voxel's coordinate inside that region (voxelCentreAxis) -- which is what a tile
/// coordinate has to be turned back into first.
__device__ inline double globalVoxelCentreAxis(const MapConst &mc, int a, int g)
{
  int region, local;
  splitGlobal(g, mc.kdim[a], region, local);
  return voxelCentreAxis(mc, a, region, local);
}

/// Time at which the j-th step (j >= 1) along an axis is taken: the value `time_next[axis]` holds after j-1 steps on
/// that axis (ohm/LineWalkCompute.h:299-301 and :375-378).
__device__ inline double stepTime(double init, double delta, int j)
{
  return (j <= 1) ? init : init + delta * double(j - 1);
}

/// Does the i-th step of axis b come before the j-th step of axis a in the walk?  The walk always takes the smallest
/// time_next, ties going to the HIGHER axis index (ohm/LineWalkCompute.h:282-289).
__device__ inline bool stepPrecedes(double tb, int b, double ta, int a)
{
  return tb < ta || (tb == ta && b > a);
}

/// Select one of three per-axis values without indexing an array at run time (run-time indexed locals end up in
/// scratch memory on gfx950; every per-axis quantity in the kernels is therefore a named scalar).
template <typename T>
__device__ inline T sel3(int axis, T v0, T v1, T v2)
{
  return (axis == 0) ? v0 : ((axis == 1) ? v1 : v2);
}

/// Number of steps already taken along axis b at the moment the j-th step of axis a (time ta) is about to be taken.
/// T_b(i) is non-decreasing in i so the preceding steps form a prefix [1, n]; estimate n arithmetically and fix up
/// with the exact predicate so the result is identical to running the reference walk step by step.  The estimate only
/// has to be close: `rdelta` is a (rounded) reciprocal of delta computed once per ray, which keeps the fp64 division
/// out of the per-crossing work.
__device__ inline int stepsBefore(double init, double delta, double rdelta, int total, int b, int a, double ta)
{
  if (total == 0)
  {
    return 0;
  }
  int n;
  if (delta > 0 && delta < dInf())
  {
    const double x = (ta - init) * rdelta;
    // In real arithmetic T_b(i) <= ta  <=>  i - 1 <= x.  The fp64 values the exact predicate compares -- T_b(i) =
    // fl(init + fl(delta * (i - 1))) against ta -- differ from that by rounding of at most 2^-51 * (i + |init| / delta)
    // steps, and x itself carries three roundings (2^-51 * |x|): for |x| < 2^20 and |init| <= 4 delta both stay below
    // 1e-9 steps.  So when x keeps a distance of 1e-5 from every integer the predicate's outcome for every i is the real
    // one -- no tie, no doubt -- and the count follows without evaluating it (the loops below evaluate it at least twice
    // per call; the set-up kernels call this twice per ray-region segment).  Everything else takes the exact route.
    const double whole = floor(x);
    const double frac = x - whole;
    if (frac > 1e-5 && frac < 1.0 - 1e-5 && x > -1048576.0 && x < 1048576.0 && fabs(init) <= 4.0 * delta)
    {
      const int count = int(whole) + 1;
      return (x < 0) ? 0 : ((count > total) ? total : count);
    }
    // T_b(i) <= ta  <=>  i - 1 <= x
    n = (x < 0) ? 0 : ((x >= double(total)) ? total : int(x) + 1);
    n = (n > total) ? total : n;
    while (n < total && stepPrecedes(stepTime(init, delta, n + 1), b, ta, a))
    {
      ++n;
    }
    while (n > 0 && !stepPrecedes(stepTime(init, delta, n), b, ta, a))
    {
      --n;
    }
    return n;
  }
  // Degenerate deltas (zero-length rays): binary search on the monotone predicate.
  int lo = 0;
  int hi = total;
  while (lo < hi)
  {
    const int mid = (lo + hi + 1) >> 1;
    if (stepPrecedes(stepTime(init, delta, mid), b, ta, a))
    {
      lo = mid;
    }
    else
    {
      hi = mid - 1;
    }
  }
  return lo;
}

/// Ray filter: ohm/RayFilter.cpp:12-58 (goodRayFilter / clipRayFilter).  May move `end` (clip).  Returns false for
/// a rejected ray.
__device__ inline bool filterRay(const MapConst &mc, const double start[3], double end[3], bool &clipped_end,
                                 uint32_t ray)
{
  clipped_end = false;
  if (mc.batch_filter_flags)
  {
    // Filtered (and possibly moved) by the caller's RayFilterFunction: only the flag is consumed here -- kRffClippedEnd
    // makes the end voxel part of the ray and suppresses the sample (ohm/RayMapperOccupancy.cpp:209-223).
    clipped_end = (mc.batch_filter_flags[ray] & 4u) != 0;  // kRffClippedEnd, ohm/RayFilter.h:21-29
    return true;
  }
  if (mc.filter_mode == OHMHIP_FILTER_NONE)
  {
    return true;
  }
  bool good = isfinite(start[0]) && isfinite(start[1]) && isfinite(start[2]) && isfinite(end[0]) && isfinite(end[1]) &&
              isfinite(end[2]);
  const double rx = end[0] - start[0];
  const double ry = end[1] - start[1];
  const double rz = end[2] - start[2];
  const double len2 = (rx * rx + ry * ry) + rz * rz;
  if (mc.filter_mode == OHMHIP_FILTER_GOOD)
  {
    good = good && (mc.filter_range <= 0 || len2 <= mc.filter_range * mc.filter_range);
  }
  else if (good && mc.filter_range > 0 && len2 > mc.filter_range * mc.filter_range)
  {
    const double len = sqrt(len2);
    end[0] = start[0] + (rx / len) * mc.filter_range;
    end[1] = start[1] + (ry / len) * mc.filter_range;
    end[2] = start[2] + (rz / len) * mc.filter_range;
    clipped_end = true;
  }
  return good;
}

/// Ray filter + key + line-walk set-up for one ray.  Mirrors, in order:
///   ohm/RayFilter.cpp:12-58 (goodRayFilter / clipRayFilter), ohm/LineWalk.h:112-129 (walkSegmentKeys),
///   ohm/LineWalkCompute.h:188-248 (walkInitRay) and :260-280 (walkCalculateSteps).
/// `start`/`end` may be modified by the clip filter.
__device__ inline void setupRay(const MapConst &mc, double start[3], double end[3], unsigned ray_flags, RayWalk &rw,
                                uint32_t ray)
{
  rw.flags = 0;
  rw.pad = 0;
  rw.length = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    rw.init[a] = rw.delta[a] = 0;
    rw.g0[a] = rw.total[a] = 0;
  }

  bool clipped_end = false;
  if (!filterRay(mc, start, end, clipped_end, ray))
  {
    return;
  }
  rw.flags = kRwPassed;

  int r0[3], l0[3], r1[3], l1[3];
  bool beyond0 = false, beyond1 = false;
  const bool addressable0 = voxelKey(mc, start, r0, l0, &beyond0);
  const bool addressable1 = voxelKey(mc, end, r1, l1, &beyond1);
  const bool ok0 = !keyIsNull(addressable0, r0);
  const bool ok1 = !keyIsNull(addressable1, r1);
  if (!ok0 || !ok1)
  {
    // walkSegmentKeys returns 0 for null keys (ohm/LineWalk.h:119-122): no voxel of the ray is visited.  The mappers
    // still apply the SAMPLE update afterwards with voxelKey(end) and no null check (ohm/RayMapperOccupancy.cpp:234-239,
    // ohm/RayMapperNdt.cpp:280-286): an addressable end point gets its hit although the start was not addressable, and an
    // unaddressable end point lands on Key::kNull -- region (-32768)^3, voxel 0 (an end point in that corner region IS
    // addressable and keeps its voxel; only the walk treats its key as null).  Mirrored here so the maps stay identical: a ray with no walk whose "start voxel" is the voxel the sample goes to.
    const bool include_end = clipped_end || (ray_flags & OHMHIP_RF_END_POINT_AS_FREE);
    // (a map whose regions are cut into tiles cannot address Key::kNull's region -- its tile coordinates leave the
    // packed key's range: there the sample of an unaddressable end point is dropped)
    const bool tiled = (mc.tile_split[1] | mc.tile_split[2]) > 1;
    if (!include_end && !(ray_flags & OHMHIP_RF_EXCLUDE_SAMPLE) && (addressable1 || !tiled))
    {
#pragma unroll
      for (int a = 0; a < 3; ++a)
      {
        rw.g0[a] = addressable1 ? (r1[a] * mc.kdim[a] + l1[a]) : (-32768 * mc.kdim[a]);
      }
      rw.flags = kRwPassed | kRwValid | kRwApplySample;
    }
    // (the caller can see that rays were cut for key-range reasons: ohmhip_map_rays_beyond_tiles)
    rw.flags |= (beyond0 || beyond1) ? unsigned(kRwBeyondTiles) : 0u;
    return;
  }

  // walkInitRay
  double dir[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    dir[a] = end[a] - start[a];
  }
  double length = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
  length = (length > 1e-6) ? sqrt(length) : 0;
  unsigned flags = kRwValid | kRwPassed;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    const int sign = dir[a] < 0;
    flags |= sign ? (kRwSign0 << a) : 0u;
    dir[a] /= length;
    const double dir_inv = (length > 0) ? 1 / dir[a] : 0;
    const double centre = voxelCentreAxis(mc, a, r0[a], l0[a]);
    double vmin = centre - 0.5 * mc.resolution;
    double vmax = centre + 0.5 * mc.resolution;
    const double exit0 = ((sign ? vmin : vmax) - start[a]) * dir_inv;
    const double shift = double(-2 * sign + 1) * mc.resolution;
    vmin += shift;
    vmax += shift;
    double exit1 = ((sign ? vmin : vmax) - start[a]) * dir_inv;
    if (exit1 != dInf())
    {
      exit1 -= exit0;
    }
    rw.init[a] = exit0;
    rw.delta[a] = exit1;
    rw.g0[a] = r0[a] * mc.kdim[a] + l0[a];
    const int g1 = r1[a] * mc.kdim[a] + l1[a];
    const int diff = g1 - rw.g0[a];
    rw.total[a] = diff < 0 ? -diff : diff;
    // The step direction the reference uses comes from the sign of the ray direction, while the step COUNT comes
    // from the key difference.  They agree whenever there is at least one step to take.
  }

  rw.length = length;
  const bool include_end = clipped_end || (ray_flags & OHMHIP_RF_END_POINT_AS_FREE);
  flags |= include_end ? kRwIncludeEnd : 0u;
  flags |= (!include_end && !(ray_flags & OHMHIP_RF_EXCLUDE_SAMPLE)) ? kRwApplySample : 0u;
  flags |= (ray_flags & OHMHIP_RF_EXCLUDE_ORIGIN) ? kRwExcludeStart : 0u;
  flags |= !(ray_flags & OHMHIP_RF_EXCLUDE_RAY) ? kRwWalk : 0u;
  rw.flags = flags;
}

__device__ inline int rwSign(const RayWalk &rw, int a)
{
  return (rw.flags >> (1 + a)) & 1u;
}

__device__ inline int rwDir(const RayWalk &rw, int a)
{
  return 1 - 2 * rwSign(rw, a);
}
}  // namespace ohmhip

#endif  // OHMHIP_WALK_DEVICE_H
