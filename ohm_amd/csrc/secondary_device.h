// secondary_device.h -- the secondary voxel layers the ray mappers maintain next to occupancy:
// incident normal (ohm/VoxelIncidentCompute.h, float maths), touch time (ohm/VoxelTouchTimeCompute.h) and the sample
// part of traversal.  Device restatement of the CPU instantiation; citations are reference file:line.
#ifndef OHMHIP_SECONDARY_DEVICE_H
#define OHMHIP_SECONDARY_DEVICE_H

#include "walk_device.h"

namespace ohmhip
{
/// Optional secondary layers of a batch (null pointers == layer absent / no timestamps).
struct SecondaryLayers
{
  float *traversal;           ///< [slot * region_voxels] accumulated ray length through the voxel
  uint32_t *touch_time;       ///< [slot * region_voxels] ms since the map's first ray
  uint32_t *incident;         ///< [slot * region_voxels] packed incident normal
  const double *timestamps;   ///< per ray, or null
  double time_base;           ///< OccupancyMap::firstRayTime()
};

__device__ inline float fMaxStd(float a, float b) { return (a < b) ? b : a; }  // std::max
__device__ inline float fMinStd(float a, float b) { return (b < a) ? b : a; }  // std::min

/// ohm/VoxelIncidentCompute.h:35-55
__device__ inline void decodeNormal(uint32_t packed, float n[3])
{
  n[0] = (2.0f * (float((packed >> 0) & 0x3FFFu) / 16383.0f)) - 1.0f;
  n[1] = (2.0f * (float((packed >> 15) & 0x3FFFu) / 16383.0f)) - 1.0f;
  n[0] = fMaxStd(-1.0f, fMinStd(n[0], 1.0f));
  n[1] = fMaxStd(-1.0f, fMinStd(n[1], 1.0f));
  n[2] = fMaxStd(-1.0f, fMinStd(1.0f - (n[0] * n[0] + n[1] * n[1]), 1.0f));
  const bool set = (packed & (1u << 30)) != 0;
  n[0] = set ? n[0] : 0.0f;
  n[1] = set ? n[1] : 0.0f;
  n[2] = set ? sqrtf(n[2]) : 0.0f;
  n[2] *= (packed & (1u << 31)) ? -1.0f : 1.0f;
}

/// ohm/VoxelIncidentCompute.h:57-80
__device__ inline uint32_t encodeNormal(const float in[3])
{
  const float nx = 0.5f * (fMaxStd(-1.0f, fMinStd(in[0], 1.0f)) + 1.0f);
  const float ny = 0.5f * (fMaxStd(-1.0f, fMinStd(in[1], 1.0f)) + 1.0f);
  uint32_t n = 0;
  n |= (uint32_t(nx * 16383.0f) & 0x3FFFu) << 0;
  n |= (uint32_t(ny * 16383.0f) & 0x3FFFu) << 15;
  n &= ~((1u << 30) | (1u << 31));
  n |= (in[2] < 0) ? (1u << 31) : 0u;
  // The reference tests the REMAPPED x, y (and raw z) for non-zero.
  n |= (nx != 0.0f || ny != 0.0f || in[2] != 0.0f) ? (1u << 30) : 0u;
  return n;
}

/// ohm/VoxelIncidentCompute.h:82-112
__device__ inline uint32_t updateIncidentNormal(uint32_t packed, const float ray_in[3], uint32_t point_count)
{
  float normal[3];
  decodeNormal(packed, normal);
  point_count = ((normal[0] != 0 || normal[1] != 0 || normal[2] != 0) && point_count) ? point_count : 0;
  const float one_on_count_plus_one = 1.0f / float(point_count + 1);
  float len2 = ray_in[0] * ray_in[0] + ray_in[1] * ray_in[1] + ray_in[2] * ray_in[2];
  float s = (len2 > 1e-6f) ? 1.0f / sqrtf(len2) : 0.0f;
  const float ray[3] = { ray_in[0] * s, ray_in[1] * s, ray_in[2] * s };
  normal[0] += (ray[0] - normal[0]) * one_on_count_plus_one;
  normal[1] += (ray[1] - normal[1]) * one_on_count_plus_one;
  normal[2] += (ray[2] - normal[2]) * one_on_count_plus_one;
  len2 = normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2];
  s = (len2 > 1e-6f) ? 1.0f / sqrtf(len2) : 0.0f;
  normal[0] *= s;
  normal[1] *= s;
  normal[2] *= s;
  return encodeNormal(normal);
}

/// ohm/VoxelTouchTimeCompute.h:24-27
__device__ inline uint32_t encodeVoxelTouchTime(double timebase, double timestamp)
{
  return uint32_t((timestamp - timebase) / 0.001);
}

/// Range at which ray `r`'s LAST reported voxel is exited, as the CPU mapper's `last_exit_range` holds it when the
/// ray's sample is applied (ohm/RayMapperOccupancy.cpp:186, 299-305).  The CPU variable persists across rays: a ray
/// which reports no voxel inherits the value of the nearest earlier ray of the same integrateRays() call, 0 initially.
__device__ inline double lastExitRange(const RayWalk *__restrict__ walks, uint32_t r)
{
  for (long long q = (long long)r; q >= 0; --q)
  {
    const RayWalk rw = walks[q];
    if (!(rw.flags & kRwValid) || !(rw.flags & kRwWalk))
    {
      continue;
    }
    const int manhattan = rw.total[0] + rw.total[1] + rw.total[2];
    const bool include_end = (rw.flags & kRwIncludeEnd) != 0;
    const int visited = manhattan - (((rw.flags & kRwExcludeStart) && manhattan > 0) ? 1 : 0) + (include_end ? 1 : 0);
    if (visited <= 0)
    {
      continue;
    }
    if (include_end)
    {
      return rw.length;
    }
    // Exit range of the voxel before the end voxel == time of the walk's final step.
    double t = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      if (rw.total[a] > 0)
      {
        const double ta = stepTime(rw.init[a], rw.delta[a], rw.total[a]);
        t = (ta > t) ? ta : t;
      }
    }
    return t;
  }
  return 0.0;
}
}  // namespace ohmhip

#endif  // OHMHIP_SECONDARY_DEVICE_H
