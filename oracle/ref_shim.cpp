// ref_shim.cpp -- exports, through a C ABI, the functions of the reference's glm-free shared compute headers,
// compiled from the reference sources WHERE THEY LIE (-I/root/reference).  TEST INFRASTRUCTURE ONLY.
// No reference source is copied: this file only #includes it.  Used by tests/test_oracle_vs_ref.py to validate
// the C restatement in oracle/ohm_oracle.c.  The remaining headers on the path (LineWalkCompute.h,
// VoxelMeanCompute.h, CovarianceVoxelCompute.h, the RayMapper*.cpp files) require glm, which is not installed in
// this image, so they are NOT built (and no stand-in for glm is written).
#include <cmath>
#include <cstdint>
#include <algorithm>

#include <ohm/MapCoord.h>                // pointToRegionCoord, pointToRegionVoxel, regionCentreCoord
#include <ohm/VoxelOccupancyCompute.h>   // occupancyAdjustHit/Miss/Up/Down
#include <ohm/VoxelTouchTimeCompute.h>   // encodeVoxelTouchTime
#include <ohm/VoxelTsdfCompute.h>        // calculateTsdf<Vec3> (templated on the vector type)
#include <ohm/RayFlag.h>                 // RayFlag bit values
#include <ohmgpu/GpuKey.h>               // the device key record, host side (its "MapCoord.h" is ohm/'s)

#include <cstddef>
#include <cstring>

namespace
{
// The TSDF header is a template over Vec3; it needs operator- and dot() for whatever type it is given.
struct D3
{
  double x, y, z;
};
inline D3 operator-(const D3 &a, const D3 &b) { return D3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
// glm::dot(dvec3) evaluation order: (x*x' + y*y') + z*z'
inline double dot(const D3 &a, const D3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
}  // namespace

extern "C" {
int ref_point_to_region_coord(double coord, double resolution) { return ohm::pointToRegionCoord(coord, resolution); }
int ref_point_to_region_voxel(double coord, double voxel_resolution, double region_resolution)
{
  return ohm::pointToRegionVoxel(coord, voxel_resolution, region_resolution);
}
double ref_region_centre_coord(int region_coord, double region_dimension)
{
  return ohm::regionCentreCoord(region_coord, region_dimension);
}
void ref_occupancy_adjust_hit(float *occ, float initial, float adj, float uninit, float max_value, float sat_min,
                              float sat_max, int null_update)
{
  occupancyAdjustHit(occ, initial, adj, uninit, max_value, sat_min, sat_max, null_update != 0);
}
void ref_occupancy_adjust_miss(float *occ, float initial, float adj, float uninit, float min_value, float sat_min,
                               float sat_max, int null_update)
{
  occupancyAdjustMiss(occ, initial, adj, uninit, min_value, sat_min, sat_max, null_update != 0);
}
void ref_occupancy_adjust_up(float *occ, float initial, float adjusted, float uninit, float max_value, float sat_min,
                             float sat_max, int null_update)
{
  occupancyAdjustUp(occ, initial, adjusted, uninit, max_value, sat_min, sat_max, null_update != 0);
}
void ref_occupancy_adjust_down(float *occ, float initial, float adjusted, float uninit, float min_value, float sat_min,
                               float sat_max, int null_update)
{
  occupancyAdjustDown(occ, initial, adjusted, uninit, min_value, sat_min, sat_max, null_update != 0);
}
unsigned ref_encode_touch_time(double timebase, double timestamp) { return encodeVoxelTouchTime(timebase, timestamp); }
int ref_calculate_tsdf(const double sensor[3], const double sample[3], const double centre[3], float trunc,
                       float max_weight, float dropoff, float sparsity, float *weight, float *distance)
{
  return calculateTsdf<D3>(D3{ sensor[0], sensor[1], sensor[2] }, D3{ sample[0], sample[1], sample[2] },
                           D3{ centre[0], centre[1], centre[2] }, trunc, max_weight, dropoff, sparsity, weight,
                           distance) ?
           1 :
           0;
}
/// sizeof / alignof / member offsets of the reference's GpuKey (ohmgpu/GpuKey.h:37-46).
void ref_gpukey_layout(unsigned out[4])
{
  out[0] = unsigned(sizeof(ohm::GpuKey));
  out[1] = unsigned(alignof(ohm::GpuKey));
  out[2] = unsigned(offsetof(ohm::GpuKey, region));
  out[3] = unsigned(offsetof(ohm::GpuKey, voxel));
}
/// The bytes of a GpuKey holding the given members (the layout a replacement's key records must reproduce).
void ref_gpukey_bytes(const short region[3], const unsigned char voxel[4], unsigned char *out)
{
  ohm::GpuKey key;
  std::memset(&key, 0, sizeof(key));
  for (int i = 0; i < 3; ++i)
  {
    key.region[i] = region[i];
  }
  for (int i = 0; i < 4; ++i)
  {
    key.voxel[i] = voxel[i];
  }
  std::memcpy(out, &key, sizeof(key));
}
/// RayFlag values in declaration order (ohm/RayFlag.h:16-60).
void ref_ray_flags(unsigned out[12])
{
  const unsigned v[12] = { ohm::kRfDefault,         ohm::kRfEndPointAsFree, ohm::kRfStopOnFirstOccupied,
                           ohm::kRfExcludeOrigin,   ohm::kRfExcludeSample,  ohm::kRfExcludeRay,
                           ohm::kRfExcludeUnobserved, ohm::kRfExcludeFree,  ohm::kRfExcludeOccupied,
                           ohm::kRfReverseWalk,     ohm::kRfInternal,       ohm::kRfInternalTimestamps };
  for (int i = 0; i < 12; ++i)
  {
    out[i] = v[i];
  }
}
}
