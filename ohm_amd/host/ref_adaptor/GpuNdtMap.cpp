// GpuNdtMap.cpp -- ohm::GpuNdtMap (declared in the reference's ohmgpu/GpuNdtMap.h) over libohmhip.so.  Replaces
// ohmgpu/GpuNdtMap.cpp:74-503: the two dependent kernel launches of the reference (misses for all rays, then hits per
// sample voxel) are one batch of the library's NDT mode, whose ordered replay follows RayMapperNdt ray by ray
// (ohm/RayMapperNdt.cpp:84-407).
#include <ohmgpu/GpuNdtMap.h>

#include "private/HipMapBinding.h"

#include <ohm/NdtMap.h>

namespace ohm
{
GpuNdtMap::GpuNdtMap(OccupancyMap *map, bool borrowed_map, unsigned expected_element_count, size_t gpu_mem_size,
                     NdtMode ndt_mode)
  : GpuMap(new GpuNdtMapDetail(map, borrowed_map, ndt_mode), expected_element_count, gpu_mem_size)
{
  setGroupedRays(true);
}

GpuNdtMap::~GpuNdtMap() = default;

void GpuNdtMap::setSensorNoise(float noise_range)
{
  detail()->ndt_map->setSensorNoise(noise_range);  // reaches the device with the next batch (HipMapBinding::pushConfig)
}

float GpuNdtMap::sensorNoise() const
{
  return detail()->ndt_map->sensorNoise();
}

NdtMap &GpuNdtMap::ndtMap()
{
  return *detail()->ndt_map;
}

const NdtMap &GpuNdtMap::ndtMap() const
{
  return *detail()->ndt_map;
}

GpuNdtMapDetail *GpuNdtMap::detail()
{
  return static_cast<GpuNdtMapDetail *>(imp_);
}

const GpuNdtMapDetail *GpuNdtMap::detail() const
{
  return static_cast<const GpuNdtMapDetail *>(imp_);
}

void GpuNdtMap::cacheGpuProgram(bool with_voxel_mean, bool with_traversal, bool force)
{
  (void)with_voxel_mean;
  (void)with_traversal;
  (void)force;
}

void GpuNdtMap::finaliseBatch(unsigned region_update_flags)
{
  (void)region_update_flags;
}

void GpuNdtMap::invokeNdt(unsigned region_update_flags, int buf_idx, gputil::EventList &wait,
                          TouchedCacheSet &used_caches)
{
  (void)region_update_flags;
  (void)buf_idx;
  (void)wait;
  (void)used_caches;
}

void GpuNdtMap::releaseGpuProgram() {}
}  // namespace ohm
