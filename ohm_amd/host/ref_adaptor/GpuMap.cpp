// GpuMap.cpp -- ohm::GpuMap and the gpumap:: free functions (declared in the reference's ohmgpu/GpuMap.h) over
// libohmhip.so.  Replaces ohmgpu/GpuMap.cpp:106-1224 of the reference: no gputil buffers, no GpuLayerCache uploads, no
// kernel launches here -- one call into the C ABI per batch (include/ohmhip.h), which keeps the whole map resident.
//
// Behaviour kept from the reference class: integrateRays() is asynchronous and returns the number of points accepted
// (0 on failure); the effective ray filter (GpuMap's own or the map's) runs per ray on the host before upload
// (GpuMap.cpp:736-746 there); syncVoxels() brings the host MapChunk blocks up to date with the stamp protocol of
// GpuLayerCache::syncToMainMemory; map / mapper parameters are read again at every batch.
// Behaviour that differs, on purpose: results follow the CPU mappers bit for bit -- the line walk runs in
// fp64 from the origin, nothing is dropped under contention -- so setRaySegmentLength() and kRfReverseWalk, which exist
// in the reference only to reduce GPU contention and change results, are accepted and have no effect.
#include <ohmgpu/GpuMap.h>

#include "private/HipMapBinding.h"

#include <ohmgpu/GpuCache.h>

#include <ohm/Aabb.h>
#include <ohm/OccupancyMap.h>
#include <ohm/RayFilter.h>
#include <ohm/private/OccupancyMapDetail.h>

#include <logutil/Logger.h>

#include <algorithm>

namespace ohm
{
namespace gpumap
{
GpuCache *enableGpu(OccupancyMap &map)
{
  return enableGpu(map, GpuCache::kDefaultTargetMemSize, kGpuAllowMappedBuffers);
}

GpuCache *enableGpu(OccupancyMap &map, size_t target_gpu_mem_size, unsigned gpu_flags)
{
  OccupancyMapDetail &map_imp = *map.detail();
  if (!map_imp.gpu_cache)
  {
    map_imp.gpu_cache =
      new GpuCache(map, target_gpu_mem_size ? target_gpu_mem_size : size_t(GpuCache::kDefaultTargetMemSize), gpu_flags);
  }
  return static_cast<GpuCache *>(map_imp.gpu_cache);
}

void sync(OccupancyMap &map)
{
  if (HipMapBinding *binding = hipBinding(map))
  {
    binding->download({}, true);
  }
}

void sync(OccupancyMap &map, unsigned layer_index)
{
  // `layer_index` is a GpuCacheId in the reference (ohmgpu/GpuCache.h:32-44).
  HipMapBinding *binding = hipBinding(map);
  const int layer_id = ohmhip_adaptor::cacheIdToLayer(layer_index);
  if (binding && layer_id >= 0)
  {
    binding->download({ layer_id }, false);
  }
}

GpuCache *gpuCache(OccupancyMap &map)
{
  return static_cast<GpuCache *>(map.detail()->gpu_cache);
}

void walkRegions(const OccupancyMap &map, const glm::dvec3 &start_point, const glm::dvec3 &end_point,
                 const RegionWalkFunction &on_visit)
{
  // (The device enumerates regions itself -- k_ray_setup -- so nothing in this backend calls this; it is kept because it
  // is part of the header.  The merge of the three axes' boundary crossings is the core's.)
  struct Visit
  {
    const RegionWalkFunction &fn;
    const glm::dvec3 &start, &end;
  } visit{ on_visit, start_point, end_point };
  const glm::i16vec3 k0 = map.regionKey(start_point), k1 = map.regionKey(end_point);
  const glm::dvec3 extent = map.regionSpatialResolution(), centre = map.regionSpatialCentre(k0);
  const int16_t key0[3] = { k0.x, k0.y, k0.z }, key1[3] = { k1.x, k1.y, k1.z };
  ohmhip_adaptor::walkRegionKeys(
    &start_point.x, &end_point.x, key0, key1, &extent.x, &centre.x,
    [](const int16_t key[3], void *user) {
      const Visit &v = *static_cast<const Visit *>(user);
      v.fn(glm::i16vec3(key[0], key[1], key[2]), v.start, v.end);
    },
    &visit);
}
}  // namespace gpumap

GpuMap::GpuMap(GpuMapDetail *detail, unsigned expected_element_count, size_t gpu_mem_size)
  : imp_(detail)
{
  setMap(imp_->map, imp_->borrowed_map, expected_element_count, gpu_mem_size, true);
}

GpuMap::GpuMap(OccupancyMap *map, bool borrowed_map, unsigned expected_element_count, size_t gpu_mem_size)
  : GpuMap(new GpuMapDetail(map, borrowed_map, HipMapKind::kOccupancy), expected_element_count, gpu_mem_size)
{}

GpuMap::~GpuMap()
{
  if (imp_ && imp_->map)
  {
    // Outstanding device work must not outlive the object (the reference waits on its events here,
    // ohmgpu/GpuMap.cpp:290-305).
    if (HipMapBinding *binding = hipBinding(*imp_->map))
    {
      binding->core.sync();
    }
  }
  delete imp_;
}

bool GpuMap::gpuOk() const
{
  return imp_->gpu_ok;
}

OccupancyMap &GpuMap::map()
{
  return *imp_->map;
}

const OccupancyMap &GpuMap::map() const
{
  return *imp_->map;
}

bool GpuMap::borrowedMap() const
{
  return imp_->borrowed_map;
}

void GpuMap::syncVoxels()
{
  if (imp_->map)
  {
    gpumap::sync(*imp_->map);
    onSyncVoxels(0);
  }
}

void GpuMap::syncVoxels(const std::vector<int> &layer_indices)
{
  // `layer_indices` are HOST layer indices (ohmgpu/GpuMap.cpp:326-346): map them to device layer ids.
  HipMapBinding *binding = imp_->map ? hipBinding(*imp_->map) : nullptr;
  if (!binding)
  {
    return;
  }
  std::vector<int> ids;
  for (int id = 0; id < OHMHIP_LID_COUNT; ++id)
  {
    const int host_layer = binding->hostLayer(id);
    if (host_layer >= 0 && std::find(layer_indices.begin(), layer_indices.end(), host_layer) != layer_indices.end())
    {
      ids.push_back(id);
    }
  }
  if (!ids.empty())
  {
    binding->download(ids, false);
    onSyncVoxels(0);
  }
}

void GpuMap::setRayFilter(const RayFilterFunction &ray_filter)
{
  imp_->ray_filter = ray_filter;
  imp_->ray_filter_set = true;
}

const RayFilterFunction &GpuMap::rayFilter() const
{
  return imp_->ray_filter;
}

const RayFilterFunction &GpuMap::effectiveRayFilter() const
{
  return (imp_->ray_filter_set || !imp_->map) ? imp_->ray_filter : imp_->map->rayFilter();
}

void GpuMap::clearRayFilter()
{
  imp_->ray_filter = RayFilterFunction();
  imp_->ray_filter_set = true;  // an explicitly empty filter, not the map's (ohmgpu/GpuMap.cpp:364-369)
}

float GpuMap::hitValue() const
{
  return imp_->map ? imp_->map->hitValue() : 0.0f;
}

void GpuMap::setHitValue(float value)
{
  if (imp_->map)
  {
    imp_->map->setHitValue(value);
  }
}

float GpuMap::missValue() const
{
  return imp_->map ? imp_->map->missValue() : 0.0f;
}

void GpuMap::setMissValue(float value)
{
  if (imp_->map)
  {
    imp_->map->setMissValue(value);
  }
}

double GpuMap::raySegmentLength() const
{
  return imp_->ray_segment_length;
}

void GpuMap::setRaySegmentLength(double length)
{
  imp_->ray_segment_length = length;
}

bool GpuMap::groupedRays() const
{
  return imp_->grouped_rays;
}

size_t GpuMap::integrateRays(const glm::dvec3 *rays, size_t element_count, const float *intensities,
                             const double *timestamps, unsigned region_update_flags)
{
  return integrateRays(rays, element_count, intensities, timestamps, region_update_flags, effectiveRayFilter());
}

GpuCache *GpuMap::gpuCache() const
{
  return imp_->map ? gpumap::gpuCache(*imp_->map) : nullptr;
}

void GpuMap::setMap(OccupancyMap *map, bool borrowed_map, unsigned expected_element_count, size_t gpu_mem_size,
                    bool force_gpu_program_release)
{
  (void)expected_element_count;     // device buffers size themselves per batch
  (void)force_gpu_program_release;  // kernels live in libohmhip.so: nothing to build or release
  imp_->map = map;
  imp_->borrowed_map = borrowed_map;
  imp_->gpu_ok = false;
  if (!map)
  {
    return;
  }
  GpuCache *cache = gpumap::enableGpu(*map, gpu_mem_size, gpumap::kGpuAllowMappedBuffers);
  HipMapBinding *binding = cache ? hipBinding(*map) : nullptr;
  if (!binding)
  {
    return;
  }
  if (!binding->core.valid())
  {
    imp_->gpu_ok = binding->create(imp_->kind, imp_->ndt(), imp_->tsdf());
  }
  else
  {
    // A second mapper over a map that is on the device already must integrate the same way (one device map per host map).
    imp_->gpu_ok = binding->core.kind() == imp_->kind;
    if (!imp_->gpu_ok)
    {
      logutil::error("GpuMap: the map is already bound to a GPU mapper of another kind\n");
    }
  }
}

void GpuMap::setGroupedRays(bool group)
{
  imp_->grouped_rays = group;  // the device orders samples per voxel itself; kept for the accessor
}

void GpuMap::cacheGpuProgram(bool with_voxel_mean, bool with_traversal, bool force)
{
  (void)with_voxel_mean;
  (void)with_traversal;
  (void)force;
}

void GpuMap::releaseGpuProgram() {}

size_t GpuMap::integrateRays(const glm::dvec3 *rays, size_t element_count, const float *intensities,
                             const double *timestamps, unsigned region_update_flags, const RayFilterFunction &filter)
{
  HipMapBinding *binding = (imp_->map && imp_->gpu_ok && rays) ? hipBinding(*imp_->map) : nullptr;
  // A layout change since the last batch (layers added on the host) rebuilds the device map through
  // GpuCache::reinitialise(); parameters and CPU-side edits travel before the rays.
  if (!binding || !binding->pushConfig(imp_->ndt(), imp_->tsdf()) || !binding->uploadHostEdits())
  {
    return 0u;
  }
  imp_->map->touch();
  // The host filter pass (ohmgpu/GpuMap.cpp:736-746) is the core's; the RayFilterFunction rides behind a plain pointer.
  const ohmhip_adaptor::RayFilterC trampoline = [](double start[3], double end[3], unsigned *flags, void *user) {
    glm::dvec3 s(start[0], start[1], start[2]), e(end[0], end[1], end[2]);
    const bool keep = (*static_cast<const RayFilterFunction *>(user))(&s, &e, flags);
    for (int a = 0; a < 3; ++a)
    {
      start[a] = s[a];
      end[a] = e[a];
    }
    return keep;
  };
  const size_t done = binding->core.integrate(&rays[0].x, element_count, intensities, timestamps, region_update_flags,
                                              filter ? trampoline : nullptr,
                                              const_cast<RayFilterFunction *>(&filter));
  if (binding->core.lastStatus() != OHMHIP_OK)
  {
    logutil::error("GpuMap::integrateRays: ", ohmhip_error_string(binding->core.lastStatus()), "\n");
  }
  return done;
}

void GpuMap::waitOnPreviousOperation(int buffer_index)
{
  (void)buffer_index;
  if (HipMapBinding *binding = imp_->map ? hipBinding(*imp_->map) : nullptr)
  {
    binding->core.sync();
  }
}

// The three hooks below are the reference's internal batch pipeline (region upload, kernel launch).  The batch runs
// inside the library; they stay as no-ops for derived classes that call them.
void GpuMap::enqueueRegions(int buffer_index, unsigned region_update_flags)
{
  (void)buffer_index;
  (void)region_update_flags;
}

bool GpuMap::enqueueRegion(const glm::i16vec3 &region_key, int buffer_index)
{
  (void)region_key;
  (void)buffer_index;
  return true;
}

void GpuMap::finaliseBatch(unsigned region_update_flags)
{
  (void)region_update_flags;
}

int GpuMap::enableVoxelUpload(int cache_id, bool enable)
{
  (void)cache_id;
  (void)enable;
  return -1;
}
}  // namespace ohm
