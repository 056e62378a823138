// HipMapBinding.cpp -- see the header: ohm / glm types in, plain values and pointers out; the logic is HipBindingCore's.
#include "HipMapBinding.h"

#include <ohm/DefaultLayer.h>
#include <ohm/MapChunk.h>
#include <ohm/MapLayout.h>
#include <ohm/NdtMap.h>
#include <ohm/OccupancyMap.h>
#include <ohm/VoxelBlock.h>
#include <ohm/VoxelBuffer.h>

#include <mutex>
#include <unordered_map>

namespace ohm
{
namespace
{
std::mutex g_registry_mutex;
std::unordered_map<const OccupancyMap *, HipMapBinding *> g_registry;

ohmhip_adaptor::MapValues mapValues(const OccupancyMap &map)
{
  ohmhip_adaptor::MapValues v;
  v.resolution = map.resolution();
  const glm::u8vec3 dim = map.regionVoxelDimensions();
  const glm::dvec3 origin = map.origin();
  for (int a = 0; a < 3; ++a)
  {
    v.region_dim[a] = dim[a];
    v.origin[a] = origin[a];
  }
  v.hit_value = map.hitValue();
  v.miss_value = map.missValue();
  v.threshold_value = map.occupancyThresholdValue();
  v.min_value = map.minVoxelValue();
  v.max_value = map.maxVoxelValue();
  v.saturate_at_min = map.saturateAtMinValue();
  v.saturate_at_max = map.saturateAtMaxValue();
  return v;
}

ohmhip_adaptor::NdtValues ndtValues(const NdtMap *ndt)
{
  ohmhip_adaptor::NdtValues v;
  if (ndt)
  {
    v.present = true;
    v.sensor_noise = ndt->sensorNoise();
    v.sample_threshold = const_cast<NdtMap *>(ndt)->ndtSampleThreshold();  // (not const in the reference)
    v.adaptation_rate = ndt->adaptationRate();
    v.reinit_threshold = ndt->reinitialiseCovarianceThreshold();
    v.reinit_count = ndt->reinitialiseCovariancePointCount();
    v.initial_intensity_cov = ndt->initialIntensityCovariance();
  }
  return v;
}

ohmhip_adaptor::TsdfValues tsdfValues(const TsdfOptions *tsdf)
{
  ohmhip_adaptor::TsdfValues v;
  if (tsdf)
  {
    v.present = true;
    v.max_weight = tsdf->max_weight;
    v.default_truncation_distance = tsdf->default_truncation_distance;
    v.dropoff_epsilon = tsdf->dropoff_epsilon;
    v.sparsity_compensation_factor = tsdf->sparsity_compensation_factor;
  }
  return v;
}
}  // namespace

HipMapBinding *hipBinding(const OccupancyMap &map)
{
  std::lock_guard<std::mutex> guard(g_registry_mutex);
  const auto it = g_registry.find(&map);
  return (it != g_registry.end()) ? it->second : nullptr;
}

void registerHipBinding(OccupancyMap &map, HipMapBinding *binding)
{
  std::lock_guard<std::mutex> guard(g_registry_mutex);
  g_registry[&map] = binding;
}

void unregisterHipBinding(OccupancyMap &map)
{
  std::lock_guard<std::mutex> guard(g_registry_mutex);
  g_registry.erase(&map);
}

int HipMapBinding::hostLayer(int layer_id) const
{
  const char *name = ohmhip_adaptor::hostLayerName(layer_id);
  return (map && name) ? map->layout().layerIndex(name) : -1;
}

bool HipMapBinding::create(HipMapKind kind, const NdtMap *ndt, const TsdfOptions *tsdf)
{
  unsigned host_layers = 0;
  for (int id = 0; id < OHMHIP_LID_COUNT; ++id)
  {
    host_layers |= (hostLayer(id) >= 0) ? OHMHIP_LAYER_BIT(id) : 0u;
  }
  // Everything the host map already holds goes up (a GpuMap over a CPU-built map, GpuMapTest PopulateMultiple etc.).
  return core.create(kind, mapValues(*map), host_layers, gpu_mem_size, ndtValues(ndt), tsdfValues(tsdf)) &&
         uploadHostEdits();
}

bool HipMapBinding::pushConfig(const NdtMap *ndt, const TsdfOptions *tsdf)
{
  return core.pushConfig(mapValues(*map), ndtValues(ndt), tsdfValues(tsdf));
}

bool HipMapBinding::uploadHostEdits()
{
  if (!core.valid())
  {
    return false;
  }
  std::vector<std::pair<uint64_t, glm::i16vec3>> regions;
  map->collectDirtyRegions(core.syncedStamp(), regions);
  std::vector<int16_t> keys;
  std::vector<MapChunk *> chunks;
  for (const auto &entry : regions)
  {
    if (MapChunk *chunk = map->region(entry.second, false))
    {
      keys.insert(keys.end(), { entry.second.x, entry.second.y, entry.second.z });
      chunks.push_back(chunk);
    }
  }
  for (int id = 0; id < OHMHIP_LID_COUNT && !chunks.empty(); ++id)
  {
    const int host_layer = hostLayer(id);
    if (!(core.layers() & OHMHIP_LAYER_BIT(id)) || host_layer < 0)
    {
      continue;
    }
    // VoxelBuffer retains the block (uncompressed) while we read it (ohm/VoxelBuffer.h, ohm/VoxelBlock.cpp:123-150).
    std::vector<VoxelBuffer<const VoxelBlock>> buffers;
    std::vector<const void *> srcs;
    buffers.reserve(chunks.size());
    for (MapChunk *chunk : chunks)
    {
      buffers.emplace_back(chunk->voxel_blocks[size_t(host_layer)].get());
      srcs.push_back(buffers.back().voxelMemory());
    }
    if (!core.uploadBlocks(id, keys.data(), chunks.size(), srcs.data()))
    {
      return false;
    }
  }
  core.uploadsDone(map->stamp());
  return true;
}

bool HipMapBinding::download(const std::vector<int> &layer_ids, bool clear_dirty)
{
  std::vector<int16_t> keys;
  if (!core.dirtyRegions(keys))
  {
    return false;
  }
  const size_t count = keys.size() / 3;
  std::vector<MapChunk *> chunks(count);
  for (size_t i = 0; i < count; ++i)
  {
    chunks[i] = map->region(glm::i16vec3(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]), true);
  }
  const glm::ivec3 dim(map->regionVoxelDimensions());
  const uint64_t stamp = count ? map->touch() : map->stamp();
  for (int id = 0; id < OHMHIP_LID_COUNT && count; ++id)
  {
    const int host_layer = hostLayer(id);
    if (!core.downloadsLayer(id, layer_ids) || host_layer < 0)
    {
      continue;
    }
    std::vector<VoxelBuffer<VoxelBlock>> buffers;
    std::vector<void *> dsts;
    buffers.reserve(count);
    for (MapChunk *chunk : chunks)
    {
      buffers.emplace_back(chunk->voxel_blocks[size_t(host_layer)].get());
      dsts.push_back(buffers.back().voxelMemory());
    }
    if (!core.downloadBlocks(id, keys.data(), count, dsts.data()))
    {
      return false;
    }
    // Stamp protocol of GpuLayerCache::syncToMainMemory (ohmgpu/GpuLayerCache.cpp:685-694) and the post-sync handler
    // for the layers that carry "first valid" information (ohmgpu/private/GpuMapDetail.cpp:28-31).
    for (MapChunk *chunk : chunks)
    {
      chunk->dirty_stamp = stamp;
      chunk->touched_stamps[size_t(host_layer)].store(stamp, std::memory_order_relaxed);
      if (ohmhip_adaptor::layerCarriesFirstValid(id))
      {
        chunk->invalidateFirstValidIndex();
        chunk->searchAndUpdateFirstValid(dim);
      }
    }
  }
  return core.downloadsDone(clear_dirty, map->stamp());
}

GpuMapDetail::~GpuMapDetail()
{
  if (!borrowed_map)
  {
    delete map;
  }
}

GpuNdtMapDetail::GpuNdtMapDetail(OccupancyMap *map_in, bool borrowed, NdtMode mode)
  : GpuMapDetail(map_in, borrowed,
                 (mode == NdtMode::kTraversability) ? HipMapKind::kNdtTraversability : HipMapKind::kNdtOccupancy)
  , ndt_map(new NdtMap(map_in, true, mode))  // adds the mean / covariance (/ intensity / hit-miss) layers
{}

GpuNdtMapDetail::~GpuNdtMapDetail() = default;

GpuTsdfMapDetail::GpuTsdfMapDetail(OccupancyMap *map_in, bool borrowed)
  : GpuMapDetail(map_in, borrowed, HipMapKind::kTsdf)
{
  if (map_in && map_in->layout().layerIndex(default_layer::tsdfLayerName()) == -1)
  {
    MapLayout layout = map_in->layout();
    addTsdf(layout);
    map_in->updateLayout(layout);
  }
}
}  // namespace ohm
