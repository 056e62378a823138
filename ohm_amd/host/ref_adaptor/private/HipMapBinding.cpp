// HipMapBinding.cpp -- device map <-> ohm::OccupancyMap plumbing shared by the adaptor classes (see the header).
#include "HipMapBinding.h"

#include <ohm/DefaultLayer.h>
#include <ohm/MapChunk.h>
#include <ohm/MapLayer.h>
#include <ohm/MapLayout.h>
#include <ohm/NdtMap.h>
#include <ohm/OccupancyMap.h>
#include <ohm/VoxelBlock.h>
#include <ohm/VoxelBuffer.h>

#include <logutil/Logger.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace ohm
{
namespace
{
std::mutex g_registry_mutex;
std::unordered_map<const OccupancyMap *, HipMapBinding *> g_registry;

/// Host layer name of a device layer id (ohm/DefaultLayer.h:17-44).
const char *layerName(int layer_id)
{
  switch (layer_id)
  {
  case OHMHIP_LID_OCCUPANCY:
    return default_layer::occupancyLayerName();
  case OHMHIP_LID_MEAN:
    return default_layer::meanLayerName();
  case OHMHIP_LID_COVARIANCE:
    return default_layer::covarianceLayerName();
  case OHMHIP_LID_TRAVERSAL:
    return default_layer::traversalLayerName();
  case OHMHIP_LID_TOUCH_TIME:
    return default_layer::touchTimeLayerName();
  case OHMHIP_LID_INCIDENT:
    return default_layer::incidentNormalLayerName();
  case OHMHIP_LID_INTENSITY:
    return default_layer::intensityLayerName();
  case OHMHIP_LID_HIT_MISS:
    return default_layer::hitMissCountLayerName();
  case OHMHIP_LID_TSDF:
    return default_layer::tsdfLayerName();
  default:
    return nullptr;
  }
}

/// Device layers a map kind integrates into, given what the host layout offers.
unsigned deviceLayers(const OccupancyMap &map, HipMapKind kind)
{
  const MapLayout &layout = map.layout();
  auto has = [&layout](int id) { return layout.layerIndex(layerName(id)) >= 0; };
  unsigned bits = 0;
  if (kind == HipMapKind::kTsdf)
  {
    return has(OHMHIP_LID_TSDF) ? OHMHIP_LAYER_BIT(OHMHIP_LID_TSDF) : 0u;
  }
  const int ids[] = { OHMHIP_LID_OCCUPANCY, OHMHIP_LID_MEAN, OHMHIP_LID_TRAVERSAL, OHMHIP_LID_TOUCH_TIME,
                      OHMHIP_LID_INCIDENT };
  for (int id : ids)
  {
    bits |= has(id) ? OHMHIP_LAYER_BIT(id) : 0u;
  }
  if (kind != HipMapKind::kOccupancy)
  {
    bits |= has(OHMHIP_LID_COVARIANCE) ? OHMHIP_LAYER_BIT(OHMHIP_LID_COVARIANCE) : 0u;
  }
  if (kind == HipMapKind::kNdtTraversability)
  {
    bits |= has(OHMHIP_LID_INTENSITY) ? OHMHIP_LAYER_BIT(OHMHIP_LID_INTENSITY) : 0u;
    bits |= has(OHMHIP_LID_HIT_MISS) ? OHMHIP_LAYER_BIT(OHMHIP_LID_HIT_MISS) : 0u;
  }
  return bits;
}

bool sameValues(const ohmhip_map_config &a, const ohmhip_map_config &b)
{
  return std::memcmp(&a, &b, sizeof(a)) == 0;
}
}  // namespace

HipMapBinding *hipBinding(const OccupancyMap &map)
{
  std::lock_guard<std::mutex> guard(g_registry_mutex);
  const auto it = g_registry.find(&map);
  return (it != g_registry.end()) ? it->second : nullptr;
}

HipMapBinding *hipBinding(OccupancyMap &map)
{
  return hipBinding(static_cast<const OccupancyMap &>(map));
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

void fillConfig(ohmhip_map_config &cfg, const OccupancyMap &map, HipMapKind kind, const NdtMap *ndt,
                const TsdfOptions *tsdf)
{
  cfg.resolution = map.resolution();
  const glm::u8vec3 dim = map.regionVoxelDimensions();
  cfg.region_dim[0] = dim.x;
  cfg.region_dim[1] = dim.y;
  cfg.region_dim[2] = dim.z;
  const glm::dvec3 origin = map.origin();
  cfg.origin[0] = origin.x;
  cfg.origin[1] = origin.y;
  cfg.origin[2] = origin.z;
  cfg.hit_value = map.hitValue();
  cfg.miss_value = map.missValue();
  cfg.threshold_value = map.occupancyThresholdValue();
  cfg.min_value = map.minVoxelValue();
  cfg.max_value = map.maxVoxelValue();
  cfg.saturate_at_min = map.saturateAtMinValue() ? 1 : 0;
  cfg.saturate_at_max = map.saturateAtMaxValue() ? 1 : 0;
  // The map's own RayFilterFunction is host code: GpuMap runs it per ray (GpuMap.cpp in this directory) and hands the
  // survivors over with their flags, so the device applies no filter of its own.
  cfg.ray_filter = OHMHIP_FILTER_NONE;
  cfg.ray_filter_range = 0;
  switch (kind)
  {
  case HipMapKind::kOccupancy:
    cfg.mode = OHMHIP_MODE_OCCUPANCY;
    break;
  case HipMapKind::kNdtOccupancy:
    cfg.mode = OHMHIP_MODE_NDT_OM;
    break;
  case HipMapKind::kNdtTraversability:
    cfg.mode = OHMHIP_MODE_NDT_TM;
    break;
  case HipMapKind::kTsdf:
    cfg.mode = OHMHIP_MODE_TSDF;
    break;
  }
  if (ndt)
  {
    cfg.ndt_sensor_noise = ndt->sensorNoise();
    cfg.ndt_sample_threshold = const_cast<NdtMap *>(ndt)->ndtSampleThreshold();  // (not const in the reference)
    cfg.ndt_adaptation_rate = ndt->adaptationRate();
    cfg.ndt_reinit_threshold = ndt->reinitialiseCovarianceThreshold();
    cfg.ndt_reinit_count = ndt->reinitialiseCovariancePointCount();
    cfg.ndt_initial_intensity_cov = ndt->initialIntensityCovariance();
  }
  if (tsdf)
  {
    cfg.tsdf_max_weight = tsdf->max_weight;
    cfg.tsdf_trunc = tsdf->default_truncation_distance;
    cfg.tsdf_dropoff = tsdf->dropoff_epsilon;
    cfg.tsdf_sparsity = tsdf->sparsity_compensation_factor;
  }
}

HipMapBinding::~HipMapBinding()
{
  destroy();
}

void HipMapBinding::destroy()
{
  if (hip)
  {
    ohmhip_map_destroy(hip);
    hip = nullptr;
  }
}

int HipMapBinding::hostLayer(int layer_id) const
{
  const char *name = layerName(layer_id);
  return (map && name) ? map->layout().layerIndex(name) : -1;
}

bool HipMapBinding::create(HipMapKind new_kind, const NdtMap *ndt, const TsdfOptions *tsdf)
{
  destroy();
  kind = new_kind;
  ohmhip_map_config_default(&config);
  fillConfig(config, *map, kind, ndt, tsdf);
  config.layers = deviceLayers(*map, kind);
  config.gpu_mem_size = gpu_mem_size;
  last_status = ohmhip_map_create(&hip, &config);
  if (last_status != OHMHIP_OK)
  {
    logutil::error("ohmhip_map_create failed: ", ohmhip_error_string(last_status), "\n");
    hip = nullptr;
    return false;
  }
  // Everything the host map already holds goes up (a GpuMap over a CPU-built map, GpuMapTest PopulateMultiple etc.).
  synced_stamp = 0;
  return uploadHostEdits();
}

bool HipMapBinding::pushConfig(const NdtMap *ndt, const TsdfOptions *tsdf)
{
  ohmhip_map_config current = config;
  fillConfig(current, *map, kind, ndt, tsdf);
  if (sameValues(current, config))
  {
    return true;
  }
  last_status = ohmhip_map_update_config(hip, &current);
  if (last_status != OHMHIP_OK)
  {
    return false;
  }
  config = current;
  return true;
}

bool HipMapBinding::uploadHostEdits()
{
  if (!hip)
  {
    return false;
  }
  std::vector<std::pair<uint64_t, glm::i16vec3>> regions;
  map->collectDirtyRegions(synced_stamp, regions);
  if (regions.empty())
  {
    synced_stamp = map->stamp();
    return true;
  }
  std::vector<int16_t> keys;
  keys.reserve(regions.size() * 3);
  std::vector<MapChunk *> chunks;
  for (const auto &entry : regions)
  {
    MapChunk *chunk = map->region(entry.second, false);
    if (chunk)
    {
      keys.push_back(entry.second.x);
      keys.push_back(entry.second.y);
      keys.push_back(entry.second.z);
      chunks.push_back(chunk);
    }
  }
  for (int id = 0; id < OHMHIP_LID_COUNT; ++id)
  {
    const int host_layer = hostLayer(id);
    if (!(config.layers & OHMHIP_LAYER_BIT(id)) || host_layer < 0)
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
    last_status = ohmhip_map_write_regions(hip, id, keys.data(), chunks.size(), srcs.data());
    if (last_status != OHMHIP_OK)
    {
      logutil::error("ohmhip_map_write_regions failed: ", ohmhip_error_string(last_status), "\n");
      return false;
    }
  }
  synced_stamp = map->stamp();
  return true;
}

bool HipMapBinding::download(const std::vector<int> &layer_ids, bool clear_dirty)
{
  if (!hip)
  {
    return false;
  }
  size_t count = 0;
  last_status = ohmhip_map_dirty_regions(hip, nullptr, 0, &count);
  if (last_status != OHMHIP_OK)
  {
    return false;
  }
  std::vector<int16_t> keys(3 * std::max<size_t>(count, 1));
  last_status = ohmhip_map_dirty_regions(hip, keys.data(), count, &count);
  if (last_status != OHMHIP_OK)
  {
    return false;
  }
  std::vector<MapChunk *> chunks(count);
  for (size_t i = 0; i < count; ++i)
  {
    chunks[i] = map->region(glm::i16vec3(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]), true);
  }
  const glm::ivec3 dim(map->regionVoxelDimensions());
  const uint64_t stamp = count ? map->touch() : map->stamp();
  for (int id = 0; id < OHMHIP_LID_COUNT; ++id)
  {
    const int host_layer = hostLayer(id);
    if (!(config.layers & OHMHIP_LAYER_BIT(id)) || host_layer < 0 || count == 0)
    {
      continue;
    }
    if (!layer_ids.empty() && std::find(layer_ids.begin(), layer_ids.end(), id) == layer_ids.end())
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
    last_status = ohmhip_map_read_regions(hip, id, keys.data(), count, dsts.data());
    if (last_status != OHMHIP_OK)
    {
      logutil::error("ohmhip_map_read_regions failed: ", ohmhip_error_string(last_status), "\n");
      return false;
    }
    // Stamp protocol of GpuLayerCache::syncToMainMemory (ohmgpu/GpuLayerCache.cpp:685-694) and the post-sync handler
    // for the layers that carry "first valid" information (ohmgpu/private/GpuMapDetail.cpp:28-31).
    for (MapChunk *chunk : chunks)
    {
      chunk->dirty_stamp = stamp;
      chunk->touched_stamps[size_t(host_layer)].store(stamp, std::memory_order_relaxed);
      if (id == OHMHIP_LID_OCCUPANCY || id == OHMHIP_LID_TSDF)
      {
        chunk->invalidateFirstValidIndex();
        chunk->searchAndUpdateFirstValid(dim);
      }
    }
  }
  if (clear_dirty)
  {
    last_status = ohmhip_map_clear_dirty(hip);
  }
  // What we just wrote is not a CPU-side edit.
  synced_stamp = map->stamp();
  return last_status == OHMHIP_OK;
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
