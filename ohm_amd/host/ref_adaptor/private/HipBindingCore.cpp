// HipBindingCore.cpp -- see the header: the adaptor's logic on the C ABI alone (no ohm:: type, no glm).
#include "HipBindingCore.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace ohmhip_adaptor
{
const char *hostLayerName(int layer_id)
{
  // default_layer::occupancyLayerName() ... tsdfLayerName() (ohm/DefaultLayer.cpp:29-67): the names a map file stores.
  switch (layer_id)
  {
  case OHMHIP_LID_OCCUPANCY:
    return "occupancy";
  case OHMHIP_LID_MEAN:
    return "mean";
  case OHMHIP_LID_COVARIANCE:
    return "covariance";
  case OHMHIP_LID_TRAVERSAL:
    return "traversal";
  case OHMHIP_LID_TOUCH_TIME:
    return "touch_time";
  case OHMHIP_LID_INCIDENT:
    return "incident_normal";
  case OHMHIP_LID_INTENSITY:
    return "intensity";
  case OHMHIP_LID_HIT_MISS:
    return "hit_miss_count";
  case OHMHIP_LID_TSDF:
    return "tsdf";
  default:
    return nullptr;
  }
}

int cacheIdToLayer(unsigned cache_id)
{
  // enum GpuCacheId (ohmgpu/GpuCache.h:32-44): occupancy, clearance, voxel mean, covariance, intensity, hit-miss,
  // traversal, touch time, incident normal, tsdf.
  static const int kCacheToLayer[] = { OHMHIP_LID_OCCUPANCY, -1 /* clearance: not on this path */, OHMHIP_LID_MEAN,
                                       OHMHIP_LID_COVARIANCE, OHMHIP_LID_INTENSITY, OHMHIP_LID_HIT_MISS,
                                       OHMHIP_LID_TRAVERSAL, OHMHIP_LID_TOUCH_TIME, OHMHIP_LID_INCIDENT,
                                       OHMHIP_LID_TSDF };
  return (cache_id < sizeof(kCacheToLayer) / sizeof(kCacheToLayer[0])) ? kCacheToLayer[cache_id] : -1;
}

bool layerCarriesFirstValid(int layer_id)
{
  return layer_id == OHMHIP_LID_OCCUPANCY || layer_id == OHMHIP_LID_TSDF;
}

unsigned deviceLayers(unsigned host_layer_bits, MapKind kind)
{
  auto has = [host_layer_bits](int id) { return (host_layer_bits & OHMHIP_LAYER_BIT(id)) != 0; };
  if (kind == MapKind::kTsdf)
  {
    return has(OHMHIP_LID_TSDF) ? OHMHIP_LAYER_BIT(OHMHIP_LID_TSDF) : 0u;
  }
  unsigned bits = 0;
  const int ids[] = { OHMHIP_LID_OCCUPANCY, OHMHIP_LID_MEAN, OHMHIP_LID_TRAVERSAL, OHMHIP_LID_TOUCH_TIME,
                      OHMHIP_LID_INCIDENT };
  for (int id : ids)
  {
    bits |= has(id) ? OHMHIP_LAYER_BIT(id) : 0u;
  }
  if (kind != MapKind::kOccupancy)
  {
    bits |= has(OHMHIP_LID_COVARIANCE) ? OHMHIP_LAYER_BIT(OHMHIP_LID_COVARIANCE) : 0u;
  }
  if (kind == MapKind::kNdtTraversability)
  {
    bits |= has(OHMHIP_LID_INTENSITY) ? OHMHIP_LAYER_BIT(OHMHIP_LID_INTENSITY) : 0u;
    bits |= has(OHMHIP_LID_HIT_MISS) ? OHMHIP_LAYER_BIT(OHMHIP_LID_HIT_MISS) : 0u;
  }
  return bits;
}

void fillConfig(ohmhip_map_config &cfg, const MapValues &values, MapKind kind, const NdtValues &ndt,
                const TsdfValues &tsdf)
{
  cfg.resolution = values.resolution;
  for (int a = 0; a < 3; ++a)
  {
    cfg.region_dim[a] = values.region_dim[a];
    cfg.origin[a] = values.origin[a];
  }
  cfg.hit_value = values.hit_value;
  cfg.miss_value = values.miss_value;
  cfg.threshold_value = values.threshold_value;
  cfg.min_value = values.min_value;
  cfg.max_value = values.max_value;
  cfg.saturate_at_min = values.saturate_at_min ? 1 : 0;
  cfg.saturate_at_max = values.saturate_at_max ? 1 : 0;
  // The effective RayFilterFunction is host code: integrate() runs it per ray and hands the survivors over with their
  // flags, so the device applies no filter of its own.
  cfg.ray_filter = OHMHIP_FILTER_NONE;
  cfg.ray_filter_range = 0;
  switch (kind)
  {
  case MapKind::kOccupancy:
    cfg.mode = OHMHIP_MODE_OCCUPANCY;
    break;
  case MapKind::kNdtOccupancy:
    cfg.mode = OHMHIP_MODE_NDT_OM;
    break;
  case MapKind::kNdtTraversability:
    cfg.mode = OHMHIP_MODE_NDT_TM;
    break;
  case MapKind::kTsdf:
    cfg.mode = OHMHIP_MODE_TSDF;
    break;
  }
  if (ndt.present)
  {
    cfg.ndt_sensor_noise = ndt.sensor_noise;
    cfg.ndt_sample_threshold = ndt.sample_threshold;
    cfg.ndt_adaptation_rate = ndt.adaptation_rate;
    cfg.ndt_reinit_threshold = ndt.reinit_threshold;
    cfg.ndt_reinit_count = ndt.reinit_count;
    cfg.ndt_initial_intensity_cov = ndt.initial_intensity_cov;
  }
  if (tsdf.present)
  {
    cfg.tsdf_max_weight = tsdf.max_weight;
    cfg.tsdf_trunc = tsdf.default_truncation_distance;
    cfg.tsdf_dropoff = tsdf.dropoff_epsilon;
    cfg.tsdf_sparsity = tsdf.sparsity_compensation_factor;
  }
}

void walkRegionKeys(const double start[3], const double end[3], const int16_t start_key[3], const int16_t end_key[3],
                    const double region_extent[3], const double start_centre[3],
                    void (*visit)(const int16_t key[3], void *user), void *user)
{
  // The boundary crossings of the three axes merged by their parameter along the segment.
  int16_t key[3] = { start_key[0], start_key[1], start_key[2] };
  double next[3], pitch[3];  // parameter in [0, 1] of the next boundary crossing per axis, and its period
  int step[3];
  for (int a = 0; a < 3; ++a)
  {
    const double delta = end[a] - start[a];
    const int remaining = int(end_key[a]) - int(key[a]);
    step[a] = (remaining > 0) - (remaining < 0);
    if (step[a] == 0 || delta == 0)
    {
      next[a] = pitch[a] = std::numeric_limits<double>::infinity();
      step[a] = 0;
      continue;
    }
    const double face = start_centre[a] + 0.5 * double(step[a]) * region_extent[a];
    next[a] = (face - start[a]) / delta;
    pitch[a] = region_extent[a] / std::abs(delta);
  }
  visit(key, user);
  int guard = 3 * 65536;
  while ((key[0] != end_key[0] || key[1] != end_key[1] || key[2] != end_key[2]) && guard-- > 0)
  {
    int axis = 0;
    axis = (next[1] < next[axis]) ? 1 : axis;
    axis = (next[2] < next[axis]) ? 2 : axis;
    if (step[axis] == 0)
    {
      break;
    }
    key[axis] = int16_t(key[axis] + step[axis]);
    next[axis] = (key[axis] == end_key[axis]) ? std::numeric_limits<double>::infinity() : next[axis] + pitch[axis];
    visit(key, user);
  }
}

BindingCore::~BindingCore()
{
  destroy();
}

void BindingCore::destroy()
{
  if (hip_)
  {
    ohmhip_map_destroy(hip_);
    hip_ = nullptr;
  }
}

bool BindingCore::create(MapKind kind, const MapValues &values, unsigned host_layer_bits, size_t gpu_mem_size,
                         const NdtValues &ndt, const TsdfValues &tsdf)
{
  destroy();
  kind_ = kind;
  ohmhip_map_config_default(&config_);
  fillConfig(config_, values, kind, ndt, tsdf);
  config_.layers = deviceLayers(host_layer_bits, kind);
  config_.gpu_mem_size = gpu_mem_size;
  last_status_ = ohmhip_map_create(&hip_, &config_);
  if (last_status_ != OHMHIP_OK)
  {
    hip_ = nullptr;
    return false;
  }
  // Everything the host map already holds is newer than what the (empty) device map has.
  synced_stamp_ = 0;
  return true;
}

unsigned BindingCore::layerCount() const
{
  unsigned count = 0;
  for (int id = 0; id < OHMHIP_LID_COUNT; ++id)
  {
    count += (config_.layers & OHMHIP_LAYER_BIT(id)) ? 1u : 0u;
  }
  return count;
}

bool BindingCore::pushConfig(const MapValues &values, const NdtValues &ndt, const TsdfValues &tsdf)
{
  if (!hip_)
  {
    return false;
  }
  ohmhip_map_config current = config_;
  fillConfig(current, values, kind_, ndt, tsdf);
  if (std::memcmp(&current, &config_, sizeof(current)) == 0)
  {
    return true;
  }
  last_status_ = ohmhip_map_update_config(hip_, &current);
  if (last_status_ != OHMHIP_OK)
  {
    return false;
  }
  config_ = current;
  return true;
}

bool BindingCore::uploadBlocks(int layer_id, const int16_t *keys_xyz, size_t count, const void *const *blocks)
{
  if (!hip_ || !(config_.layers & OHMHIP_LAYER_BIT(layer_id)))
  {
    return false;
  }
  if (count == 0)
  {
    return true;
  }
  last_status_ = ohmhip_map_write_regions(hip_, layer_id, keys_xyz, count, blocks);
  return last_status_ == OHMHIP_OK;
}

bool BindingCore::dirtyRegions(std::vector<int16_t> &keys_xyz)
{
  keys_xyz.clear();
  if (!hip_)
  {
    return false;
  }
  size_t count = 0;
  last_status_ = ohmhip_map_dirty_regions(hip_, nullptr, 0, &count);
  if (last_status_ != OHMHIP_OK)
  {
    return false;
  }
  keys_xyz.resize(3 * std::max<size_t>(count, 1));
  last_status_ = ohmhip_map_dirty_regions(hip_, keys_xyz.data(), count, &count);
  keys_xyz.resize(3 * count);
  return last_status_ == OHMHIP_OK;
}

bool BindingCore::downloadsLayer(int layer_id, const std::vector<int> &only) const
{
  if (layer_id < 0 || layer_id >= OHMHIP_LID_COUNT || !(config_.layers & OHMHIP_LAYER_BIT(layer_id)))
  {
    return false;
  }
  return only.empty() || std::find(only.begin(), only.end(), layer_id) != only.end();
}

bool BindingCore::downloadBlocks(int layer_id, const int16_t *keys_xyz, size_t count, void *const *blocks)
{
  if (!hip_)
  {
    return false;
  }
  if (count == 0)
  {
    return true;
  }
  last_status_ = ohmhip_map_read_regions(hip_, layer_id, keys_xyz, count, blocks);
  return last_status_ == OHMHIP_OK;
}

bool BindingCore::downloadsDone(bool clear_dirty, uint64_t map_stamp)
{
  if (!hip_)
  {
    return false;
  }
  if (clear_dirty)
  {
    last_status_ = ohmhip_map_clear_dirty(hip_);
  }
  synced_stamp_ = map_stamp;
  return last_status_ == OHMHIP_OK;
}

size_t BindingCore::integrate(const double *rays, size_t element_count, const float *intensities,
                              const double *timestamps, unsigned region_update_flags, RayFilterC filter,
                              void *filter_user)
{
  if (!hip_ || !rays || element_count < 2)
  {
    return 0u;
  }
  const size_t ray_count = element_count / 2;
  size_t done = 0;
  if (!filter)
  {
    last_status_ = ohmhip_map_integrate_rays(hip_, rays, 2 * ray_count, intensities, timestamps, region_update_flags, &done);
    return (last_status_ == OHMHIP_OK) ? done : 0u;
  }
  // Host filter pass (ohmgpu/GpuMap.cpp:736-746): rejected rays are dropped, accepted ones go on with their possibly
  // moved end points and the RayFilterFlag bits the filter set.
  kept_rays_.clear();
  kept_intensities_.clear();
  kept_timestamps_.clear();
  kept_flags_.clear();
  for (size_t i = 0; i < ray_count; ++i)
  {
    double pair[6];
    std::memcpy(pair, rays + 6 * i, sizeof(pair));
    unsigned filter_flags = 0;
    if (!filter(pair, pair + 3, &filter_flags, filter_user))
    {
      continue;
    }
    kept_rays_.insert(kept_rays_.end(), pair, pair + 6);
    kept_flags_.push_back(static_cast<unsigned char>(filter_flags));
    if (intensities)
    {
      kept_intensities_.push_back(intensities[i]);
    }
    if (timestamps)
    {
      kept_timestamps_.push_back(timestamps[i]);
    }
  }
  const size_t kept = kept_flags_.size();
  if (kept == 0)
  {
    last_status_ = OHMHIP_OK;
    return 0u;
  }
  last_status_ = ohmhip_map_integrate_rays_filtered(hip_, kept_rays_.data(), 2 * kept,
                                                    intensities ? kept_intensities_.data() : nullptr,
                                                    timestamps ? kept_timestamps_.data() : nullptr, region_update_flags,
                                                    kept_flags_.data(), &done);
  return (last_status_ == OHMHIP_OK) ? done : 0u;
}

void BindingCore::sync()
{
  if (hip_)
  {
    last_status_ = ohmhip_map_sync(hip_);
  }
}

void BindingCore::clearResidency()
{
  if (hip_)
  {
    last_status_ = ohmhip_map_clear(hip_);
    synced_stamp_ = 0;
  }
}

void BindingCore::removeRegion(const int16_t key[3])
{
  if (hip_)
  {
    size_t removed = 0;
    last_status_ = ohmhip_map_remove_regions(hip_, key, 1, &removed);
  }
}

bool BindingCore::cacheStats(ohmhip_cache_stats &stats)
{
  if (!hip_)
  {
    return false;
  }
  last_status_ = ohmhip_map_cache_stats(hip_, &stats, 0);
  return last_status_ == OHMHIP_OK;
}
}  // namespace ohmhip_adaptor
