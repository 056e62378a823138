// OhmGpuMap.h -- C++14 host mirror of the reference's ray-integration interface over the ohmhip C ABI.
//
//   ohm::OccupancyMap (ohm/OccupancyMap.h:291)  parameters + MapChunk / VoxelBlock layout host storage
//   ohm::RayMapper    (ohm/RayMapper.h:22-65)
//   ohm::GpuMap       (ohmgpu/GpuMap.h:143-384)   ohm::GpuNdtMap (ohmgpu/GpuNdtMap.h:63-132)
//   ohm::GpuTsdfMap   (ohmgpu/GpuTsdfMap.h:37-94)
//
// Same names, argument meaning and error behaviour as the reference for this path: integrateRays() takes
// `element_count` POINTS (2 per ray) and returns the number integrated (0 on failure); construction throws
// gputil::ApiException when device memory cannot be allocated; syncVoxels() is the fence that copies modified regions
// back into MapChunk::voxel_blocks-equivalent host memory (index = x + y*dx + z*dx*dy, ohm/MapChunk.h:33-50).
// Only what the path needs is mirrored: this is not a re-implementation of ohm::OccupancyMap.
//
// glm: callers which have glm pass glm::dvec3 arrays (3 packed doubles); this header does not need glm.
#ifndef OHMHIP_OHMGPUMAP_H
#define OHMHIP_OHMGPUMAP_H

#include "gputil_hip.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace ohm
{
/// ohm/RayFlag.h:16-60
enum RayFlag : unsigned
{
  kRfDefault = 0,
  kRfEndPointAsFree = (1u << 0u),
  kRfStopOnFirstOccupied = (1u << 1u),
  kRfExcludeOrigin = (1u << 2u),
  kRfExcludeSample = (1u << 3u),
  kRfExcludeRay = (1u << 4u),
  kRfExcludeUnobserved = (1u << 5u),
  kRfExcludeFree = (1u << 6u),
  kRfExcludeOccupied = (1u << 7u),
  kRfReverseWalk = (1u << 8u)
};

/// ohm/NdtMode.h
enum class NdtMode
{
  kNone = 0,
  kOccupancy,
  kTraversability
};

/// Plain 3 x double point, layout compatible with glm::dvec3.
struct dvec3
{
  double x, y, z;
};

/// ohm/RayFilter.h:21-29
enum RayFilterFlag : unsigned
{
  kRffInvalid = (1u << 0u),
  kRffClippedStart = (1u << 1u),
  kRffClippedEnd = (1u << 2u)
};

/// ohm/RayFilter.h:45: called per ray before integration; may move the points, set RayFilterFlag bits, or reject (false).
using RayFilterFunction = std::function<bool(dvec3 *, dvec3 *, unsigned *)>;

/// The parts of ohm::Aabb (ohm/Aabb.h) the stock ray filters use.
class Aabb
{
public:
  enum : unsigned
  {
    kClippedStart = 1u,
    kClippedEnd = 2u
  };
  Aabb(const dvec3 &min_ext, const dvec3 &max_ext) : corners_{ min_ext, max_ext } {}
  /// ohm/Aabb.h:301-312 with epsilon 0: closed box.
  bool contains(const dvec3 &p) const
  {
    return !(corners_[1].x < p.x || corners_[1].y < p.y || corners_[1].z < p.z) &&
           !(corners_[0].x > p.x || corners_[0].y > p.y || corners_[0].z > p.z);
  }
  /// ohm/Aabb.h:386-446 (allow_clamp false).
  bool clipLine(dvec3 &start, dvec3 &end, unsigned *clip_flags = nullptr) const
  {
    const dvec3 origin = start;
    dvec3 dir{ end.x - start.x, end.y - start.y, end.z - start.z };
    if (clip_flags)
    {
      *clip_flags = 0;
    }
    const double d2 = (dir.x * dir.x + dir.y * dir.y) + dir.z * dir.z;
    if (d2 < 1e-9)
    {
      return false;
    }
    const double length = std::sqrt(d2);
    dir.x /= length;
    dir.y /= length;
    dir.z /= length;
    double t[2] = { 0, 0 };
    if (!rayIntersect(origin, dir, t))
    {
      return false;
    }
    int hits = 0;
    if (t[0] > 0 && t[0] < length)
    {
      start = dvec3{ origin.x + dir.x * t[0], origin.y + dir.y * t[0], origin.z + dir.z * t[0] };
      if (clip_flags)
      {
        *clip_flags |= kClippedStart;
      }
      ++hits;
    }
    if (t[1] > 0 && t[1] < length)
    {
      end = dvec3{ origin.x + dir.x * t[1], origin.y + dir.y * t[1], origin.z + dir.z * t[1] };
      if (clip_flags)
      {
        *clip_flags |= kClippedEnd;
      }
      ++hits;
    }
    return hits > 0;
  }

private:
  /// ohm/Aabb.h:330-383 (slab test).
  bool rayIntersect(const dvec3 &o, const dvec3 &d, double t[2]) const
  {
    const double oo[3] = { o.x, o.y, o.z };
    const double dd[3] = { d.x, d.y, d.z };
    const double lo[3] = { corners_[0].x, corners_[0].y, corners_[0].z };
    const double hi[3] = { corners_[1].x, corners_[1].y, corners_[1].z };
    bool miss = false;
    for (int a = 0; a < 3; ++a)
    {
      const double inv = 1.0 / dd[a];
      const bool neg = dd[a] < 0.0;
      const double tmin = ((neg ? hi[a] : lo[a]) - oo[a]) * inv;
      const double tmax = ((neg ? lo[a] : hi[a]) - oo[a]) * inv;
      if (a == 0)
      {
        t[0] = tmin;
        t[1] = tmax;
        continue;
      }
      miss = miss || (t[0] > tmax) || (tmin > t[1]);
      t[0] = (tmin > t[0] || std::isnan(t[0])) ? tmin : t[0];
      t[1] = (tmax < t[1] || std::isnan(t[1])) ? tmax : t[1];
    }
    return !miss;
  }
  dvec3 corners_[2];
};

/// ohm/RayFilter.cpp:12-34
inline bool goodRayFilter(dvec3 *start, dvec3 *end, unsigned *filter_flags, double max_range)
{
  const double v[6] = { start->x, start->y, start->z, end->x, end->y, end->z };
  bool good = true;
  for (double c : v)
  {
    good = good && std::isfinite(c);
  }
  const double rx = end->x - start->x, ry = end->y - start->y, rz = end->z - start->z;
  good = good && (max_range <= 0 || (rx * rx + ry * ry) + rz * rz <= max_range * max_range);
  if (!good)
  {
    *filter_flags |= kRffInvalid;
  }
  return good;
}

/// ohm/RayFilter.cpp:37-58
inline bool clipRayFilter(dvec3 *start, dvec3 *end, unsigned *filter_flags, double max_length)
{
  const double v[6] = { start->x, start->y, start->z, end->x, end->y, end->z };
  bool good = true;
  for (double c : v)
  {
    good = good && std::isfinite(c);
  }
  double rx = end->x - start->x, ry = end->y - start->y, rz = end->z - start->z;
  const double len2 = (rx * rx + ry * ry) + rz * rz;
  if (good && max_length > 0 && len2 > max_length * max_length)
  {
    const double len = std::sqrt(len2);
    rx /= len;
    ry /= len;
    rz /= len;
    *end = dvec3{ start->x + rx * max_length, start->y + ry * max_length, start->z + rz * max_length };
    *filter_flags |= kRffClippedEnd;
  }
  *filter_flags |= good ? 0u : unsigned(kRffInvalid);
  return good;
}

/// ohm/RayFilter.cpp:61-78
inline bool clipBounded(dvec3 *start, dvec3 *end, unsigned *filter_flags, const Aabb &clip_box)
{
  unsigned line_clip_flags = 0;
  if (clip_box.clipLine(*start, *end, &line_clip_flags))
  {
    if (!clip_box.contains(*start) && !clip_box.contains(*end))
    {
      return false;
    }
  }
  *filter_flags |= (line_clip_flags & Aabb::kClippedStart) ? unsigned(kRffClippedStart) : 0u;
  *filter_flags |= (line_clip_flags & Aabb::kClippedEnd) ? unsigned(kRffClippedEnd) : 0u;
  return true;
}

/// ohm/RayFilter.cpp:81-93
inline bool clipToBounds(dvec3 * /*start*/, dvec3 *end, unsigned *filter_flags, const Aabb &clip_box)
{
  *filter_flags |= clip_box.contains(*end) ? unsigned(kRffClippedEnd) : 0u;
  return true;
}

/// ohm/MapProbability.h:20-36 (float)
inline float probabilityToValue(float probability) { return std::log(probability / (1.0f - probability)); }
inline float valueToProbability(float value)
{
  return (value == -INFINITY) ? 0.0f : 1.0f - (1.0f / (1.0f + std::exp(value)));
}

/// One region's voxel memory: one contiguous block per layer (ohm/MapChunk.h:155, ohm/VoxelBlock.h:48).
struct MapChunk
{
  std::array<int16_t, 3> region{ { 0, 0, 0 } };
  std::vector<std::vector<uint8_t>> voxel_blocks;  ///< indexed by layer id (ohmhip_layer_id); empty when absent
};

/// Host-side map description and chunk store.
class OccupancyMap
{
public:
  explicit OccupancyMap(double resolution = 1.0, int region_dim_x = 32, int region_dim_y = 32, int region_dim_z = 32)
    : resolution_(resolution)
  {
    region_dim_[0] = region_dim_x > 0 ? region_dim_x : 32;  // ohm/OccupancyMap.h:24-26
    region_dim_[1] = region_dim_y > 0 ? region_dim_y : 32;
    region_dim_[2] = region_dim_z > 0 ? region_dim_z : 32;
    // ohm/OccupancyMap.cpp:205-213
    min_voxel_value_ = -2.0f;
    max_voxel_value_ = 3.511f;
    hit_value_ = probabilityToValue(0.9f);
    miss_value_ = probabilityToValue(0.45f);
    threshold_value_ = probabilityToValue(0.5f);
    layers_ = OHMHIP_LAYER_BIT(OHMHIP_LID_OCCUPANCY);
  }

  double resolution() const { return resolution_; }
  const int *regionVoxelDimensions() const { return region_dim_; }
  size_t regionVoxelVolume() const { return size_t(region_dim_[0]) * region_dim_[1] * region_dim_[2]; }
  void setOrigin(double x, double y, double z)
  {
    origin_[0] = x;
    origin_[1] = y;
    origin_[2] = z;
  }
  const double *origin() const { return origin_; }
  void setHitProbability(float p) { hit_value_ = probabilityToValue(p); }
  void setMissProbability(float p) { miss_value_ = probabilityToValue(p); }
  /// ohm/OccupancyMap.h:623: the log-odds adjustments set directly.
  void setHitValue(float value) { hit_value_ = value; }
  void setMissValue(float value) { miss_value_ = value; }
  void setOccupancyThresholdProbability(float p) { threshold_value_ = probabilityToValue(p); }
  float hitValue() const { return hit_value_; }
  float missValue() const { return miss_value_; }
  float missProbability() const { return valueToProbability(miss_value_); }
  float occupancyThresholdValue() const { return threshold_value_; }
  float minVoxelValue() const { return min_voxel_value_; }
  float maxVoxelValue() const { return max_voxel_value_; }
  void setMinVoxelValue(float v) { min_voxel_value_ = v; }
  void setMaxVoxelValue(float v) { max_voxel_value_ = v; }
  bool saturateAtMinValue() const { return saturate_min_; }
  bool saturateAtMaxValue() const { return saturate_max_; }
  void setSaturateAtMinValue(bool s) { saturate_min_ = s; }
  void setSaturateAtMaxValue(bool s) { saturate_max_ = s; }
  /// Ray filter: 0 none, 1 goodRayFilter(range), 2 clipRayFilter(range) (ohm/RayFilter.cpp:12-58).
  void setRayFilter(int mode, double range)
  {
    ray_filter_ = mode;
    ray_filter_range_ = range;
  }
  int rayFilterMode() const { return ray_filter_; }
  double rayFilterRange() const { return ray_filter_range_; }
  unsigned layers() const { return layers_; }
  void addLayer(int layer_id) { layers_ |= OHMHIP_LAYER_BIT(layer_id); }
  bool hasLayer(int layer_id) const { return (layers_ & OHMHIP_LAYER_BIT(layer_id)) != 0; }

  using ChunkMap = std::map<std::array<int16_t, 3>, MapChunk>;
  ChunkMap &chunks() { return chunks_; }
  const ChunkMap &chunks() const { return chunks_; }
  size_t regionCount() const { return chunks_.size(); }

  /// OccupancyMap::region(key, allow_create) restricted to what syncVoxels() needs.
  MapChunk &region(const std::array<int16_t, 3> &key)
  {
    MapChunk &chunk = chunks_[key];
    chunk.region = key;
    if (chunk.voxel_blocks.size() < size_t(OHMHIP_LID_COUNT))
    {
      chunk.voxel_blocks.resize(OHMHIP_LID_COUNT);
    }
    return chunk;
  }

private:
  double resolution_;
  int region_dim_[3];
  double origin_[3] = { 0, 0, 0 };
  float hit_value_, miss_value_, threshold_value_, min_voxel_value_, max_voxel_value_;
  bool saturate_min_ = false, saturate_max_ = false;
  int ray_filter_ = OHMHIP_FILTER_GOOD;  // ohm/OccupancyMap.cpp:215-218
  double ray_filter_range_ = 1e10;
  unsigned layers_;
  ChunkMap chunks_;
};

/// ohm/RayMapper.h:22-65
class RayMapper
{
public:
  virtual ~RayMapper() = default;
  virtual bool valid() const = 0;
  virtual size_t integrateRays(const dvec3 *rays, size_t element_count, const float *intensities,
                               const double *timestamps, unsigned ray_update_flags) = 0;
  size_t integrateRays(const dvec3 *rays, size_t element_count)
  {
    return integrateRays(rays, element_count, nullptr, nullptr, kRfDefault);
  }
};

/// ohm::GpuMap (ohmgpu/GpuMap.h:143-384)
class GpuMap : public RayMapper
{
public:
  /// @param expected_element_count points per call the per-batch device buffers are sized for up front (as the reference does, ohmgpu/GpuMap.cpp:429-470); larger batches still work, the buffers then grow on demand.
  /// @param gpu_mem_size device memory budget for the resident map, 0 => default.
  explicit GpuMap(OccupancyMap *map, bool borrowed_map = true, unsigned expected_element_count = 2048,
                  size_t gpu_mem_size = 0)
    : GpuMap(map, borrowed_map, expected_element_count, gpu_mem_size, OHMHIP_MODE_OCCUPANCY, nullptr)
  {}

  ~GpuMap() override
  {
    if (handle_)
    {
      ohmhip_map_destroy(handle_);
    }
    if (!borrowed_map_)
    {
      delete map_;
    }
  }
  GpuMap(const GpuMap &) = delete;
  GpuMap &operator=(const GpuMap &) = delete;

  bool gpuOk() const { return handle_ != nullptr; }
  bool valid() const override { return gpuOk(); }
  OccupancyMap &map() { return *map_; }
  const OccupancyMap &map() const { return *map_; }
  bool borrowedMap() const { return borrowed_map_; }
  float hitValue() const { return map_->hitValue(); }
  float missValue() const { return map_->missValue(); }
  /// Pass-throughs to the map for API compatibility (ohmgpu/GpuMap.h:234-244); the device takes the new values with the
  /// next batch.
  void setHitValue(float value) { map_->setHitValue(value); }
  void setMissValue(float value) { map_->setMissValue(value); }
  /// ohmgpu/GpuMap.h:271, 323: the reference can sort a batch's rays by region before upload.  Here every batch is
  /// binned per region on the device: stored only.
  void setGroupedRays(bool group) { grouped_rays_ = group; }
  bool groupedRays() const { return grouped_rays_; }
  /// ohmgpu/GpuMap.h:246-262.  Stored only: this backend bins rays per region, it does not need segmentation.
  void setRaySegmentLength(double length) { ray_segment_length_ = length; }
  double raySegmentLength() const { return ray_segment_length_; }

  /// ohmgpu/GpuMap.cpp:348-369.  With a filter set, integrateRays() runs it per ray on the host, as the reference does
  /// (GpuMap.cpp:736-746), and hands the surviving rays and their RayFilterFlag bits to the device; without one the
  /// map's built-in filter (OccupancyMap::setRayFilter(mode, range)) runs on the device.
  void setRayFilter(const RayFilterFunction &ray_filter) { ray_filter_ = ray_filter; }
  const RayFilterFunction &rayFilter() const { return ray_filter_; }
  const RayFilterFunction &effectiveRayFilter() const { return ray_filter_; }
  void clearRayFilter() { ray_filter_ = RayFilterFunction(); }

  using RayMapper::integrateRays;
  /// ohmgpu/GpuMap.cpp:416: returns points integrated, 0 on failure (gpuOk() false, device error).
  size_t integrateRays(const dvec3 *rays, size_t element_count, const float *intensities, const double *timestamps,
                       unsigned ray_update_flags) override
  {
    if (!gpuOk() || !rays || element_count < 2)
    {
      return 0;
    }
    if (!pushConfigIfChanged())
    {
      return 0;
    }
    size_t integrated = 0;
    if (ray_filter_)
    {
      std::vector<dvec3> kept;
      std::vector<unsigned char> kept_flags;
      std::vector<float> kept_intensities;
      std::vector<double> kept_timestamps;
      kept.reserve(element_count);
      for (size_t i = 0; i + 1 < element_count; i += 2)
      {
        dvec3 start = rays[i], end = rays[i + 1];
        unsigned filter_flags = 0;
        if (!ray_filter_(&start, &end, &filter_flags))
        {
          continue;
        }
        kept.push_back(start);
        kept.push_back(end);
        kept_flags.push_back(static_cast<unsigned char>(filter_flags));
        if (intensities)
        {
          kept_intensities.push_back(intensities[i >> 1]);
        }
        if (timestamps)
        {
          kept_timestamps.push_back(timestamps[i >> 1]);
        }
      }
      if (kept.empty())
      {
        return 0;
      }
      last_status_ = ohmhip_map_integrate_rays_filtered(
        handle_, reinterpret_cast<const double *>(kept.data()), kept.size(),
        intensities ? kept_intensities.data() : nullptr, timestamps ? kept_timestamps.data() : nullptr,
        ray_update_flags, kept_flags.data(), &integrated);
      return (last_status_ == OHMHIP_OK) ? integrated : 0;
    }
    last_status_ = ohmhip_map_integrate_rays(handle_, reinterpret_cast<const double *>(rays), element_count,
                                             intensities, timestamps, ray_update_flags, &integrated);
    last_partial_ = (last_status_ == OHMHIP_OK) ? 0 : integrated;
    return (last_status_ == OHMHIP_OK) ? integrated : 0;
  }
  /// glm-compatible overload: any 3-double point type.
  template <typename Vec3>
  size_t integrateRays(const Vec3 *rays, size_t element_count, const float *intensities = nullptr,
                       const double *timestamps = nullptr, unsigned ray_update_flags = kRfDefault)
  {
    static_assert(sizeof(Vec3) == 3 * sizeof(double), "rays must be packed double triples (glm::dvec3)");
    return integrateRays(reinterpret_cast<const dvec3 *>(rays), element_count, intensities, timestamps,
                         ray_update_flags);
  }
  /// Rays already resident in device memory (e.g. the output buffer of GpuTransformSamples::transform): the same
  /// integration without the host-to-device staging.  The buffer's content must be complete when the call is made (the
  /// transform is synchronous; wait for any other producer): the map reads it on streams of its own.
  size_t integrateRays(const gputil::Buffer &device_rays, size_t element_count, unsigned ray_update_flags = kRfDefault)
  {
    if (!gpuOk() || !device_rays.isValid() || element_count < 2)
    {
      return 0;
    }
    if (!pushConfigIfChanged())
    {
      return 0;
    }
    void *ptr = nullptr;
    size_t integrated = 0;
    last_status_ = ohmhip_buffer_ptr(device_rays.handle(), &ptr);
    if (last_status_ == OHMHIP_OK)
    {
      last_status_ = ohmhip_map_integrate_rays_device(handle_, static_cast<const double *>(ptr), element_count, nullptr,
                                                      nullptr, ray_update_flags, &integrated);
    }
    return (last_status_ == OHMHIP_OK) ? integrated : 0;
  }
  int lastStatus() const { return last_status_; }
  /// After a failed integrateRays: the leading elements of that call that WERE integrated (non-zero only when a batch over
  /// the residency limit was split and a later part still did not fit, include/ohmhip.h); do not present those again.
  size_t lastPartialCount() const { return last_partial_; }

  /// ohmgpu/GpuMap.cpp:308-324 -> GpuLayerCache::syncToMainMemory: fence, then copy regions modified on the device
  /// into the host chunks (every enabled layer).
  void syncVoxels() { syncLayers(nullptr); }

  /// ohmgpu/GpuMap.cpp:327-345: only the listed layers (OHMHIP_LID_* ids here).  The regions stay marked as modified
  /// -- there is one mark per region, not per layer -- so a later syncVoxels() still brings the other layers over.
  void syncVoxels(const std::vector<int> &layer_indices) { syncLayers(&layer_indices); }

private:
  void syncLayers(const std::vector<int> *only)
  {
    if (!gpuOk())
    {
      return;
    }
    size_t count = 0;
    OHMHIP_GPUAPICHECK(ohmhip_map_dirty_regions(handle_, nullptr, 0, &count));
    std::vector<int16_t> keys(3 * count);
    if (count)
    {
      OHMHIP_GPUAPICHECK(ohmhip_map_dirty_regions(handle_, keys.data(), count, &count));
    }
    const size_t voxels = map_->regionVoxelVolume();
    for (int layer = 0; layer < OHMHIP_LID_COUNT; ++layer)
    {
      if (!map_->hasLayer(layer) || count == 0)
      {
        continue;
      }
      if (only && std::find(only->begin(), only->end(), layer) == only->end())
      {
        continue;
      }
      std::vector<void *> dsts(count);
      for (size_t i = 0; i < count; ++i)
      {
        MapChunk &chunk = map_->region({ { keys[3 * i], keys[3 * i + 1], keys[3 * i + 2] } });
        chunk.voxel_blocks[layer].resize(voxels * ohmhip_layer_voxel_bytes(layer));
        dsts[i] = chunk.voxel_blocks[layer].data();
      }
      OHMHIP_GPUAPICHECK(ohmhip_map_read_regions(handle_, layer, keys.data(), count, dsts.data()));
    }
    if (!only)
    {
      OHMHIP_GPUAPICHECK(ohmhip_map_clear_dirty(handle_));
    }
    OHMHIP_GPUAPICHECK(ohmhip_map_sync(handle_));
  }

public:

  ohmhip_batch_stats lastBatchStats() const
  {
    ohmhip_batch_stats st{};
    ohmhip_map_last_stats(handle_, &st);
    return st;
  }
  /// Device batches launched so far (a call that only collects its rays, or is rejected, launches none): what a caller
  /// recycling device ray buffers by the "two batches in flight" rule counts (include/ohmhip.h).
  uint64_t batchesLaunched() const
  {
    uint64_t n = 0;
    ohmhip_map_batches_launched(handle_, &n);
    return n;
  }
  /// Regions above 32768 voxels only: rays cut because a tile coordinate left the key range although the reference
  /// addresses the region (include/ohmhip.h: ohmhip_map_rays_beyond_tiles).
  uint64_t raysBeyondTiles() const
  {
    uint64_t n = 0;
    ohmhip_map_rays_beyond_tiles(handle_, &n);
    return n;
  }
  /// OccupancyMap::setFirstRayTime / firstRayTime (ohm/OccupancyMap.h:342-351): the touch-time layer's time base; the
  /// ranks of a partitioned map share one.
  void setFirstRayTime(double time) { OHMHIP_GPUAPICHECK(ohmhip_map_set_first_ray_time(handle_, time)); }
  double firstRayTime() const
  {
    double t = -1.0;
    ohmhip_map_first_ray_time(handle_, &t);
    return t;
  }
  /// Start markers of the set-up and binning passes for lastBatchStats().ms_setup (a few microseconds per batch; off by
  /// default, see ohmhip_map_set_phase_timing).
  void setPhaseTiming(bool enable) { OHMHIP_GPUAPICHECK(ohmhip_map_set_phase_timing(handle_, enable ? 1 : 0)); }
  /// The MapRegionCache face of the reference's GpuCache (ohm/MapRegionCache.h; ohmgpu/GpuCache.h:80): what the core
  /// map and the tests call through gpuCache() -- flush / clear / remove.  There is no separate cache object here (the
  /// whole map is resident), so this is a view of the GpuMap.
  class GpuCache
  {
  public:
    explicit GpuCache(GpuMap *owner) : owner_(owner) {}
    /// GpuCache::flush: bring the host map up to date.
    void flush() { owner_->syncVoxels(); }
    /// GpuCache::clear: drop every resident region (modified or not).
    void clear() { OHMHIP_GPUAPICHECK(ohmhip_map_clear(owner_->handle_)); }
    /// MapRegionCache::remove(region_key)
    void remove(const std::array<int16_t, 3> &region_key) { owner_->removeRegions(region_key.data(), 1); }
    /// GpuCache::reinitialise (ohmgpu/GpuCache.h:103): the reference rebuilds its layer caches after the map's layout
    /// changed, dropping what they held; here the device map keeps its layout for life, so this is clear().
    void reinitialise() { clear(); }
    /// GpuCache::targetGpuAllocSize (ohmgpu/GpuCache.h:139): the byte budget of the device-side voxel storage
    /// (0: bounded by the device's free memory only).
    size_t targetGpuAllocSize() const
    {
      ohmhip_cache_stats st{};
      OHMHIP_GPUAPICHECK(ohmhip_map_cache_stats(owner_->handle_, &st, 0));
      return size_t(st.memory_limit);
    }
    /// GpuCache::layerCount (ohmgpu/GpuCache.h:143): voxel layers held on the device.
    unsigned layerCount() const
    {
      unsigned n = 0;
      for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
      {
        n += owner_->map().hasLayer(l) ? 1u : 0u;
      }
      return n;
    }

  private:
    GpuMap *owner_;
  };
  GpuCache *gpuCache() { return &cache_view_; }

  /// MapRegionCache::remove, as OccupancyMap::cullRegions calls it on the map's GPU cache
  /// (ohm/OccupancyMap.cpp:1202-1234): drop regions (packed int16 x, y, z triples) from the device map.
  size_t removeRegions(const int16_t *keys_xyz, size_t count)
  {
    size_t removed = 0;
    OHMHIP_GPUAPICHECK(ohmhip_map_remove_regions(handle_, keys_xyz, count, &removed));
    return removed;
  }

  /// GpuLayerCache::queryStats (ohmgpu/GpuLayerCache.h:334-339): hits / misses / full of the resident region pool
  /// (ohmhip_cache_stats); @p reset clears the counters as GpuLayerCache::resetStats() does.
  ohmhip_cache_stats queryStats(bool reset = false) const
  {
    ohmhip_cache_stats st{};
    OHMHIP_GPUAPICHECK(ohmhip_map_cache_stats(handle_, &st, reset ? 1 : 0));
    return st;
  }

  /// Bound the region pool in bytes (include/ohmhip.h "RESIDENCY LIMIT"; the reference's gpu_mem_size): a batch that
  /// needs more fails -- integrateRays() returns 0 -- and leaves the map as it was.  0 removes the bound.
  void setMemoryLimit(uint64_t bytes) { OHMHIP_GPUAPICHECK(ohmhip_map_set_memory_limit(handle_, bytes)); }

  /// With a memory limit: spill the least recently used regions to a host store instead of failing the batch
  /// (include/ohmhip.h "SPILL TO HOST"; the reference reuses its least recently used cache slot,
  /// ohmgpu/GpuLayerCache.cpp:530-584).  Stored regions stay part of the map and come back when touched.
  void setSpillToHost(bool enable = true)
  {
    OHMHIP_GPUAPICHECK(ohmhip_map_set_spill_to_host(handle_, enable ? 1 : 0));
  }

  /// Opt-in background write-back of the spill path (include/ohmhip.h "WRITE-BACK"; the reference overlaps the download
  /// of a reused cache slot with queued work, ohmgpu/GpuLayerCache.cpp:550-584).
  void setSpillWriteback(bool enable = true)
  {
    OHMHIP_GPUAPICHECK(ohmhip_map_set_spill_writeback(handle_, enable ? 1 : 0));
  }

  /// Not in the reference: run consecutive small integrateRays() batches as one device batch of at least @p min_rays
  /// rays (see ohmhip_map_set_batch_coalescing; on by default with 65536); 0 turns it off.
  void setBatchCoalescing(size_t min_rays) { OHMHIP_GPUAPICHECK(ohmhip_map_set_batch_coalescing(handle_, min_rays)); }
  /// Large host batches return once staged; the launch sequence runs on a thread of the map (see
  /// ohmhip_map_set_async_launch).  Off by default.
  void setAsyncLaunch(bool enable) { OHMHIP_GPUAPICHECK(ohmhip_map_set_async_launch(handle_, enable ? 1 : 0)); }

  /// Not in the reference (single device): owner-computes multi-GPU mode, see ohmhip_map_set_region_ownership.  The
  /// map integrates only what falls in the regions @p rank owns among @p world_size maps fed the same ray stream.
  void setRegionOwnership(unsigned world_size, unsigned rank, int block_shift = 0)
  {
    OHMHIP_GPUAPICHECK(ohmhip_map_set_region_ownership(handle_, world_size, rank, block_shift));
  }

  /// Not in the reference (single device): the partitioned map (include/ohmhip.h "Partitioned map") -- like
  /// setRegionOwnership(), with the blocks of 2^block_shift regions dealt by a table (x fastest) over
  /// [grid_origin, grid_origin + grid_dims) in block coordinates; an empty table selects the block hash.
  struct RegionPartition
  {
    unsigned world_size = 1;
    unsigned rank = 0;
    int block_shift = 0;
    int grid_origin[3] = { 0, 0, 0 };
    unsigned grid_dims[3] = { 0, 0, 0 };
    std::vector<uint8_t> owners;
  };
  void setRegionPartition(const RegionPartition &partition)
  {
    ohmhip_partition p;
    p.world_size = partition.world_size;
    p.rank = partition.rank;
    p.block_shift = partition.block_shift;
    for (int a = 0; a < 3; ++a)
    {
      p.grid_origin[a] = partition.grid_origin[a];
      p.grid_dims[a] = partition.grid_dims[a];
    }
    p.owners = partition.owners.empty() ? nullptr : partition.owners.data();
    OHMHIP_GPUAPICHECK(ohmhip_map_set_region_partition(handle_, &p));
    partition_world_ = (partition.world_size > 1) ? partition.world_size : 1u;
  }
  /// Route @p element_count / 2 rays resident in @p device_rays to the ranks owning a region they touch
  /// (ohmhip_map_route_rays): @p routed is resized to hold them per destination, blocks back to back in rank order,
  /// rays in order inside a block; @p counts[d] = rays addressed to rank d.  Returns the total number of routed rays.
  size_t routeRays(const gputil::Buffer &device_rays, size_t element_count, unsigned ray_update_flags,
                   gputil::Buffer &routed, std::vector<uint32_t> &counts)
  {
    counts.assign(partition_world_, 0u);
    void *src = nullptr;
    OHMHIP_GPUAPICHECK(ohmhip_buffer_ptr(device_rays.handle(), &src));
    size_t capacity = routed.isValid() ? routed.size() / (6 * sizeof(double)) : 0;
    for (;;)
    {
      void *dst = nullptr;
      if (routed.isValid())
      {
        OHMHIP_GPUAPICHECK(ohmhip_buffer_ptr(routed.handle(), &dst));
      }
      const int status = ohmhip_map_route_rays(handle_, static_cast<const double *>(src), element_count / 2,
                                               ray_update_flags, static_cast<double *>(dst), nullptr, capacity,
                                               counts.data(), nullptr);
      size_t total = 0;
      for (uint32_t c : counts)
      {
        total += c;
      }
      if (status == OHMHIP_OK)
      {
        return total;
      }
      if (status != OHMHIP_ERR_CAPACITY)
      {
        OHMHIP_GPUAPICHECK(status);
      }
      capacity = total + total / 4 + 64;
      if (routed.isValid())
      {
        routed.resize(capacity * 6 * sizeof(double));
      }
      else
      {
        routed.create(capacity * 6 * sizeof(double));
      }
      capacity = routed.size() / (6 * sizeof(double));
    }
  }

  ohmhip_map_t handle() const { return handle_; }

protected:
  using ConfigHook = void (*)(ohmhip_map_config &, void *);
  GpuMap(OccupancyMap *map, bool borrowed_map, unsigned expected_element_count, size_t gpu_mem_size, int mode,
         void *hook_data, ConfigHook hook = nullptr)
    : map_(map)
    , borrowed_map_(borrowed_map)
  {
    ohmhip_map_config &cfg = cfg_;
    std::memset(&cfg, 0, sizeof(cfg));
    ohmhip_map_config_default(&cfg);
    cfg.resolution = map->resolution();
    for (int a = 0; a < 3; ++a)
    {
      cfg.region_dim[a] = map->regionVoxelDimensions()[a];
      cfg.origin[a] = map->origin()[a];
    }
    cfg.mode = mode;
    fillMapValues(cfg);
    cfg.gpu_mem_size = gpu_mem_size;
    if (hook)
    {
      hook(cfg, hook_data);
    }
    cfg.layers = map->layers();
    // gputil::Exception on allocation failure from the ctor (ohmgpu/GpuMap.h:53-54,159-160).
    OHMHIP_GPUAPICHECK(ohmhip_map_create(&handle_, &cfg));
    if (expected_element_count > 2048u)
    {
      // the reference sizes its ray / key buffers for expected_element_count points here (ohmgpu/GpuMap.cpp:429-470)
      // (best effort: a reservation the device cannot hold is not an error, the batches grow their buffers on demand)
      (void)ohmhip_map_reserve_rays(handle_, expected_element_count / 2u);
    }
    uploadExisting();
  }

  /// The host map's probabilities, clamps and built-in filter.
  void fillMapValues(ohmhip_map_config &cfg) const
  {
    cfg.hit_value = map_->hitValue();
    cfg.miss_value = map_->missValue();
    cfg.threshold_value = map_->occupancyThresholdValue();
    cfg.min_value = map_->minVoxelValue();
    cfg.max_value = map_->maxVoxelValue();
    cfg.saturate_at_min = map_->saturateAtMinValue();
    cfg.saturate_at_max = map_->saturateAtMaxValue();
    cfg.ray_filter = map_->rayFilterMode();
    cfg.ray_filter_range = map_->rayFilterRange();
  }
  /// Mapper-specific parameters (NDT, TSDF) as they stand now.
  virtual void fillMapperValues(ohmhip_map_config &) const {}

  /// The reference reads the map's parameters at every launch (ohmgpu/GpuMap.cpp:1036-1191): setters called on the
  /// OccupancyMap / mapper after construction apply from the next batch.
  /// @return false when the device refuses the new values (lastStatus() says why); nothing is integrated then.
  bool pushConfigIfChanged()
  {
    ohmhip_map_config now = cfg_;
    fillMapValues(now);
    fillMapperValues(now);
    if (std::memcmp(&now, &cfg_, sizeof(now)) != 0)
    {
      last_status_ = ohmhip_map_update_config(handle_, &now);
      if (last_status_ != OHMHIP_OK)
      {
        return false;
      }
      cfg_ = now;
    }
    return true;
  }

  /// gpumap::enableGpu + GpuLayerCache::upload for chunks the CPU map already holds.
  void uploadExisting() { uploadRegions(nullptr, 0); }

public:
  /// Push host chunks to the device: all of them (keys_xyz == nullptr) or the listed regions -- what
  /// GpuLayerCache::upload does for a region whose CPU copy is newer than the device's
  /// (ohmgpu/GpuLayerCache.cpp:172-182), e.g. after CPU-side integration into the host map.
  void uploadRegions(const int16_t *keys_xyz, size_t count)
  {
    for (int layer = 0; layer < OHMHIP_LID_COUNT; ++layer)
    {
      if (!map_->hasLayer(layer))
      {
        continue;
      }
      std::vector<const void *> srcs;
      std::vector<int16_t> layer_keys;
      auto take = [&](const std::array<int16_t, 3> &key, const MapChunk &chunk) {
        if (chunk.voxel_blocks.size() > size_t(layer) && !chunk.voxel_blocks[layer].empty())
        {
          srcs.push_back(chunk.voxel_blocks[layer].data());
          layer_keys.insert(layer_keys.end(), key.begin(), key.end());
        }
      };
      if (!keys_xyz)
      {
        for (const auto &entry : map_->chunks())
        {
          take(entry.first, entry.second);
        }
      }
      else
      {
        for (size_t i = 0; i < count; ++i)
        {
          const std::array<int16_t, 3> key{ { keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2] } };
          const auto found = map_->chunks().find(key);
          if (found != map_->chunks().end())
          {
            take(found->first, found->second);
          }
        }
      }
      if (!srcs.empty())
      {
        OHMHIP_GPUAPICHECK(ohmhip_map_write_regions(handle_, layer, layer_keys.data(), srcs.size(), srcs.data()));
      }
    }
  }

protected:
  OccupancyMap *map_ = nullptr;
  bool borrowed_map_ = true;
  ohmhip_map_t handle_ = nullptr;
  double ray_segment_length_ = 0;
  bool grouped_rays_ = false;
  int last_status_ = OHMHIP_OK;
  size_t last_partial_ = 0;
  unsigned partition_world_ = 1;
  RayFilterFunction ray_filter_;
  ohmhip_map_config cfg_;
  GpuCache cache_view_{ this };
};

/// ohm::GpuNdtMap (ohmgpu/GpuNdtMap.h:63-132).  NDT parameters default as in ohm/private/NdtMapDetail.h:20-45 and may
/// be changed BEFORE construction through NdtParams.
struct NdtParams
{
  float sensor_noise = 0.05f;
  unsigned sample_threshold = 3;
  float adaptation_rate = -1.0f;  ///< <= 0: derived from the map's miss probability (ohm/NdtMap.cpp:31-36)
  float reinitialise_covariance_threshold = probabilityToValue(0.2f);
  unsigned reinitialise_covariance_point_count = 100;
  float initial_intensity_covariance = 1.0f;
};

class GpuNdtMap : public GpuMap
{
public:
  GpuNdtMap(OccupancyMap *map, bool borrowed_map = true, unsigned expected_element_count = 2048,
            size_t gpu_mem_size = 0, NdtMode ndt_mode = NdtMode::kOccupancy, const NdtParams &params = NdtParams())
    : GpuMap(prepare(map, ndt_mode), borrowed_map, expected_element_count, gpu_mem_size,
             ndt_mode == NdtMode::kTraversability ? OHMHIP_MODE_NDT_TM : OHMHIP_MODE_NDT_OM,
             const_cast<NdtParams *>(&params), &GpuNdtMap::fill)
    , params_(params)
    , mode_(ndt_mode)
  {}
  NdtMode mode() const { return mode_; }
  float sensorNoise() const { return params_.sensor_noise; }
  /// ohmgpu/GpuNdtMap.h (setSensorNoise) and the NdtMap setters reached through ndtMap() in the reference
  /// (ohm/NdtMap.h:100-160): apply from the next batch.
  void setSensorNoise(float noise) { params_.sensor_noise = noise; }
  void setSampleThreshold(unsigned count) { params_.sample_threshold = count; }
  void setAdaptationRate(float rate) { params_.adaptation_rate = rate; }
  void setReinitialiseCovarianceThreshold(float value) { params_.reinitialise_covariance_threshold = value; }
  void setReinitialiseCovariancePointCount(unsigned count) { params_.reinitialise_covariance_point_count = count; }
  const NdtParams &ndtParams() const { return params_; }

protected:
  void fillMapperValues(ohmhip_map_config &cfg) const override { fill(cfg, const_cast<NdtParams *>(&params_)); }

public:

  /// ohm/NdtMap.h:146-149
  static float ndtAdaptationRateFromMissProbability(float miss_probability, float scale = 2.0f)
  {
    return std::max(0.0f, std::min(scale * (1.0f - 2.0f * miss_probability), 1.0f));
  }

private:
  static OccupancyMap *prepare(OccupancyMap *map, NdtMode mode)
  {
    // NdtMap::enableNdt (ohm/NdtMap.cpp:194-213): voxel mean + covariance (+ intensity, hit/miss for NDT-TM).
    map->addLayer(OHMHIP_LID_OCCUPANCY);
    map->addLayer(OHMHIP_LID_MEAN);
    map->addLayer(OHMHIP_LID_COVARIANCE);
    if (mode == NdtMode::kTraversability)
    {
      map->addLayer(OHMHIP_LID_INTENSITY);
      map->addLayer(OHMHIP_LID_HIT_MISS);
    }
    return map;
  }
  static void fill(ohmhip_map_config &cfg, void *data)
  {
    const NdtParams &p = *static_cast<const NdtParams *>(data);
    cfg.ndt_sensor_noise = p.sensor_noise;
    cfg.ndt_sample_threshold = p.sample_threshold;
    // cfg.miss_value is already the map's; derive the rate exactly as NdtMap's ctor does.
    cfg.ndt_adaptation_rate = (p.adaptation_rate > 0) ?
                                p.adaptation_rate :
                                ndtAdaptationRateFromMissProbability(valueToProbability(cfg.miss_value));
    cfg.ndt_reinit_threshold = p.reinitialise_covariance_threshold;
    cfg.ndt_reinit_count = p.reinitialise_covariance_point_count;
    cfg.ndt_initial_intensity_cov = p.initial_intensity_covariance;
  }
  NdtParams params_;
  NdtMode mode_;
};

/// ohm/VoxelTsdf.h:27-37
struct TsdfOptions
{
  float max_weight = 1e4f;
  float default_truncation_distance = 0.1f;
  float dropoff_epsilon = 0.0f;
  float sparsity_compensation_factor = 1.0f;
};

/// ohm::GpuTsdfMap (ohmgpu/GpuTsdfMap.h:37-94)
class GpuTsdfMap : public GpuMap
{
public:
  GpuTsdfMap(OccupancyMap *map, bool borrowed_map = true, unsigned expected_element_count = 2048,
             size_t gpu_mem_size = 0, const TsdfOptions &options = TsdfOptions())
    : GpuMap(prepare(map), borrowed_map, expected_element_count, gpu_mem_size, OHMHIP_MODE_TSDF,
             const_cast<TsdfOptions *>(&options), &GpuTsdfMap::fill)
    , options_(options)
  {}
  const TsdfOptions &tsdfOptions() const { return options_; }
  float maxWeight() const { return options_.max_weight; }
  float defaultTruncationDistance() const { return options_.default_truncation_distance; }
  /// ohmgpu/GpuTsdfMap.h:37-94 setters: apply from the next batch.
  void setTsdfOptions(const TsdfOptions &options) { options_ = options; }
  void setMaxWeight(float max_weight) { options_.max_weight = max_weight; }
  void setDefaultTruncationDistance(float distance) { options_.default_truncation_distance = distance; }
  void setSparsityCompensationFactor(float factor) { options_.sparsity_compensation_factor = factor; }
  float sparsityCompensationFactor() const { return options_.sparsity_compensation_factor; }
  void setDropoffEpsilon(float dropoff_epsilon) { options_.dropoff_epsilon = dropoff_epsilon; }
  float dropoffEpsilon() const { return options_.dropoff_epsilon; }

protected:
  void fillMapperValues(ohmhip_map_config &cfg) const override { fill(cfg, const_cast<TsdfOptions *>(&options_)); }

public:

private:
  static OccupancyMap *prepare(OccupancyMap *map)
  {
    map->addLayer(OHMHIP_LID_TSDF);
    return map;
  }
  static void fill(ohmhip_map_config &cfg, void *data)
  {
    const TsdfOptions &o = *static_cast<const TsdfOptions *>(data);
    cfg.tsdf_max_weight = o.max_weight;
    cfg.tsdf_trunc = o.default_truncation_distance;
    cfg.tsdf_dropoff = o.dropoff_epsilon;
    cfg.tsdf_sparsity = o.sparsity_compensation_factor;
  }
  TsdfOptions options_;
};

/// Quaternion in (x, y, z, w) member order (glm::dquat exposes the same members; its memory order depends on the glm
/// version, ohmgpu/GpuTransformSamples.cpp:169-174).
struct dquat
{
  double x, y, z, w;
};

/// ohm::GpuTransformSamples (ohmgpu/GpuTransformSamples.h:30-83): sensor-frame samples + timestamped trajectory ->
/// world-frame ray pairs in a device buffer.  fp64 on the device (the reference kernel is fp32).
class GpuTransformSamples
{
public:
  explicit GpuTransformSamples(gputil::Device &) {}
  GpuTransformSamples() = default;

  /// @return 2 x the number of valid samples written to @p output_buffer (as the reference), 0 on failure.
  unsigned transform(const double *transform_times, const dvec3 *transform_translations, const dquat *transform_rotations,
                     unsigned transform_count, const double *sample_times, const dvec3 *local_samples,
                     unsigned point_count, gputil::Queue &gpu_queue, gputil::Buffer &output_buffer,
                     double max_range = INFINITY)
  {
    if (!output_buffer.isValid())
    {
      output_buffer.create(sizeof(double) * 6 * (point_count ? point_count : 1u));
    }
    uint32_t elements = 0;
    last_status_ = ohmhip_transform_samples(
      transform_times, reinterpret_cast<const double *>(transform_translations),
      reinterpret_cast<const double *>(transform_rotations), transform_count, sample_times,
      reinterpret_cast<const double *>(local_samples), point_count, max_range, gpu_queue.handle(),
      output_buffer.handle(), &elements);
    return (last_status_ == OHMHIP_OK) ? elements : 0u;
  }
  int lastStatus() const { return last_status_; }

private:
  int last_status_ = OHMHIP_OK;
};

/// Voxel key as the queries report it: region + local coordinates (ohm/Key.h; the device writes the reference's GpuKey
/// records, ohmgpu/GpuKey.h:37-46).
struct Key
{
  int16_t region[3];
  uint8_t local[3];
  bool operator==(const Key &o) const
  {
    return region[0] == o.region[0] && region[1] == o.region[1] && region[2] == o.region[2] && local[0] == o.local[0] &&
           local[1] == o.local[1] && local[2] == o.local[2];
  }
};

/// ohm::LineKeysQueryGpu (ohmgpu/LineKeysQueryGpu.h; the interface of ohm/LineKeysQuery.h:47-101 + ohm/Query.h:51-121):
/// the voxel keys along a set of query lines, walked on the device with the CPU walk's fp64 semantics
/// (ohmhip_map_line_keys).  The query needs the map's geometry only; it is given the GpuMap that holds the device
/// handle.  Results as in the reference: one result per ray, resultIndices() / resultCounts() index
/// intersectedVoxels().
class LineKeysQueryGpu
{
public:
  explicit LineKeysQueryGpu(GpuMap &gpu_map, unsigned query_flags = 0)
    : gpu_map_(&gpu_map)
    , query_flags_(query_flags)
  {}
  void setGpuMap(GpuMap *gpu_map) { gpu_map_ = gpu_map; }
  unsigned queryFlags() const { return query_flags_; }
  void setQueryFlags(unsigned flags) { query_flags_ = flags; }

  /// Ray start / end point pairs; @p point_count is twice the number of rays.
  void setRays(const dvec3 *rays, size_t point_count)
  {
    rays_.assign(rays, rays + (point_count & ~size_t(1)));
  }
  const dvec3 *rays() const { return rays_.data(); }
  size_t rayPointCount() const { return rays_.size(); }

  size_t numberOfResults() const { return result_counts_.size(); }
  const size_t *resultIndices() const { return result_indices_.data(); }
  const size_t *resultCounts() const { return result_counts_.data(); }
  const Key *intersectedVoxels() const { return intersected_voxels_.data(); }
  const double *ranges() const { return nullptr; }  // (not reported by this query, as in the reference)

  /// Synchronous query (ohm/Query.h:93).  @return true on success.
  bool execute()
  {
    reset(false);
    const size_t ray_count = rays_.size() / 2;
    if (!gpu_map_ || !gpu_map_->gpuOk())
    {
      return false;
    }
    if (ray_count == 0)
    {
      return true;
    }
    // Worst case keys per line, as the reference sizes its buffer (ohmgpu/LineKeysQueryGpu.cpp:112-119).
    const double res = gpu_map_->map().resolution();
    uint32_t max_keys = 1;
    for (size_t i = 0; i < ray_count; ++i)
    {
      const dvec3 &a = rays_[2 * i], &b = rays_[2 * i + 1];
      const double len = std::sqrt((b.x - a.x) * (b.x - a.x) + (b.y - a.y) * (b.y - a.y) + (b.z - a.z) * (b.z - a.z));
      max_keys = std::max<uint32_t>(max_keys, uint32_t(std::ceil(len / res * std::sqrt(3.0))) + 4u);
    }
    std::vector<unsigned char> records(ray_count * size_t(max_keys) * 10u);
    std::vector<uint32_t> counts(ray_count);
    if (ohmhip_map_line_keys(gpu_map_->handle(), reinterpret_cast<const double *>(rays_.data()), ray_count, max_keys,
                             records.data(), counts.data()) != OHMHIP_OK)
    {
      return false;
    }
    result_indices_.resize(ray_count);
    result_counts_.resize(ray_count);
    for (size_t i = 0; i < ray_count; ++i)
    {
      result_indices_[i] = intersected_voxels_.size();
      const size_t n = std::min<size_t>(counts[i], max_keys);
      result_counts_[i] = n;
      for (size_t j = 0; j < n; ++j)
      {
        const unsigned char *rec = records.data() + (i * size_t(max_keys) + j) * 10u;
        Key key;
        std::memcpy(key.region, rec, 6);
        std::memcpy(key.local, rec + 6, 3);
        intersected_voxels_.push_back(key);
      }
    }
    return true;
  }
  /// The device call is synchronous: the asynchronous forms complete at once (ohm/Query.h:103-121).
  bool executeAsync() { return execute(); }
  bool wait(unsigned /*timeout_ms*/ = ~0u) { return true; }
  void reset(bool hard_reset = true)
  {
    result_indices_.clear();
    result_counts_.clear();
    intersected_voxels_.clear();
    if (hard_reset)
    {
      // (hard reset releases memory, ohm/Query.h:114)
      std::vector<Key>().swap(intersected_voxels_);
    }
  }

private:
  GpuMap *gpu_map_ = nullptr;
  unsigned query_flags_ = 0;
  std::vector<dvec3> rays_;
  std::vector<size_t> result_indices_, result_counts_;
  std::vector<Key> intersected_voxels_;
};

/// Not in the reference (ohm is single device): the RCCL communicator the library owns (include/ohmhip.h, ohmhip_comm_*).
/// Rank 0 makes the id (uniqueId) and the host program carries its 128 bytes to the other ranks over whatever it has
/// (MPI, a socket, a file); every rank then constructs its communicator -- a collective call.
class RayCommunicator
{
public:
  static std::array<unsigned char, OHMHIP_COMM_ID_BYTES> uniqueId()
  {
    std::array<unsigned char, OHMHIP_COMM_ID_BYTES> id{};
    OHMHIP_GPUAPICHECK(ohmhip_comm_unique_id(id.data()));
    return id;
  }
  RayCommunicator(const std::array<unsigned char, OHMHIP_COMM_ID_BYTES> &id, int world_size, int rank)
    : world_(world_size)
    , rank_(rank)
  {
    OHMHIP_GPUAPICHECK(ohmhip_comm_init_rank(&handle_, id.data(), world_size, rank));
  }
  RayCommunicator(const RayCommunicator &) = delete;
  RayCommunicator &operator=(const RayCommunicator &) = delete;
  ~RayCommunicator()
  {
    if (handle_)
    {
      ohmhip_comm_destroy(handle_);
    }
  }
  ohmhip_comm_t handle() const { return handle_; }
  int worldSize() const { return world_; }
  int rank() const { return rank_; }

private:
  ohmhip_comm_t handle_ = nullptr;
  int world_ = 1;
  int rank_ = 0;
};

/// Not in the reference: exact multi-GPU integration into a region-partitioned map (include/ohmhip.h "Partitioned map";
/// the C++ face of ohm_amd.distributed.PartitionedIntegrator).  Every rank constructs one around its EMPTY GpuMap with the
/// same territory table and calls integrateRays() with its own rays, all ranks the same number of times: the rays are
/// routed on the device to the owners of the regions they cross, exchanged over RCCL (48 B per routed ray) and what
/// arrives is integrated in (source rank, ray) order.  The union of the ranks' maps is bit-identical to one map
/// integrating rank 0's batch, then rank 1's, ...  The batch a call launches stays in flight; three receive buffers are
/// used in turn because the map keeps at most two batches in flight (ohmhip_map_integrate_rays_device).
class PartitionedIntegrator
{
public:
  PartitionedIntegrator(GpuMap &gpu_map, const GpuMap::RegionPartition &partition, RayCommunicator &comm)
    : map_(gpu_map)
    , comm_(comm)
    , exchange_queue_(gputil::Queue::create())
  {
    map_.setRegionPartition(partition);
  }

  /// Collective.  @p device_rays: this rank's rays (origin, sample pairs; element_count points), complete when the call
  /// is made and free again when it returns.  Returns the points this rank integrated (its own and received rays that
  /// pass the ray filter), 0 on failure (lastStatus()).
  size_t integrateRays(const gputil::Buffer &device_rays, size_t element_count, unsigned ray_update_flags = kRfDefault)
  {
    const int world = comm_.worldSize();
    send_counts_.assign(size_t(world), 0u);
    recv_counts_.assign(size_t(world), 0u);
    if (element_count >= 2 && device_rays.isValid())
    {
      map_.routeRays(device_rays, element_count, ray_update_flags, routed_, send_counts_);
    }
    last_status_ = ohmhip_comm_exchange_counts(comm_.handle(), send_counts_.data(), recv_counts_.data(),
                                               exchange_queue_.handle());
    if (last_status_ != OHMHIP_OK)
    {
      return 0;
    }
    size_t n_in = 0;
    for (uint32_t c : recv_counts_)
    {
      n_in += c;
    }
    // The turn advances per LAUNCH, not per call: a step that launches nothing -- no rays received, or rays
    // the map only collects -- leaves the two batches launched last in flight, possibly both still reading their
    // buffers.  A buffer is free once two more batches were launched after the one that reads it.
    const uint64_t launched = map_.batchesLaunched();
    size_t turn = 0;
    while (turn < 2 && recv_batch_[turn] != 0 && recv_batch_[turn] + 2 > launched)
    {
      ++turn;  // (at most two buffers are held by batches in flight: the third is always free)
    }
    recv_batch_[turn] = 0;
    gputil::Buffer &in = recv_[turn];
    const size_t in_bytes = std::max<size_t>(n_in, 1) * 6 * sizeof(double);
    if (!in.isValid())
    {
      in.create(in_bytes + in_bytes / 4);
    }
    else if (in.size() < in_bytes)
    {
      in.resize(in_bytes + in_bytes / 4);  // (the batch that read it has ended: see above)
    }
    void *d_routed = nullptr, *d_in = nullptr;
    if (routed_.isValid())
    {
      OHMHIP_GPUAPICHECK(ohmhip_buffer_ptr(routed_.handle(), &d_routed));
    }
    OHMHIP_GPUAPICHECK(ohmhip_buffer_ptr(in.handle(), &d_in));
    last_status_ = ohmhip_comm_exchange_rays(comm_.handle(), static_cast<const double *>(d_routed), send_counts_.data(),
                                             static_cast<double *>(d_in), recv_counts_.data(), exchange_queue_.handle());
    if (last_status_ != OHMHIP_OK)
    {
      return 0;
    }
    exchange_queue_.finish();  // the exchange only: the batches in flight stay in flight
    rays_received_ = n_in;
    if (n_in == 0)
    {
      return 0;
    }
    const size_t done = map_.integrateRays(in, 2 * n_in, ray_update_flags);
    last_status_ = map_.lastStatus();
    const uint64_t after = map_.batchesLaunched();
    recv_batch_[turn] = after > launched ? after : 0;  // (no launch: the rays were copied, the buffer is free)
    return done;
  }

  size_t raysReceived() const { return rays_received_; }
  const std::vector<uint32_t> &sendCounts() const { return send_counts_; }
  const std::vector<uint32_t> &receiveCounts() const { return recv_counts_; }
  int lastStatus() const { return last_status_; }

private:
  GpuMap &map_;
  RayCommunicator &comm_;
  gputil::Queue exchange_queue_;
  gputil::Buffer routed_, recv_[3];
  std::vector<uint32_t> send_counts_, recv_counts_;
  uint64_t recv_batch_[3] = { 0, 0, 0 };  ///< launch count of the batch reading each receive buffer (0: none)
  size_t rays_received_ = 0;
  int last_status_ = OHMHIP_OK;
};

/// ohm::configureGpu / gpuDevice (ohmgpu/OhmGpu.h:40-66): select the process-wide device.
inline int configureGpu(int device_index = 0)
{
  int count = 0;
  if (ohmhip_device_count(&count) != OHMHIP_OK || device_index >= count)
  {
    return 1;
  }
  return ohmhip_device_select(device_index) == OHMHIP_OK ? 0 : 1;
}
}  // namespace ohm

#endif  // OHMHIP_OHMGPUMAP_H
