// HipMapBinding.h -- what the adaptor classes share: one libohmhip.so map per ohm::OccupancyMap.
//
// Level-2 adaptor: the reference's PUBLIC ohmgpu headers (ohmgpu/GpuMap.h, GpuNdtMap.h, GpuTsdfMap.h,
// GpuCache.h, OhmGpu.h) are used as they are, from the reference checkout; this directory supplies the member
// definitions on top of the C ABI in include/ohmhip.h instead of gputil + GpuLayerCache + the OpenCL/CUDA kernels.
// Needs the real reference tree and real glm: it cannot be compiled in the development image (no glm) and has never
// been built -- see README.md in this directory.
#ifndef OHMHIP_REF_ADAPTOR_HIPMAPBINDING_H
#define OHMHIP_REF_ADAPTOR_HIPMAPBINDING_H

#include <ohm/NdtMode.h>
#include <ohm/RayFilter.h>
#include <ohm/VoxelTsdf.h>

#include <ohmhip.h>

#include <glm/glm.hpp>

#include <cstdint>
#include <memory>
#include <vector>

namespace ohm
{
class OccupancyMap;
class NdtMap;

/// What kind of map object drives the device map (fixes ohmhip_map_config::mode).
enum class HipMapKind
{
  kOccupancy,
  kNdtOccupancy,
  kNdtTraversability,
  kTsdf
};

/// One device map bound to one host OccupancyMap.  Owned by the map's GpuCache (ohm/private/OccupancyMapDetail.h:
/// gpu_cache), shared by every GpuMap / GpuNdtMap / GpuTsdfMap constructed over that map -- like the reference's
/// GpuCache is (ohmgpu/private/GpuMapDetail.cpp:42-56).
struct HipMapBinding
{
  OccupancyMap *map = nullptr;
  ohmhip_map_t hip = nullptr;
  ohmhip_map_config config{};   ///< as last sent to the device
  HipMapKind kind = HipMapKind::kOccupancy;
  size_t gpu_mem_size = 0;
  /// Host map stamp up to which host and device agree: regions whose dirty_stamp is newer were edited on the CPU and
  /// are uploaded before the next batch (the GpuLayerCache::upload case, ohmgpu/GpuLayerCache.cpp:462-485).
  uint64_t synced_stamp = 0;
  int last_status = OHMHIP_OK;

  ~HipMapBinding();

  /// (Re)create the device map for the host map's current layout and upload every region the host holds.
  bool create(HipMapKind new_kind, const NdtMap *ndt, const TsdfOptions *tsdf);
  void destroy();
  /// Host layer index of a device layer id, or -1 when the host map has no such layer.
  int hostLayer(int layer_id) const;
  /// Send probabilities / clamps / NDT / TSDF parameters that changed since the last batch (the reference reads them at
  /// every launch, ohmgpu/GpuMap.cpp:1036-1191).
  bool pushConfig(const NdtMap *ndt, const TsdfOptions *tsdf);
  /// Upload regions edited on the CPU since the last sync.
  bool uploadHostEdits();
  /// GpuLayerCache::syncToMainMemory for the given device layer ids (empty: all enabled layers).
  bool download(const std::vector<int> &layer_ids, bool clear_dirty);
};

/// The binding of @p map (created by gpumap::enableGpu), or null.
HipMapBinding *hipBinding(OccupancyMap &map);
HipMapBinding *hipBinding(const OccupancyMap &map);
/// Registry maintenance (GpuCache ctor / dtor).
void registerHipBinding(OccupancyMap &map, HipMapBinding *binding);
void unregisterHipBinding(OccupancyMap &map);

/// Fill the value half of a configuration from the host map (and NDT / TSDF parameters when given).
void fillConfig(ohmhip_map_config &cfg, const OccupancyMap &map, HipMapKind kind, const NdtMap *ndt,
                const TsdfOptions *tsdf);

struct GpuMapDetail
{
  OccupancyMap *map = nullptr;
  bool borrowed_map = true;
  HipMapKind kind = HipMapKind::kOccupancy;
  RayFilterFunction ray_filter;      ///< GpuMap::setRayFilter (overrides the map's own filter)
  bool ray_filter_set = false;
  double ray_segment_length = 0;     ///< accepted and ignored: segmenting changes results vs the CPU mapper
  bool grouped_rays = false;
  bool gpu_ok = false;
  // scratch for host-filtered batches
  std::vector<double> kept_rays;
  std::vector<float> kept_intensities;
  std::vector<double> kept_timestamps;
  std::vector<unsigned char> kept_flags;

  GpuMapDetail(OccupancyMap *map_in, bool borrowed, HipMapKind kind_in)
    : map(map_in), borrowed_map(borrowed), kind(kind_in)
  {}
  virtual ~GpuMapDetail();
  /// NDT / TSDF parameter sources of the derived details (null in the base).
  virtual const NdtMap *ndt() const { return nullptr; }
  virtual const TsdfOptions *tsdf() const { return nullptr; }
};

struct GpuNdtMapDetail : public GpuMapDetail
{
  std::unique_ptr<NdtMap> ndt_map;
  GpuNdtMapDetail(OccupancyMap *map_in, bool borrowed, NdtMode mode);
  ~GpuNdtMapDetail() override;
  const NdtMap *ndt() const override { return ndt_map.get(); }
};

struct GpuTsdfMapDetail : public GpuMapDetail
{
  TsdfOptions tsdf_options;
  /// Adds the TSDF layer to the host layout when it is missing (the reference does so in the GpuTsdfMap constructor,
  /// ohmgpu/GpuTsdfMap.cpp:66-77; here it has to exist before the device map is created by the base constructor).
  GpuTsdfMapDetail(OccupancyMap *map_in, bool borrowed);
  const TsdfOptions *tsdf() const override { return &tsdf_options; }
};
}  // namespace ohm

#endif  // OHMHIP_REF_ADAPTOR_HIPMAPBINDING_H
