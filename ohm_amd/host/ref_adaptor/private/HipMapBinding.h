// HipMapBinding.h -- the glue between ohm::OccupancyMap and the glm-free binding core (HipBindingCore.h): one device
// map per host map, shared by every GpuMap / GpuNdtMap / GpuTsdfMap over that map (ohmgpu/private/GpuMapDetail.cpp:42-56).
// Everything here NAMES ohm / glm types and does nothing else: values are read out of the host map into the core's
// plain structs, MapChunk blocks become pointers, stamps become integers.  Needs the reference tree WITH glm: not
// compiled in the development image (README.md in the parent directory says what has been).
#ifndef OHMHIP_REF_ADAPTOR_HIPMAPBINDING_H
#define OHMHIP_REF_ADAPTOR_HIPMAPBINDING_H

#include "HipBindingCore.h"

#include <ohm/NdtMode.h>
#include <ohm/RayFilter.h>
#include <ohm/VoxelTsdf.h>

#include <glm/glm.hpp>

#include <memory>
#include <vector>

namespace ohm
{
class OccupancyMap;
class NdtMap;
using HipMapKind = ohmhip_adaptor::MapKind;

struct HipMapBinding
{
  OccupancyMap *map = nullptr;
  ohmhip_adaptor::BindingCore core;
  size_t gpu_mem_size = 0;

  /// (Re)create the device map for the host map's current layout and upload every region the host holds.
  bool create(HipMapKind kind, const NdtMap *ndt, const TsdfOptions *tsdf);
  /// Host layer index of a device layer id, or -1 when the host map has no such layer.
  int hostLayer(int layer_id) const;
  bool pushConfig(const NdtMap *ndt, const TsdfOptions *tsdf);
  /// Upload regions edited on the CPU since the last sync (GpuLayerCache::upload by stamp).
  bool uploadHostEdits();
  /// GpuLayerCache::syncToMainMemory for the given device layer ids (empty: all enabled layers).
  bool download(const std::vector<int> &layer_ids, bool clear_dirty);
};

/// The binding of @p map (created by gpumap::enableGpu), or null.
HipMapBinding *hipBinding(const OccupancyMap &map);
void registerHipBinding(OccupancyMap &map, HipMapBinding *binding);
void unregisterHipBinding(OccupancyMap &map);

struct GpuMapDetail
{
  OccupancyMap *map = nullptr;
  bool borrowed_map = true;
  HipMapKind kind = HipMapKind::kOccupancy;
  RayFilterFunction ray_filter;   ///< GpuMap::setRayFilter (overrides the map's own filter)
  bool ray_filter_set = false;
  double ray_segment_length = 0;  ///< accepted and ignored: segmenting changes results vs the CPU mapper
  bool grouped_rays = false;
  bool gpu_ok = false;

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
  /// Adds the TSDF layer to the host layout when it is missing (ohmgpu/GpuTsdfMap.cpp:66-77; it has to exist before
  /// the base constructor creates the device map).
  GpuTsdfMapDetail(OccupancyMap *map_in, bool borrowed);
  const TsdfOptions *tsdf() const override { return &tsdf_options; }
};
}  // namespace ohm

#endif  // OHMHIP_REF_ADAPTOR_HIPMAPBINDING_H
