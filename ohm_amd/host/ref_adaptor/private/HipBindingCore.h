// HipBindingCore.h -- the glm-free half of the Level-2 adaptor: everything the ohm::GpuMap / GpuCache
// member definitions of this directory DO, written against the C ABI (include/ohmhip.h) and plain pointers only --
// no ohm:: type, no glm type.  It is compiled and run in this repository (ohm_amd/lib/binding_core_driver,
// tests/test_gpu_binding_core.py: upload -> integrate -> stamp-checked download against the oracle); what is left in
// GpuMap.cpp / GpuCache.cpp / GpuNdtMap.cpp / GpuTsdfMap.cpp / HipMapBinding.cpp is the glue that names glm and ohm types
// (glm::dvec3* -> double*, glm::i16vec3 -> int16_t[3], MapChunk -> block pointers and stamps), which needs the
// reference's headers WITH glm and has not been through a compiler here (README.md in the parent directory).
//
// Reference behaviour restated here, by piece:
//   config values          ohmgpu/GpuMap.cpp:1036-1191 reads the map's probabilities / clamps at every launch
//   layer selection        ohmgpu/GpuMap.cpp:429-470 (caches per layer the map's layout offers), GpuNdtMap.cpp:80-110
//   upload of CPU edits    ohmgpu/GpuLayerCache.cpp:462-502 (chunk stamp newer than the cache's -> upload)
//   download + stamps      ohmgpu/GpuLayerCache.cpp:670-700 (syncToMainMemory), private/GpuMapDetail.cpp:28-31
//   host ray filter pass   ohmgpu/GpuMap.cpp:736-746
//   GpuCacheId -> layer    ohmgpu/GpuCache.h:32-44
//   region walk            ohmgpu/GpuMap.cpp:106-176 (gpumap::walkRegions)
#ifndef OHMHIP_REF_ADAPTOR_HIPBINDINGCORE_H
#define OHMHIP_REF_ADAPTOR_HIPBINDINGCORE_H

#include <ohmhip.h>

#include <cstddef>
#include <cstdint>
#include <vector>

namespace ohmhip_adaptor
{
/// What kind of map object drives the device map (fixes ohmhip_map_config::mode).
enum class MapKind
{
  kOccupancy,
  kNdtOccupancy,
  kNdtTraversability,
  kTsdf
};

/// The values GpuMap re-reads from its OccupancyMap before every batch.
struct MapValues
{
  double resolution = 0;
  int region_dim[3] = { 32, 32, 32 };
  double origin[3] = { 0, 0, 0 };
  float hit_value = 0, miss_value = 0, threshold_value = 0, min_value = 0, max_value = 0;
  bool saturate_at_min = false, saturate_at_max = false;
};

/// NdtMap parameters (ohm/NdtMap.h:120-160); `present` false: not an NDT mapper.
struct NdtValues
{
  bool present = false;
  float sensor_noise = 0;
  unsigned sample_threshold = 0;
  float adaptation_rate = 0;
  float reinit_threshold = 0;
  unsigned reinit_count = 0;
  float initial_intensity_cov = 0;
};

/// ohm::TsdfOptions (ohm/VoxelTsdf.h:22-40); `present` false: not a TSDF mapper.
struct TsdfValues
{
  bool present = false;
  float max_weight = 0, default_truncation_distance = 0, dropoff_epsilon = 0, sparsity_compensation_factor = 0;
};

/// A ray filter as the glue hands it over: the reference's RayFilterFunction behind a plain function pointer.  `start`
/// and `end` may be moved; returns false to drop the ray; `filter_flags` receives the RayFilterFlag bits.
using RayFilterC = bool (*)(double start[3], double end[3], unsigned *filter_flags, void *user);

/// Host layer name of a device layer id -- the strings of ohm/DefaultLayer.cpp (default_layer::*LayerName()); the glue
/// looks them up in the host MapLayout.  Null for an unknown id.
const char *hostLayerName(int layer_id);
/// GpuCacheId (ohmgpu/GpuCache.h:32-44) -> device layer id, -1 for caches this path does not have (clearance).
int cacheIdToLayer(unsigned cache_id);
/// Layers whose download must be followed by MapChunk::invalidateFirstValidIndex + searchAndUpdateFirstValid
/// (ohmgpu/private/GpuMapDetail.cpp:28-31).
bool layerCarriesFirstValid(int layer_id);
/// Device layers a mapper of `kind` integrates into, given the layers the host layout has (bit per OHMHIP_LID_*).
unsigned deviceLayers(unsigned host_layer_bits, MapKind kind);
/// The value half of a configuration (probabilities, clamps, geometry, mode, NDT / TSDF parameters).
void fillConfig(ohmhip_map_config &cfg, const MapValues &values, MapKind kind, const NdtValues &ndt,
                const TsdfValues &tsdf);

/// The regions a segment touches, in walk order (gpumap::walkRegions): `visit(key, user)` is called for the start
/// region and then for every region entered.  `region_extent`: edge lengths of a region; `start_centre`: spatial centre
/// of the start region.
void walkRegionKeys(const double start[3], const double end[3], const int16_t start_key[3], const int16_t end_key[3],
                    const double region_extent[3], const double start_centre[3],
                    void (*visit)(const int16_t key[3], void *user), void *user);

/// One device map and the bookkeeping that keeps a host map and it in step.
class BindingCore
{
public:
  BindingCore() = default;
  BindingCore(const BindingCore &) = delete;
  BindingCore &operator=(const BindingCore &) = delete;
  ~BindingCore();

  /// (Re)create the device map.  The caller uploads what the host holds afterwards (syncedStamp() is 0 again).
  bool create(MapKind kind, const MapValues &values, unsigned host_layer_bits, size_t gpu_mem_size, const NdtValues &ndt,
              const TsdfValues &tsdf);
  void destroy();
  bool valid() const { return hip_ != nullptr; }
  ohmhip_map_t handle() const { return hip_; }
  MapKind kind() const { return kind_; }
  unsigned layers() const { return config_.layers; }
  unsigned layerCount() const;
  int lastStatus() const { return last_status_; }
  const ohmhip_map_config &config() const { return config_; }

  /// Send values that changed since the last batch; nothing is sent when none did.
  bool pushConfig(const MapValues &values, const NdtValues &ndt, const TsdfValues &tsdf);

  // ---- upload of CPU-side edits (GpuLayerCache::upload by stamp) --------------------------------------------------
  /// Host map stamp up to which host and device agree: a region whose dirty stamp is newer was edited on the CPU.
  uint64_t syncedStamp() const { return synced_stamp_; }
  bool needsUpload(uint64_t region_dirty_stamp) const { return region_dirty_stamp > synced_stamp_; }
  /// Upload one layer's blocks of `count` regions (keys: 3 x int16 each, MapChunk layout blocks).
  bool uploadBlocks(int layer_id, const int16_t *keys_xyz, size_t count, const void *const *blocks);
  /// The upload pass is complete: host and device agree up to `map_stamp`.
  void uploadsDone(uint64_t map_stamp) { synced_stamp_ = map_stamp; }

  // ---- download (GpuLayerCache::syncToMainMemory) ------------------------------------------------------------------
  /// Keys of the regions modified on the device since the last clearing download.
  bool dirtyRegions(std::vector<int16_t> &keys_xyz);
  /// Is `layer_id` part of a download restricted to `only` (empty: every enabled layer)?
  bool downloadsLayer(int layer_id, const std::vector<int> &only) const;
  /// Download one layer's blocks of `count` regions into the host blocks.
  bool downloadBlocks(int layer_id, const int16_t *keys_xyz, size_t count, void *const *blocks);
  /// The download pass is complete; `clear_dirty`: forget the device's modified marks.  What was just written is not a
  /// CPU-side edit: host and device agree up to `map_stamp`.
  bool downloadsDone(bool clear_dirty, uint64_t map_stamp);

  // ---- batches ----------------------------------------------------------------------------------------------------
  /// GpuMap::integrateRays body: `rays` are element_count points (origin, sample pairs, 3 doubles each).  With a filter
  /// the host pass drops / moves rays first.  Returns the number of points accepted, 0 on failure (lastStatus()).
  size_t integrate(const double *rays, size_t element_count, const float *intensities, const double *timestamps,
                   unsigned region_update_flags, RayFilterC filter, void *filter_user);
  /// Wait for queued device work (GpuMap::waitOnPreviousOperation, the destructor's fence).
  void sync();

  // ---- GpuCache face ----------------------------------------------------------------------------------------------
  /// Drop residency without download; the host copy is authoritative afterwards (everything uploads again).
  void clearResidency();
  void removeRegion(const int16_t key[3]);
  bool cacheStats(ohmhip_cache_stats &stats);

private:
  ohmhip_map_t hip_ = nullptr;
  ohmhip_map_config config_{};
  MapKind kind_ = MapKind::kOccupancy;
  uint64_t synced_stamp_ = 0;
  int last_status_ = OHMHIP_OK;
  // scratch of host-filtered batches
  std::vector<double> kept_rays_;
  std::vector<float> kept_intensities_;
  std::vector<double> kept_timestamps_;
  std::vector<unsigned char> kept_flags_;
};
}  // namespace ohmhip_adaptor

#endif  // OHMHIP_REF_ADAPTOR_HIPBINDINGCORE_H
