// binding_core_driver.cpp -- GPU test driver for the glm-free binding core (HipBindingCore.h): plays the part of the
// glue (HipMapBinding.cpp) with a minimal host map -- regions as plain blocks in the MapChunk layout, a map stamp and a
// dirty stamp per region, like ohm::OccupancyMap / MapChunk keep them (ohm/MapChunk.h:33-60) -- and drives the core
// through what ohm::GpuMap does with it:
//   phase A  create, integrate the first half of the rays in batches (every other batch through a host ray filter that
//            keeps everything), stamp-checked download into the host blocks
//   phase B  destroy the device map (GpuCache::clear / a second GpuMap over the same OccupancyMap), create it again,
//            upload every host region whose dirty stamp is newer than the core's synced stamp -- all of them --,
//            integrate the second half, download
//   phase C  one CPU-side edit (a region's dirty stamp moves past the synced stamp): exactly that region uploads
// and writes every region layer for tests/test_gpu_binding_core.py to compare with the CPU oracle integrating all rays.
// Built by __graft_entry__.build() with plain g++ (no hipcc, no glm); links libohmhip.so only.
//
//   binding_core_driver <mode: occ|occmean|ndt|tsdf> <resolution> <batch_rays> <rays.bin> <out.bin>
//   (file formats: ohm_amd/host/gpumap_driver.cpp)
#include "HipBindingCore.h"

#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace ohmhip_adaptor;

namespace
{
struct HostRegion
{
  std::map<int, std::vector<unsigned char>> blocks;  // device layer id -> MapChunk-layout block
  uint64_t dirty_stamp = 0;
};

struct HostMap
{
  std::map<std::array<int16_t, 3>, HostRegion> regions;
  uint64_t stamp = 0;
  uint64_t touch() { return ++stamp; }
};

void clearBlock(int layer_id, std::vector<unsigned char> &block)
{
  if (layer_id == OHMHIP_LID_OCCUPANCY)
  {
    const uint32_t inf = 0x7f800000u;  // unobserved (ohm/DefaultLayer.cpp:87-91)
    for (size_t i = 0; i + 4 <= block.size(); i += 4)
    {
      std::memcpy(&block[i], &inf, 4);
    }
  }
}

#define REQUIRE(cond)                                                                  \
  if (!(cond))                                                                         \
  {                                                                                    \
    std::fprintf(stderr, "binding_core_driver: %s failed (line %d, status %d)\n", #cond, __LINE__, core.lastStatus()); \
    return 3;                                                                          \
  }

/// The glue's uploadHostEdits(): regions whose dirty stamp is newer than the core's synced stamp.  Returns the count.
int uploadHostEdits(BindingCore &core, HostMap &map, size_t &uploaded)
{
  std::vector<int16_t> keys;
  std::vector<HostRegion *> picked;
  for (auto &entry : map.regions)
  {
    if (core.needsUpload(entry.second.dirty_stamp))
    {
      keys.insert(keys.end(), entry.first.begin(), entry.first.end());
      picked.push_back(&entry.second);
    }
  }
  uploaded = picked.size();
  for (int id = 0; id < OHMHIP_LID_COUNT && !picked.empty(); ++id)
  {
    if (!(core.layers() & OHMHIP_LAYER_BIT(id)))
    {
      continue;
    }
    std::vector<const void *> srcs;
    for (HostRegion *region : picked)
    {
      srcs.push_back(region->blocks.at(id).data());
    }
    if (!core.uploadBlocks(id, keys.data(), picked.size(), srcs.data()))
    {
      return 3;
    }
  }
  core.uploadsDone(map.stamp);
  return 0;
}

/// The glue's download(): dirty regions -> host blocks, stamps as GpuLayerCache::syncToMainMemory sets them.
int download(BindingCore &core, HostMap &map, size_t region_voxels, size_t &downloaded)
{
  std::vector<int16_t> keys;
  if (!core.dirtyRegions(keys))
  {
    return 3;
  }
  const size_t count = keys.size() / 3;
  downloaded = count;
  std::vector<HostRegion *> regions(count);
  for (size_t i = 0; i < count; ++i)
  {
    HostRegion &region = map.regions[{ keys[3 * i], keys[3 * i + 1], keys[3 * i + 2] }];
    for (int id = 0; id < OHMHIP_LID_COUNT; ++id)
    {
      if ((core.layers() & OHMHIP_LAYER_BIT(id)) && region.blocks[id].empty())
      {
        region.blocks[id].assign(region_voxels * ohmhip_layer_voxel_bytes(id), 0);
        clearBlock(id, region.blocks[id]);
      }
    }
    regions[i] = &region;
  }
  const uint64_t stamp = count ? map.touch() : map.stamp;
  for (int id = 0; id < OHMHIP_LID_COUNT && count; ++id)
  {
    if (!core.downloadsLayer(id, {}))
    {
      continue;
    }
    std::vector<void *> dsts;
    for (HostRegion *region : regions)
    {
      dsts.push_back(region->blocks.at(id).data());
    }
    if (!core.downloadBlocks(id, keys.data(), count, dsts.data()))
    {
      return 3;
    }
    for (HostRegion *region : regions)
    {
      region->dirty_stamp = stamp;
    }
  }
  return core.downloadsDone(true, map.stamp) ? 0 : 3;
}

bool keepAll(double start[3], double end[3], unsigned *flags, void *user)
{
  (void)start;
  (void)end;
  *flags = 0;
  ++*static_cast<size_t *>(user);
  return true;
}
}  // namespace

int main(int argc, char **argv)
{
  if (argc < 6)
  {
    std::fprintf(stderr, "usage: %s <occ|occmean|ndt|tsdf> <resolution> <batch_rays> <rays.bin> <out.bin>\n", argv[0]);
    return 2;
  }
  const std::string mode = argv[1];
  const size_t batch_rays = size_t(std::atoll(argv[3]));
  FILE *in = std::fopen(argv[4], "rb");
  uint64_t n_points = 0;
  if (!in || std::fread(&n_points, sizeof(n_points), 1, in) != 1)
  {
    return 4;
  }
  std::vector<double> rays(3 * n_points);
  if (std::fread(rays.data(), sizeof(double), rays.size(), in) != rays.size())
  {
    return 4;
  }
  std::fclose(in);

  MapValues values;
  values.resolution = std::atof(argv[2]);
  // the reference's defaults (ohm/OccupancyMap.cpp:205-213, ohm/private/NdtMapDetail.h:20-45, ohm/VoxelTsdf.h:22-40) as
  // the C ABI states them -- what an ohm::OccupancyMap / NdtMap / TsdfOptions would hand the glue
  ohmhip_map_config def;
  ohmhip_map_config_default(&def);
  values.hit_value = def.hit_value;
  values.miss_value = def.miss_value;
  values.threshold_value = def.threshold_value;
  values.min_value = def.min_value;
  values.max_value = def.max_value;
  values.saturate_at_min = def.saturate_at_min != 0;
  values.saturate_at_max = def.saturate_at_max != 0;
  NdtValues ndt;
  TsdfValues tsdf;
  MapKind kind = MapKind::kOccupancy;
  unsigned host_layers = OHMHIP_LAYER_BIT(OHMHIP_LID_OCCUPANCY);
  if (mode == "occmean")
  {
    host_layers |= OHMHIP_LAYER_BIT(OHMHIP_LID_MEAN);
  }
  else if (mode == "ndt")
  {
    kind = MapKind::kNdtOccupancy;
    host_layers |= OHMHIP_LAYER_BIT(OHMHIP_LID_MEAN) | OHMHIP_LAYER_BIT(OHMHIP_LID_COVARIANCE);
    ndt.present = true;
    ndt.sensor_noise = def.ndt_sensor_noise;
    ndt.sample_threshold = def.ndt_sample_threshold;
    ndt.adaptation_rate = def.ndt_adaptation_rate;
    ndt.reinit_threshold = def.ndt_reinit_threshold;
    ndt.reinit_count = def.ndt_reinit_count;
    ndt.initial_intensity_cov = def.ndt_initial_intensity_cov;
  }
  else if (mode == "tsdf")
  {
    kind = MapKind::kTsdf;
    host_layers |= OHMHIP_LAYER_BIT(OHMHIP_LID_TSDF);  // (the host layout keeps its occupancy layer: not integrated into)
    tsdf.present = true;
    tsdf.max_weight = def.tsdf_max_weight;
    tsdf.default_truncation_distance = def.tsdf_trunc;
    tsdf.dropoff_epsilon = def.tsdf_dropoff;
    tsdf.sparsity_compensation_factor = def.tsdf_sparsity;
  }
  const size_t region_voxels = 32u * 32u * 32u;

  HostMap map;
  BindingCore core;
  const size_t half_points = (n_points / 4) * 2;
  size_t filtered_seen = 0, integrated = 0, uploaded = 0, downloaded = 0;
  auto integrateRange = [&](size_t first_point, size_t end_point) -> int {
    size_t batch = 0;
    for (size_t i = first_point; i < end_point; i += 2 * batch_rays, ++batch)
    {
      const size_t count = std::min<size_t>(2 * batch_rays, end_point - i);
      // the values GpuMap re-reads from the host map before every batch, then the CPU-side edits
      if (!core.pushConfig(values, ndt, tsdf) || uploadHostEdits(core, map, uploaded) != 0)
      {
        return 3;
      }
      map.touch();
      const bool with_filter = (batch & 1u) != 0;
      const size_t done = core.integrate(rays.data() + 3 * i, count, nullptr, nullptr, 0u,
                                         with_filter ? keepAll : nullptr, &filtered_seen);
      if (core.lastStatus() != OHMHIP_OK || done != count)
      {
        return 3;
      }
      integrated += done;
    }
    return 0;
  };

  // ---- phase A
  REQUIRE(core.create(kind, values, host_layers, size_t(1) << 30, ndt, tsdf));
  REQUIRE(core.syncedStamp() == 0 && core.layerCount() >= 1);
  REQUIRE(integrateRange(0, half_points) == 0);
  REQUIRE(download(core, map, region_voxels, downloaded) == 0);
  const size_t regions_after_a = map.regions.size();
  REQUIRE(downloaded == regions_after_a && regions_after_a > 0);
  REQUIRE(core.syncedStamp() == map.stamp);
  REQUIRE(uploadHostEdits(core, map, uploaded) == 0 && uploaded == 0);  // a download is not a CPU-side edit
  REQUIRE(download(core, map, region_voxels, downloaded) == 0 && downloaded == 0);  // ... and cleared the dirty marks

  // ---- phase B: the device map goes away; the host copy is authoritative and travels back by stamp
  REQUIRE(core.create(kind, values, host_layers, size_t(1) << 30, ndt, tsdf));
  REQUIRE(core.syncedStamp() == 0);
  REQUIRE(uploadHostEdits(core, map, uploaded) == 0 && uploaded == regions_after_a);
  REQUIRE(integrateRange(half_points, n_points) == 0);
  REQUIRE(download(core, map, region_voxels, downloaded) == 0 && downloaded > 0);

  // ---- phase C: a CPU-side edit of ONE region (content unchanged: the stamp is what the protocol looks at)
  map.regions.begin()->second.dirty_stamp = map.touch();
  REQUIRE(uploadHostEdits(core, map, uploaded) == 0 && uploaded == 1);
  REQUIRE(download(core, map, region_voxels, downloaded) == 0);  // (an upload marks nothing modified on the device)
  core.sync();
  ohmhip_cache_stats stats{};
  REQUIRE(core.cacheStats(stats) && stats.regions_resident == map.regions.size());
  REQUIRE(integrated == n_points && filtered_seen > 0);

  FILE *out = std::fopen(argv[5], "wb");
  if (!out)
  {
    return 4;
  }
  const uint64_t n_regions = map.regions.size();
  std::fwrite(&n_regions, sizeof(n_regions), 1, out);
  for (const auto &entry : map.regions)
  {
    std::fwrite(entry.first.data(), sizeof(int16_t), 3, out);
    for (int id = 0; id < OHMHIP_LID_COUNT; ++id)
    {
      if (!(core.layers() & OHMHIP_LAYER_BIT(id)))
      {
        continue;
      }
      const std::vector<unsigned char> &block = entry.second.blocks.at(id);
      const uint32_t lid = uint32_t(id);
      const uint64_t bytes = block.size();
      std::fwrite(&lid, sizeof(lid), 1, out);
      std::fwrite(&bytes, sizeof(bytes), 1, out);
      std::fwrite(block.data(), 1, block.size(), out);
    }
  }
  std::fclose(out);
  std::printf("binding_core_driver: %zu regions, %zu points integrated (%zu rays through the host filter pass)\n",
              map.regions.size(), integrated, filtered_seen);
  return 0;
}
