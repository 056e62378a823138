// gpumap_driver.cpp -- C++14 test driver for the host mirror (ohm::GpuMap / GpuNdtMap / GpuTsdfMap over the C ABI).
// Reads a binary ray file, integrates it in batches exactly like the reference's gpuMapTest() harness
// (tests/ohmtestgpu/GpuMapTest.cpp:68-205), syncs, and dumps every region layer for the Python parity test to check
// against the CPU oracle.  Links libohmhip.so only; built with plain g++ (no hipcc, no glm).
//
//   gpumap_driver <mode: occ|occmean|occdev|ndt|tsdf|linekeys|...> <resolution> <batch_rays> <rays.bin> <out.bin>
//   occdev: the sample points (odd entries) go through ohm::GpuTransformSamples with a static identity trajectory and
//   are integrated straight from the device buffer (all rays then start at the origin).
//   rays.bin: u64 n_points, then n_points * 3 doubles.  out.bin: u64 regions, per region i16[3] key, then per enabled
//   layer (ascending id): u32 layer id, u64 bytes, payload.
#include "OhmGpuMap.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

int main(int argc, char **argv)
{
  if (argc < 6)
  {
    std::fprintf(stderr, "usage: %s <occ|occmean|occdev|occcoalesce|occowner|occclipbox|ndt|tsdf> <resolution> <batch_rays> <rays.bin> <out.bin>\n", argv[0]);
    return 2;
  }
  const std::string mode = argv[1];
  const double resolution = std::atof(argv[2]);
  const size_t batch_rays = size_t(std::atoll(argv[3]));
  if (mode.compare(0, 7, "filter:") == 0)
  {
    // Host only (no device): run one of the stock ray filters of OhmGpuMap.h over the rays and write, per ray,
    // accepted (1 byte), filter flags (1 byte), start and end (6 doubles).  Box (-1,-1,-1)..(2,2,2), length = resolution.
    FILE *in = std::fopen(argv[4], "rb");
    uint64_t n_points = 0;
    if (!in || std::fread(&n_points, sizeof(n_points), 1, in) != 1)
    {
      return 4;
    }
    std::vector<ohm::dvec3> rays(n_points);
    if (std::fread(rays.data(), sizeof(ohm::dvec3), n_points, in) != n_points)
    {
      return 4;
    }
    std::fclose(in);
    const ohm::Aabb box(ohm::dvec3{ -1.0, -1.0, -1.0 }, ohm::dvec3{ 2.0, 2.0, 2.0 });
    FILE *out = std::fopen(argv[5], "wb");
    if (!out)
    {
      return 4;
    }
    for (uint64_t i = 0; i + 1 < n_points; i += 2)
    {
      ohm::dvec3 start = rays[i], end = rays[i + 1];
      unsigned flags = 0;
      bool ok = false;
      if (mode == "filter:clipbounded")
      {
        ok = ohm::clipBounded(&start, &end, &flags, box);
      }
      else if (mode == "filter:cliptobounds")
      {
        ok = ohm::clipToBounds(&start, &end, &flags, box);
      }
      else if (mode == "filter:clipray")
      {
        ok = ohm::clipRayFilter(&start, &end, &flags, resolution);
      }
      else if (mode == "filter:goodray")
      {
        ok = ohm::goodRayFilter(&start, &end, &flags, resolution);
      }
      else
      {
        return 2;
      }
      const unsigned char head[2] = { static_cast<unsigned char>(ok), static_cast<unsigned char>(flags) };
      std::fwrite(head, 1, 2, out);
      std::fwrite(&start, sizeof(start), 1, out);
      std::fwrite(&end, sizeof(end), 1, out);
    }
    std::fclose(out);
    return 0;
  }
  try
  {
    if (ohm::configureGpu(0) != 0)
    {
      std::fprintf(stderr, "no HIP device\n");
      return 3;
    }
    FILE *in = std::fopen(argv[4], "rb");
    if (!in)
    {
      return 4;
    }
    uint64_t n_points = 0;
    if (std::fread(&n_points, sizeof(n_points), 1, in) != 1)
    {
      return 4;
    }
    std::vector<ohm::dvec3> rays(n_points);
    if (std::fread(rays.data(), sizeof(ohm::dvec3), n_points, in) != n_points)
    {
      return 4;
    }
    std::fclose(in);

    if (mode == "linekeys")
    {
      // ohm::LineKeysQueryGpu over the rays: out.bin = u64 rays, then per ray u64 index, u64 count, and after them all
      // keys (i16[3] region, u8[3] local each) in intersectedVoxels() order.
      ohm::OccupancyMap query_map(resolution);
      ohm::GpuMap query_gpu_map(&query_map, true);
      ohm::LineKeysQueryGpu query(query_gpu_map);
      query.setRays(rays.data(), rays.size());
      if (!query.executeAsync() || !query.wait() || query.numberOfResults() != rays.size() / 2)
      {
        return 8;
      }
      FILE *out = std::fopen(argv[5], "wb");
      if (!out)
      {
        return 6;
      }
      const uint64_t n = query.numberOfResults();
      std::fwrite(&n, sizeof(n), 1, out);
      uint64_t total_keys = 0;
      for (uint64_t i = 0; i < n; ++i)
      {
        const uint64_t index = query.resultIndices()[i], count = query.resultCounts()[i];
        std::fwrite(&index, sizeof(index), 1, out);
        std::fwrite(&count, sizeof(count), 1, out);
        total_keys = index + count;
      }
      for (uint64_t k = 0; k < total_keys; ++k)
      {
        std::fwrite(query.intersectedVoxels()[k].region, sizeof(int16_t), 3, out);
        std::fwrite(query.intersectedVoxels()[k].local, sizeof(uint8_t), 3, out);
      }
      std::fclose(out);
      return 0;
    }

    ohm::OccupancyMap map(resolution);
    std::unique_ptr<ohm::GpuMap> gpu_map;
    if (mode == "occ" || mode == "occmean" || mode == "occdev" || mode == "occcoalesce" || mode == "occowner" ||
        mode == "occclipbox" || mode == "occpart" || mode == "occpartint")
    {
      if (mode == "occmean")
      {
        map.addLayer(OHMHIP_LID_MEAN);
      }
      gpu_map.reset(new ohm::GpuMap(&map, true, unsigned(batch_rays * 2)));
      if (mode == "occcoalesce")
      {
        gpu_map->setBatchCoalescing(3 * batch_rays + 1);  // every fourth call launches a device batch
      }
      if (mode == "occclipbox")
      {
        // GpuMap.ClipBox (tests/ohmtestgpu/GpuMapTest.cpp:633-647): a RayFilterFunction wrapping clipBounded
        const ohm::Aabb clip_box(ohm::dvec3{ -1.0, -1.0, -1.0 }, ohm::dvec3{ 2.0, 2.0, 2.0 });
        gpu_map->setRayFilter([clip_box](ohm::dvec3 *start, ohm::dvec3 *end, unsigned *filter_flags) {
          return ohm::clipBounded(start, end, filter_flags, clip_box);
        });
      }
      if (mode == "occpart")
      {
        // rank 1 of a two-way partition by a table: region blocks (2 x 2 x 2 regions) with block x >= 1 belong to rank 1
        ohm::GpuMap::RegionPartition part;
        part.world_size = 2;
        part.rank = 1;
        part.block_shift = 1;
        part.grid_origin[0] = 0;
        part.grid_dims[0] = 2;
        part.grid_dims[1] = part.grid_dims[2] = 1;
        part.owners = { 0, 1 };
        gpu_map->setRegionPartition(part);
      }
      if (mode == "occowner")
      {
        gpu_map->setRegionOwnership(2, 1);  // this map is rank 1 of a two-way region partition
      }
    }
    else if (mode == "ndt")
    {
      gpu_map.reset(new ohm::GpuNdtMap(&map, true, unsigned(batch_rays * 2)));
    }
    else if (mode == "tsdf")
    {
      gpu_map.reset(new ohm::GpuTsdfMap(&map, true, unsigned(batch_rays * 2)));
    }
    else
    {
      return 2;
    }
    if (!gpu_map->gpuOk())
    {
      return 5;
    }
    const size_t batch_points = batch_rays ? batch_rays * 2 : size_t(n_points);
    size_t total = 0;
    if (mode == "occdev")
    {
      gputil::Device device;
      gputil::Queue queue = device.defaultQueue();
      ohm::GpuTransformSamples transform(device);
      gputil::Buffer device_rays;
      const double times[2] = { 0.0, 1.0 };
      const ohm::dvec3 translations[2] = { { 0, 0, 0 }, { 0, 0, 0 } };
      const ohm::dquat rotations[2] = { { 0, 0, 0, 1 }, { 0, 0, 0, 1 } };
      std::vector<ohm::dvec3> samples;
      std::vector<double> sample_times;
      for (size_t i = 0; i < n_points; i += batch_points)
      {
        const size_t count = std::min<size_t>(batch_points, n_points - i);
        samples.clear();
        sample_times.clear();
        for (size_t k = 1; k < count; k += 2)
        {
          samples.push_back(rays[i + k]);
          sample_times.push_back(0.25 + 0.5 * double(k) / double(count));
        }
        const unsigned elements = transform.transform(times, translations, rotations, 2, sample_times.data(),
                                                      samples.data(), unsigned(samples.size()), queue, device_rays);
        total += gpu_map->integrateRays(device_rays, elements, ohm::kRfDefault);
      }
    }
    else if (mode == "occpartint")
    {
      // ohm::PartitionedIntegrator at world size 1 over the library's own RCCL communicator: every batch is routed on
      // the device, exchanged (count all-gather + the block a rank addresses to itself) and integrated with the previous
      // batches still in flight -- each batch from a device buffer of its own, rewritten without any wait in between.
      const auto id = ohm::RayCommunicator::uniqueId();
      ohm::RayCommunicator comm(id, 1, 0);
      ohm::GpuMap::RegionPartition whole;  // world size 1: everything belongs to rank 0
      ohm::PartitionedIntegrator integrator(*gpu_map, whole, comm);
      gputil::Buffer device_rays;
      for (size_t i = 0; i < n_points; i += batch_points)
      {
        const size_t count = std::min<size_t>(batch_points, n_points - i);
        if (!device_rays.isValid())
        {
          device_rays.create(batch_points * sizeof(ohm::dvec3));
        }
        device_rays.write(rays.data() + i, count * sizeof(ohm::dvec3));  // (free again when integrateRays returned)
        const size_t done = integrator.integrateRays(device_rays, count, ohm::kRfDefault);
        if (integrator.lastStatus() != OHMHIP_OK || integrator.raysReceived() * 2 != integrator.sendCounts()[0] * 2)
        {
          return 12;
        }
        total += done;
      }
    }
    else if (mode == "occpart")
    {
      // route every batch on the device and integrate the block addressed to this rank (what the all-to-all would
      // deliver from a single source rank); the points of rays that never reach rank 1's territory count as done
      gputil::Device device;
      gputil::Buffer device_rays, routed;
      std::vector<uint32_t> counts;
      for (size_t i = 0; i < n_points; i += batch_points)
      {
        const size_t count = std::min<size_t>(batch_points, n_points - i);
        if (!device_rays.isValid())
        {
          device_rays.create(count * sizeof(ohm::dvec3));
        }
        device_rays.resize(count * sizeof(ohm::dvec3));
        device_rays.write(rays.data() + i, count * sizeof(ohm::dvec3));
        gpu_map->routeRays(device_rays, count, ohm::kRfDefault, routed, counts);
        gputil::Buffer mine;
        if (counts[1])
        {
          std::vector<ohm::dvec3> block(size_t(counts[1]) * 2);
          routed.read(block.data(), block.size() * sizeof(ohm::dvec3), size_t(counts[0]) * 2 * sizeof(ohm::dvec3));
          mine.create(block.size() * sizeof(ohm::dvec3));
          mine.write(block.data(), block.size() * sizeof(ohm::dvec3));
          const size_t done = gpu_map->integrateRays(mine, block.size(), ohm::kRfDefault);
          if (done != block.size())
          {
            return 11;
          }
          gpu_map->syncVoxels();  // (the block buffer goes out of scope: the batch must have read it)
        }
        total += count;
      }
    }
    else
    {
      for (size_t i = 0; i < n_points; i += batch_points)
      {
        const size_t count = std::min<size_t>(batch_points, n_points - i);
        total += gpu_map->integrateRays(rays.data() + i, count, nullptr, nullptr, ohm::kRfDefault);
      }
    }
    if (mode == "occcoalesce")
    {
      gpu_map->gpuCache()->flush();  // GpuCache::flush == syncVoxels
      if (gpu_map->gpuCache()->layerCount() != 1u || gpu_map->gpuCache()->targetGpuAllocSize() != 0u)
      {
        return 9;  // (occupancy only, no memory limit set)
      }
    }
    gpu_map->syncVoxels();
    std::printf("integrated %zu of %llu points, %zu regions\n", total, (unsigned long long)n_points, map.regionCount());

    FILE *out = std::fopen(argv[5], "wb");
    if (!out)
    {
      return 6;
    }
    const uint64_t regions = map.regionCount();
    std::fwrite(&regions, sizeof(regions), 1, out);
    for (const auto &entry : map.chunks())
    {
      std::fwrite(entry.first.data(), sizeof(int16_t), 3, out);
      for (uint32_t layer = 0; layer < uint32_t(OHMHIP_LID_COUNT); ++layer)
      {
        if (!map.hasLayer(int(layer)))
        {
          continue;
        }
        const auto &block = entry.second.voxel_blocks[layer];
        const uint64_t bytes = block.size();
        std::fwrite(&layer, sizeof(layer), 1, out);
        std::fwrite(&bytes, sizeof(bytes), 1, out);
        std::fwrite(block.data(), 1, bytes, out);
      }
    }
    std::fclose(out);
    return (total == n_points) ? 0 : 7;
  }
  catch (const gputil::ApiException &e)
  {
    std::fprintf(stderr, "gputil::ApiException: %s\n", e.what());
    return 10;
  }
}
