// ohmhip_map.hip -- host side of the typed hot-path C ABI: a voxel map resident in HBM, region table, ray batch
// pipeline and region download/upload.  gfx950 / ROCm only; see include/ohmhip.h for reference citations.
#include <cstring>
#include <string.h>

#include "occupancy_kernels.h"
#include "replay_kernels.h"
#include "traversal_kernels.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <unordered_map>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

using namespace ohmhip;

#include "map_state.h"
#include "pool_impl.h"
#include "batch_run.h"
#include "pending_rays.h"

extern "C" {

size_t ohmhip_layer_voxel_bytes(int layer_id)
{
  return (layer_id >= 0 && layer_id < OHMHIP_LID_COUNT) ? kLayerBytes[layer_id] : 0;
}

void ohmhip_map_config_default(ohmhip_map_config *c)
{
  if (!c)
  {
    return;
  }
  std::memset(c, 0, sizeof(*c));
  c->resolution = 0.1;
  c->region_dim[0] = c->region_dim[1] = c->region_dim[2] = 32;  // ohm/OccupancyMap.h:24-26
  c->layers = OHMHIP_LAYER_BIT(OHMHIP_LID_OCCUPANCY);
  c->mode = OHMHIP_MODE_OCCUPANCY;
  // ohm/OccupancyMap.cpp:205-213; probabilityToValue (ohm/MapProbability.h:33-36) in float.
  c->hit_value = std::log(0.9f / (1.0f - 0.9f));
  c->miss_value = std::log(0.45f / (1.0f - 0.45f));
  c->threshold_value = std::log(0.5f / (1.0f - 0.5f));
  c->min_value = -2.0f;
  c->max_value = 3.511f;
  c->ray_filter = OHMHIP_FILTER_GOOD;  // ohm/OccupancyMap.cpp:215-218
  c->ray_filter_range = 1e10;
  // ohm/private/NdtMapDetail.h:20-45
  c->ndt_sensor_noise = 0.05f;
  c->ndt_sample_threshold = 3;
  {
    // NdtMap ctor: adaptation rate from the map's miss probability (ohm/NdtMap.cpp:31-36, ohm/NdtMap.h:146-149).
    const float miss_probability = 1.0f - (1.0f / (1.0f + std::exp(c->miss_value)));
    c->ndt_adaptation_rate = std::max(0.0f, std::min(2.0f * (1.0f - 2.0f * miss_probability), 1.0f));
  }
  c->ndt_reinit_threshold = std::log(0.2f / (1.0f - 0.2f));
  c->ndt_reinit_count = 100;
  c->ndt_initial_intensity_cov = 1.0f;
  // ohm/VoxelTsdf.h:27-37
  c->tsdf_max_weight = 1e4f;
  c->tsdf_trunc = 0.1f;
  c->tsdf_dropoff = 0.0f;
  c->tsdf_sparsity = 1.0f;
  c->gpu_mem_size = 0;
  c->region_capacity = 0;
}

int ohmhip_map_create(ohmhip_map_t *map, const ohmhip_map_config *config)
try
{
  if (!map || !config || !(config->resolution > 0))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  int count = 0;
  const int derr = ohmhip_device_count(&count);
  if (derr || count == 0)
  {
    return derr ? derr : OHMHIP_ERR_NO_DEVICE;
  }
  ohmhip_map_t m = new (std::nothrow) ohmhip_map_s;
  if (!m)
  {
    return OHMHIP_ERR_INTERNAL;
  }
  m->config = *config;
  for (int a = 0; a < 3; ++a)
  {
    if (m->config.region_dim[a] <= 0)
    {
      m->config.region_dim[a] = 32;
    }
    if (m->config.region_dim[a] > 255)
    {
      delete m;
      return OHMHIP_ERR_INVALID_ARG;
    }
  }
  MapConst &mc = m->mc;
  std::memset(&mc, 0, sizeof(mc));
  mc.resolution = m->config.resolution;
  {
    // A region of more than 2^15 voxels is cut into equal tiles that fit the LDS count tile and the 15-bit voxel index
    // of the segment / sample / event keys (tiling_impl.h); a region of up to 32^3 voxels is one tile.
    const int region_dims[3] = { m->config.region_dim[0], m->config.region_dim[1], m->config.region_dim[2] };
    int tile[3];
    chooseTileDims(region_dims, 1 << kHitVoxelBits, tile);
    for (int a = 0; a < 3; ++a)
    {
      mc.kdim[a] = region_dims[a];
      mc.dim[a] = tile[a];
      mc.tile_split[a] = region_dims[a] / tile[a];
      mc.region_dim[a] = mc.kdim[a] * mc.resolution;  // ohm/OccupancyMap.cpp:200-202
      mc.origin[a] = m->config.origin[a];
    }
  }
  mc.region_voxels = mc.dim[0] * mc.dim[1] * mc.dim[2];
  {
    // Fixed-point walk predictor (ohmhip_internal.h, Segment): a TILE's diagonal maps to 2^30 / 1.01 units; the
    // trusted lead covers one truncation per candidate plus one per step a candidate can take inside a tile.
    const double tx = mc.dim[0] * mc.resolution, ty = mc.dim[1] * mc.resolution, tz = mc.dim[2] * mc.resolution;
    const double diagonal = std::sqrt(tx * tx + ty * ty + tz * tz);
    mc.fix_scale = double(kFixMaxDelta) / (1.01 * diagonal);
    mc.fix_margin = 2u * uint32_t(std::max(mc.dim[0], std::max(mc.dim[1], mc.dim[2]))) + 8u;
  }
  applyValueConfig(m);

  int err = OHMHIP_OK;
  auto fail = [&](int e) {
    ohmhip_map_destroy(m);
    return e;
  };
  if (hipGetDevice(&m->device) != hipSuccess)
  {
    return fail(OHMHIP_ERR_NO_DEVICE);
  }
  if ((err = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)) != 0)
  {
    return fail(err);
  }
  if ((err = hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking)) != 0)
  {
    return fail(err);
  }
  if ((err = hipStreamCreateWithFlags(&m->front_stream, hipStreamNonBlocking)) != 0)
  {
    return fail(err);
  }
  for (auto &e : m->ev)
  {
    if ((err = hipEventCreate(&e)) != 0)
    {
      return fail(err);
    }
  }
  for (auto &set : m->tev)
  {
    for (auto &e : set)
    {
      if ((err = hipEventCreate(&e)) != 0)
      {
        return fail(err);
      }
    }
  }
  for (auto &sl : m->ray_slots)
  {
    if ((err = hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming)) != 0 ||
        (err = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)) != 0)
    {
      return fail(err);
    }
  }
  if ((err = hipMalloc(reinterpret_cast<void **>(&m->d_n_slots), sizeof(uint32_t))) != 0)
  {
    return fail(err);
  }
  if ((err = hipMalloc(reinterpret_cast<void **>(&m->d_info), 3 * sizeof(BatchInfo))) != 0)
  {
    return fail(err);
  }
  if ((err = hipMalloc(reinterpret_cast<void **>(&m->d_event_count), 8 * sizeof(uint32_t))) != 0)
  {
    return fail(err);
  }
  if ((err = hipMalloc(reinterpret_cast<void **>(&m->d_dbg), kDbgWords * sizeof(unsigned long long))) != 0)
  {
    return fail(err);
  }
  (void)hipMemset(m->d_dbg, 0, kDbgWords * sizeof(unsigned long long));
  if ((err = hipHostMalloc(reinterpret_cast<void **>(&m->h_info), 2 * sizeof(BatchInfo),
                           hipHostMallocMapped | hipHostMallocCoherent)) != 0 ||
      (err = hipHostGetDevicePointer(reinterpret_cast<void **>(&m->h_info_dev), m->h_info, 0)) != 0)
  {
    return fail(err);
  }
  std::memset(m->h_info, 0, 2 * sizeof(BatchInfo));
  if ((err = hipHostMalloc(reinterpret_cast<void **>(&m->h_passed), 64, hipHostMallocMapped | hipHostMallocCoherent)) != 0 ||
      (err = hipHostGetDevicePointer(reinterpret_cast<void **>(&m->h_passed_dev), m->h_passed, 0)) != 0 ||
      (err = hipEventCreateWithFlags(&m->ev_passed, hipEventDisableTiming)) != 0)
  {
    return fail(err);
  }

  uint32_t capacity = m->config.region_capacity;
  if (capacity == 0)
  {
    const uint64_t budget = m->config.gpu_mem_size ? m->config.gpu_mem_size : (uint64_t(4) << 30);
    capacity = uint32_t(std::max<uint64_t>(64, budget / bytesPerRegionAllLayers(m->config, mc.region_voxels)));
  }
  capacity = std::min<uint32_t>(capacity, kMaxRegionSlots);  // 20-bit slot field of the hit key
  if ((err = allocPool(m, capacity, 0)) != 0)
  {
    return fail(err);
  }
  {
    int device = 0, cus = 0;
    if (hipGetDevice(&device) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
    {
      m->walk_workgroups = uint32_t(cus);
    }
  }
  if (const char *env = std::getenv("OHMHIP_CHUNK_SEGMENTS"))
  {
    m->chunk_segments = uint32_t(std::max(64, std::min(int(kMaxChunkSegments), std::atoi(env))));
  }
  if (const char *env = std::getenv("OHMHIP_EVENT_LIMIT"))
  {
    m->event_limit = uint32_t(std::max(0, std::atoi(env)));
  }
  if (const char *env = std::getenv("OHMHIP_WRITEBACK"))
  {
    m->writeback_off = std::atoi(env) == 0;
  }
  if (const char *env = std::getenv("OHMHIP_WRITEBACK_WGS"))
  {
    m->writeback_workgroups = uint32_t(std::max(1, std::min(4096, std::atoi(env))));
  }
  if (const char *env = std::getenv("OHMHIP_BIN_RAYS"))
  {
    m->bin_rays_per_block = uint32_t(std::max(128, std::min(int(kBinRaysPerBlock), std::atoi(env))));
  }
  if (const char *env = std::getenv("OHMHIP_MIN_CHUNK_SEGMENTS"))
  {
    m->min_chunk_segments = uint32_t(std::max(64, std::min(int(kMaxChunkSegments), std::atoi(env))));
  }
  // Small regions (16^3: at most an eighth of the tile the full shape serves) are walked by half-size workgroups, two
  // per CU.  Measured with the C1 sweep (scripts/half_probe.py, profiles/r06_notes.md): 16^3 regions 1.40 against 1.55 ms
  // per batch; 16 x 16 x 32 regions 1.23 against 1.20 (a wash); 32 x 32 x 16 regions 1.14 against 0.96 -- those hold
  // nearly as many segments as 32^3 ones, the half shape's 4096-segment chunks split most of them, and split regions
  // pay the global count round trip.  (OHMHIP_WALK_HALF_VOXELS moves the threshold for A/B runs.)
  uint32_t half_voxels = uint32_t(WalkHalf::kTileVoxels) / 4u;
  if (const char *env = std::getenv("OHMHIP_WALK_HALF_VOXELS"))
  {
    half_voxels = uint32_t(std::max(0, std::min(int(WalkHalf::kTileVoxels), std::atoi(env))));
  }
  m->walk_half = uint32_t(mc.region_voxels) <= half_voxels;
  if (const char *env = std::getenv("OHMHIP_WALK_HALF"))
  {
    m->walk_half = m->walk_half && std::atoi(env) != 0;
  }
  if (m->walk_half)
  {
    m->chunk_segments = std::min<uint32_t>(m->chunk_segments, WalkHalf::kSegments);
    m->min_chunk_segments = std::min<uint32_t>(m->min_chunk_segments, WalkHalf::kSegments / 4u);
  }
  // The walk kernel keeps a region's count tile, the staged samples and the chunk's segment order in LDS (about
  // 150 KiB of the CU's 160 KiB for 32^3 regions; half of that twice for the half shape): the chunk size gives way if
  // the region tile is large.
  const size_t lds_limit = m->walk_half ? size_t(80) * 1024 : size_t(160) * 1024;
  while (walkLdsBytes(mc, m->chunk_segments, m->walk_half) > lds_limit && m->chunk_segments > 64)
  {
    m->chunk_segments /= 2;
  }
  const size_t lds_bytes = walkLdsBytes(mc, m->chunk_segments, m->walk_half);
  const void *walk_kernels_full[3] = { reinterpret_cast<const void *>(k_region_walk<false, false, WalkFull>),
                                       reinterpret_cast<const void *>(k_region_walk<true, false, WalkFull>),
                                       reinterpret_cast<const void *>(k_region_walk<false, true, WalkFull>) };
  const void *walk_kernels_half[3] = { reinterpret_cast<const void *>(k_region_walk<false, false, WalkHalf>),
                                       reinterpret_cast<const void *>(k_region_walk<true, false, WalkHalf>),
                                       reinterpret_cast<const void *>(k_region_walk<false, true, WalkHalf>) };
  const void *const *walk_kernels = m->walk_half ? walk_kernels_half : walk_kernels_full;
  for (int k = 0; k < 3; ++k)
  {
    if ((err = hipFuncSetAttribute(walk_kernels[k], hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes))) != 0)
    {
      return fail(err);
    }
  }
  if ((err = hipFuncSetAttribute(reinterpret_cast<const void *>(k_region_traversal),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, int(traversalLdsBytes(mc)))) != 0)
  {
    return fail(err);
  }
  if (const char *env = std::getenv("OHMHIP_DEBUG_FLAGS"))
  {
    m->debug_flags = unsigned(std::atoi(env));
    m->phase_timing = m->phase_timing || (m->debug_flags & 256u) != 0;  // (the phase timeline needs every stamp)
  }
  if (const char *env = std::getenv("OHMHIP_PHASE_TIMING"))
  {
    m->phase_timing = std::atoi(env) != 0;
  }
  if (const char *env = std::getenv("OHMHIP_REFILL_MIN_IDLE"))
  {
    m->refill_min_idle = std::max(1, std::min(64, std::atoi(env)));
  }
  {
    // The walk kernel spills ~150 bytes per lane, k_ray_setup 24: the first dispatch that needs more private memory
    // than its hardware queue has set aside stalls in the command processor until the runtime has grown the queue's
    // scratch -- 0.13 ms of idle device between the sample sort and the walk of a fresh map's FIRST batch (kernel trace,
    // round 5; every map brings its own streams).  Paid here instead: one empty launch of the hungriest instantiation
    // on each of the map's streams (no chunk to fetch: the workgroup leaves at once).
    WalkArgs wa{};
    wa.mc = m->mc;
    wa.bs = batchScratch(m);
    wa.n_chunks = 0;
    wa.chunk_cursor = batchEventCount(m) + 1;
    for (hipStream_t stream : { m->stream, m->front_stream })
    {
      if (m->walk_half)
      {
        hipLaunchKernelGGL((k_region_walk<true, false, WalkHalf>), dim3(1), dim3(WalkHalf::kThreads), lds_bytes, stream, wa);
      }
      else
      {
        hipLaunchKernelGGL((k_region_walk<true, false, WalkFull>), dim3(1), dim3(WalkFull::kThreads), lds_bytes, stream, wa);
      }
      if ((err = hipMemsetAsync(batchEventCount(m), 0, 2 * sizeof(uint32_t), stream)) != 0 ||
          (err = hipStreamSynchronize(stream)) != 0)
      {
        return fail(err);
      }
    }
  }
  *map = m;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_destroy(ohmhip_map_t m)
try
{
  if (!m)
  {
    return OHMHIP_OK;
  }
  (void)settleLaunch(m);  // (a batch still being launched by the map's thread)
  if (m->stream)
  {
    (void)hipStreamSynchronize(m->stream);
  }
  if (m->copy_stream)
  {
    (void)hipStreamSynchronize(m->copy_stream);
  }
  if (m->front_stream)
  {
    (void)hipStreamSynchronize(m->front_stream);
  }
  freePool(m);
  m->walks_buf[0].release();
  m->walks_buf[1].release();
  m->hit_keys_a.release();
  m->hit_keys_b.release();
  m->interval_counts.release();
  m->segments.release();
  m->sort_temp.release();
  m->events.release();
  for (int i = 0; i < 2; ++i)
  {
    m->wg_regions[i].release();
    m->wg_region_count[i].release();
  }
  m->group_heads.release();
  m->stop_a.release();
  m->stop_b.release();
  m->merge_slots.release();
  m->merge_keys_dev.release();
  m->merge_delta.release();
  m->merge_observers.release();
  if (m->wb_stream)
  {
    (void)hipStreamSynchronize(m->wb_stream);
    (void)hipStreamDestroy(m->wb_stream);
  }
  if (m->h_use)
  {
    (void)hipHostFree(m->h_use);
  }
  for (auto &ring : m->wb_ring)
  {
    if (ring.jobs_host)
    {
      (void)hipHostFree(ring.jobs_host);
    }
    if (ring.done)
    {
      (void)hipEventDestroy(ring.done);
    }
  }
  m->partition.table_dev.release();
  m->partition.masks.release();
  m->partition.block_counts.release();
  m->partition.totals.release();
  if (m->partition.route_stream)
  {
    (void)hipStreamDestroy(m->partition.route_stream);
  }
  if (m->partition.h_totals)
  {
    (void)hipHostFree(m->partition.h_totals);
  }
  if (m->d_event_count)
  {
    (void)hipFree(m->d_event_count);
  }
  if (m->d_n_slots)
  {
    (void)hipFree(m->d_n_slots);
  }
  if (m->d_info)
  {
    (void)hipFree(m->d_info);
  }
  if (m->h_info)
  {
    (void)hipHostFree(m->h_info);
  }
  if (m->h_passed)
  {
    (void)hipHostFree(m->h_passed);
  }
  if (m->ev_passed)
  {
    (void)hipEventDestroy(m->ev_passed);
  }
  if (m->h_stage)
  {
    (void)hipHostFree(m->h_stage);
  }
  if (m->debug_flags & 512u)
  {
    std::fprintf(stderr,
                 "[ohmhip spill] evictions %llu readmissions %llu | ms: select %.1f copy-out %.1f compact %.1f copy-in %.1f "
                 "store-growth %.1f write-back scheduling (host) %.1f\n",
                 (unsigned long long)m->evictions, (unsigned long long)m->readmissions, m->spill_ms[0], m->spill_ms[1],
                 m->spill_ms[2], m->spill_ms[3], m->spill_ms[5], m->wb_host_ms);
  }
  freeHostStore(m);
  for (auto &sl : m->ray_slots)
  {
    sl.d_rays.release();
    sl.d_times.release();
    sl.d_intens.release();
    sl.d_fflags.release();
    if (sl.h)
    {
      (void)hipHostFree(sl.h);
    }
    if (sl.uploaded)
    {
      (void)hipEventDestroy(sl.uploaded);
    }
    if (sl.done)
    {
      (void)hipEventDestroy(sl.done);
    }
  }
  for (auto &e : m->ev)
  {
    if (e)
    {
      (void)hipEventDestroy(e);
    }
  }
  for (auto &set : m->tev)
  {
    for (auto &e : set)
    {
      if (e)
      {
        (void)hipEventDestroy(e);
      }
    }
  }
  if (m->stream)
  {
    (void)hipStreamDestroy(m->stream);
  }
  if (m->copy_stream)
  {
    (void)hipStreamDestroy(m->copy_stream);
  }
  if (m->front_stream)
  {
    (void)hipStreamDestroy(m->front_stream);
  }
  delete m;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

}  // extern "C"

#include "ray_entry.h"

extern "C" {

int ohmhip_map_integrate_rays(ohmhip_map_t m, const double *rays, size_t element_count, const float *intensities,
                              const double *timestamps, unsigned ray_flags, size_t *integrated)
try
{
  return integrateRaysHost(m, rays, element_count, intensities, timestamps, ray_flags, nullptr, integrated);
}
OHMHIP_ABI_CATCH

int ohmhip_map_integrate_rays_filtered(ohmhip_map_t m, const double *rays, size_t element_count,
                                       const float *intensities, const double *timestamps, unsigned ray_flags,
                                       const unsigned char *filter_flags, size_t *integrated)
try
{
  if (!filter_flags && element_count >= 2)
  {
    if (integrated)
    {
      *integrated = 0;
    }
    return OHMHIP_ERR_INVALID_ARG;
  }
  return integrateRaysHost(m, rays, element_count, intensities, timestamps, ray_flags, filter_flags, integrated);
}
OHMHIP_ABI_CATCH

int ohmhip_map_update_config(ohmhip_map_t m, const ohmhip_map_config *config)
try
{
  if (!m || !config)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  ohmhip_map_config wanted = *config;
  for (int a = 0; a < 3; ++a)
  {
    wanted.region_dim[a] = (wanted.region_dim[a] <= 0) ? 32 : wanted.region_dim[a];
    if (wanted.region_dim[a] != m->config.region_dim[a] || wanted.origin[a] != m->config.origin[a])
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
  }
  if (wanted.resolution != m->config.resolution || wanted.mode != m->config.mode || wanted.layers != m->config.layers)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);  // batches already presented keep the values they were presented under
  if (m->config.mode == OHMHIP_MODE_TSDF && wanted.tsdf_trunc != m->config.tsdf_trunc && m->slots_committed != 0)
  {
    // Free-space TSDF updates are applied as counts, which is exact only while every untouched-by-surface voxel sits at
    // the truncation distance in force: a new distance on a populated map would be an approximation.
    return OHMHIP_ERR_UNSUPPORTED;
  }
  const uint64_t gpu_mem_size = m->config.gpu_mem_size;
  const uint32_t region_capacity = m->config.region_capacity;
  m->config = wanted;
  m->config.gpu_mem_size = gpu_mem_size;
  m->config.region_capacity = region_capacity;
  applyValueConfig(m);
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_batch_coalescing(ohmhip_map_t m, size_t min_rays)
try
{
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  m->coalesce_min_rays = min_rays;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_async_launch(ohmhip_map_t m, int enable)
try
{
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  m->async_launch = enable != 0;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_sync(ohmhip_map_t m)
try
{
  OHMHIP_SETTLE(m);
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->front_stream));
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  if (m->debug_flags & (64u | 128u))
  {
    static unsigned long long c[kDbgWords];
    OHMHIP_CHECK(hipMemcpy(c, m->d_dbg, sizeof(c), hipMemcpyDeviceToHost));
    std::fprintf(stderr,
                 "[ohmhip dbg] wave-steps %llu visits %llu refills %llu flagged wave-steps %llu exact wave-steps %llu "
                 "(exact lane-steps %llu: %llu with 2 voxels left, %llu with 3, %llu on poisoned segments)\n",
                 c[0], c[1], c[2], c[3], c[4], c[7], c[5], c[6], c[8]);
    if (const char *path = std::getenv("OHMHIP_DEBUG_TRACE"))
    {
      if (FILE *f = std::fopen(path, "w"))
      {
        for (size_t b = 0; b < kTraceChunks; ++b)
        {
          const unsigned long long *rec = c + 16 + b * kTraceWords;
          if (rec[1] == 0)
          {
            continue;
          }
          std::fprintf(f, "%zu", b);
          for (int k = 0; k < 28; ++k)
          {
            std::fprintf(f, " %llu", rec[k]);
          }
          std::fprintf(f, "\n");
        }
        std::fclose(f);
      }
    }
    OHMHIP_CHECK(hipMemset(m->d_dbg, 0, sizeof(c)));
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->copy_stream));
  if ((m->debug_flags & 256u) && m->phase_timing && m->batch_seq >= 4)
  {
    // Development aid: when the phases of the last three batches started / ended, relative to the first of them
    // (set-up start, set-up + plan end, bin start, bin end, sort end, walk end, batch end).
    static const int order[7] = { 0, 5, 6, 1, 2, 3, 4 };
    static const char *const names[7] = { "setup>", "plan<", "bin>", "bin<", "sort<", "walk<", "end" };
    hipEvent_t origin = m->tev[(m->batch_seq - 3) % kTimingRing][0];
    for (uint64_t back = 3; back >= 1; --back)
    {
      hipEvent_t *tev = m->tev[(m->batch_seq - back) % kTimingRing];
      std::fprintf(stderr, "[ohmhip timeline] batch -%llu:", (unsigned long long)back);
      for (int k = 0; k < 7; ++k)
      {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, origin, tev[order[k]]);
        std::fprintf(stderr, " %s %.0f", names[k], ms * 1e3f);
      }
      std::fprintf(stderr, "\n");
    }
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_batch_timings(ohmhip_map_t m, uint32_t batches_back, float ms[4])
try
{
  OHMHIP_SETTLE(m);
  if (!m || !ms)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (batches_back >= kTimingRing || uint64_t(batches_back) >= m->batch_seq)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const uint32_t ring = uint32_t((m->batch_seq - 1 - batches_back) % kTimingRing);
  hipEvent_t *tev = m->tev[ring];
  const uint32_t have = m->tev_mask[ring];
  auto has = [&](int k) { return (have >> k) & 1u; };
  ms[0] = ms[1] = ms[2] = ms[3] = 0.0f;
  if (!has(4))
  {
    return OHMHIP_ERR_INTERNAL;
  }
  OHMHIP_CHECK(hipEventSynchronize(tev[4]));
  // Device time the batch cost: first kernel start -> last kernel end where the start was recorded (phase timing), else
  // plan end -> last kernel end; for batches presented back to back the interval between the previous batch's last
  // kernel and this one's (a batch's set-up pass runs on a second stream under the previous batch's last kernels, so
  // batches complete at intervals shorter than first start -> last end).
  if (has(0))
  {
    OHMHIP_CHECK(hipEventElapsedTime(&ms[0], tev[0], tev[4]));
  }
  else if (has(5))
  {
    OHMHIP_CHECK(hipEventElapsedTime(&ms[0], tev[5], tev[4]));
  }
  if (uint64_t(batches_back) + 1 < m->batch_seq && batches_back + 1 < kTimingRing)
  {
    const uint32_t prev_ring = uint32_t((m->batch_seq - 2 - batches_back) % kTimingRing);
    // Only for batches that really overlapped: this batch's plan had ended before the previous batch's last kernel did.
    // A batch presented after the device went idle (one batch per sensor frame, a host wait in between) keeps its own
    // span -- the gap to the previous batch is host idle time, not device time.
    float period = 0, lead = 0;
    if (((m->tev_mask[prev_ring] >> 4) & 1u) && has(5) &&
        hipEventElapsedTime(&lead, m->tev[prev_ring][4], tev[5]) == hipSuccess && lead <= 0 &&
        hipEventElapsedTime(&period, m->tev[prev_ring][4], tev[4]) == hipSuccess && period > 0 &&
        (period < ms[0] || !has(0)))
    {
      ms[0] = period;
    }
  }
  float sort_ms = 0, apply_ms = 0, front_ms = 0, bin_ms = 0;
  if (has(0) && has(5) && has(6) && has(1))
  {
    OHMHIP_CHECK(hipEventElapsedTime(&front_ms, tev[0], tev[5]));
    OHMHIP_CHECK(hipEventElapsedTime(&bin_ms, tev[6], tev[1]));
    ms[1] = front_ms + bin_ms;
  }
  const int pre = int(m->tev_pre_walk[ring]);
  if (has(3) && pre > 0 && has(pre))
  {
    OHMHIP_CHECK(hipEventElapsedTime(&ms[2], tev[pre], tev[3]));
    if (has(1) && pre != 1)
    {
      OHMHIP_CHECK(hipEventElapsedTime(&sort_ms, tev[1], tev[pre]));
    }
    OHMHIP_CHECK(hipEventElapsedTime(&apply_ms, tev[3], tev[4]));
    ms[3] = sort_ms + apply_ms;
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_phase_timing(ohmhip_map_t m, int enable)
try
{
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  m->phase_timing = enable != 0;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_reserve_rays(ohmhip_map_t m, size_t ray_count)
try
{
  if (!m || ray_count >= (size_t(1) << (kHitRayBits - 1)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  OHMHIP_CHECK(hipSetDevice(m->device));
  // What prepare() / sizeBuffers() of a batch of `ray_count` rays would otherwise allocate inside the first call (a
  // hipMalloc is a device synchronisation and ~0.1 ms each: 0.25 ms of a fresh map's first 10^6-ray batch).
  hipStream_t s = m->stream;
  const size_t n = std::max<size_t>(ray_count, 1);
  const uint32_t blocks = uint32_t((n + 127) / 128);  // (the smallest binning workgroup: the most workgroups)
  for (int p = 0; p < 2; ++p)
  {
    OHMHIP_CHECK(m->walks_buf[p].ensure(sizeof(RayWalk) * n, false, s));
    OHMHIP_CHECK(m->wg_regions[p].ensure(sizeof(WgRegion) * size_t(std::min<size_t>(blocks, (n + kBinRaysPerBlock - 1) / kBinRaysPerBlock * 8)) * kLtabSize, false, s));
    OHMHIP_CHECK(m->wg_region_count[p].ensure(sizeof(uint32_t) * size_t(blocks), false, s));
  }
  const bool occupancy = m->config.mode == OHMHIP_MODE_OCCUPANCY;
  // 16 events per ray -- the threshold above which sizeBuffers() grows the list on measured demand only -- and never more
  // than 2^27 (1 GiB of keys): a generous expected_element_count must not allocate gigabytes outside the memory limit
  // (ADVICE r5); a batch that needs more grows the list on demand as before.
  const size_t events = std::min<size_t>(std::max<size_t>(size_t(1) << 20, n * 16), size_t(1) << 27);
  const size_t keys = occupancy ? n : n + events;
  OHMHIP_CHECK(m->hit_keys_a.ensure(sizeof(unsigned long long) * keys, false, s));
  OHMHIP_CHECK(m->hit_keys_b.ensure(sizeof(unsigned long long) * keys, false, s));
  if (occupancy)
  {
    OHMHIP_CHECK(m->interval_counts.ensure(sizeof(uint32_t) * n, true, s));
    OHMHIP_CHECK(m->events.ensure(sizeof(unsigned long long) * events, false, s));
  }
  size_t sort_bytes = 0;
  OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(nullptr, sort_bytes, static_cast<unsigned long long *>(m->hit_keys_a.ptr),
                                                    static_cast<unsigned long long *>(m->hit_keys_b.ptr), keys, 0,
                                                    sortEndBit(m), s));
  OHMHIP_CHECK(m->sort_temp.ensure(sort_bytes, false, s));
  // ray-region segments: the running estimate (10 per ray until a batch has been seen: C1 has 9.3) with some head room
  OHMHIP_CHECK(m->segments.ensure(sizeof(Segment) * size_t(double(n) * std::max(m->segments_per_ray, 10.0) * 1.25), false, s));
  OHMHIP_CHECK(hipStreamSynchronize(s));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_first_ray_time(ohmhip_map_t m, double time)
try
{
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);  // batches already presented keep the base they were presented under
  m->first_ray_time = time;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_first_ray_time(ohmhip_map_t m, double *time)
try
{
  if (!m || !time)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  *time = m->first_ray_time;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_rays_beyond_tiles(ohmhip_map_t m, uint64_t *count)
try
{
  if (!m || !count)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  *count = m->rays_beyond_tiles;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_batches_launched(ohmhip_map_t m, uint64_t *count)
try
{
  if (!m || !count)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  // (no flush of collected rays: asking must not change what runs; only a launch handed to the map's own thread is
  // waited for, its status stays with the map for the next settling call)
  if (m->launch_busy)
  {
    m->launch_thread->wait();
  }
  *count = m->batch_seq;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_cache_stats(ohmhip_map_t m, ohmhip_cache_stats *stats, int reset)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !stats)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  stats->hits = m->cache_hits;
  stats->misses = m->cache_misses;
  stats->full = m->cache_full;
  stats->regions_resident = m->slots_committed;
  stats->region_capacity = m->slot_capacity;
  stats->bytes_per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
  stats->memory_limit = m->memory_limit;
  stats->evictions = m->evictions;
  stats->readmissions = m->readmissions;
  stats->regions_spilled = uint32_t(m->spilled.size());
  stats->spill_enabled = m->spill_enabled ? 1u : 0u;
  stats->writebacks = m->writebacks;
  stats->writeback_hits = m->writeback_hits;
  stats->writeback_stale = m->writeback_stale;
  if (reset)
  {
    m->cache_hits = m->cache_misses = m->cache_full = 0;
    m->evictions = m->readmissions = 0;
    m->writebacks = m->writeback_hits = m->writeback_stale = 0;
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_memory_limit(ohmhip_map_t m, uint64_t bytes)
try
{
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  m->memory_limit = bytes;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_spill_to_host(ohmhip_map_t m, int enable)
try
{
  OHMHIP_SETTLE(m);
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (enable && m->d_merge_base)
  {
    return OHMHIP_ERR_UNSUPPORTED;  // replica-merge maps keep a base copy per region: they do not spill
  }
  if (enable)
  {
    // The host store is pinned memory: reserve what the pool can hold now (pinning is slow -- of the order of a second
    // per few GB -- and belongs here, not into the first batch that overflows the pool).  It grows by slabs on demand.
    // The eager part is capped in bytes (8 GiB): a multi-layer NDT map with a large pool and no memory limit would
    // otherwise pin tens of GB before anything spills.  Reserved BEFORE the mode is switched: a failed reservation
    // leaves spilling off and the coalescing threshold as it was.
    const uint64_t per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
    const uint64_t pool_regions =
      m->memory_limit ? std::min<uint64_t>(m->memory_limit / per_region, kMaxRegionSlots) : m->slot_capacity;
    // (what the pool holds + the quarter an eviction moves out while as much again may still be waiting in the store)
    const uint64_t by_count = std::min<uint64_t>(pool_regions + pool_regions / 2 + 64, 16384);
    const uint64_t by_bytes = std::max<uint64_t>((uint64_t(8) << 30) / std::max<uint64_t>(per_region, 1), 64);
    OHMHIP_CHECK(reserveStoreRecords(m, size_t(std::min(by_count, by_bytes))));
    // A collected batch touches the regions of all its calls at once -- more than any one of them, possibly more than
    // the limit holds: with spilling on every call runs as its own device batch (the caller may still set a threshold).
    m->coalesce_min_rays = 0;
  }
  m->spill_enabled = enable != 0;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_spill_writeback(ohmhip_map_t m, int enable)
try
{
  OHMHIP_SETTLE(m);
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (enable && m->spill_enabled)
  {
    // the copies the write-back keeps ahead of the evictions live in store records of their own: pinned here, not in
    // the middle of a batch (up to half the pool, capped like the store's eager reservation)
    const uint64_t per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
    const uint64_t pool_regions =
      m->memory_limit ? std::min<uint64_t>(m->memory_limit / per_region, kMaxRegionSlots) : m->slot_capacity;
    const uint64_t extra = std::min<uint64_t>(pool_regions / 2 + 64, (uint64_t(4) << 30) / std::max<uint64_t>(per_region, 1));
    OHMHIP_CHECK(reserveStoreRecords(m, m->store.free_records.size() + size_t(extra)));
  }
  m->writeback_off = enable == 0;
  if (m->writeback_off)
  {
    dropPrecleaned(m);
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_last_stats(ohmhip_map_t m, ohmhip_batch_stats *stats)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !stats)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (m->stats_pending)
  {
    float ms[4] = { 0, 0, 0, 0 };
    OHMHIP_CHECK(ohmhip_map_batch_timings(m, 0, ms));
    m->stats.ms_total = ms[0];
    m->stats.ms_setup = ms[1];
    m->stats.ms_walk = ms[2];
    m->stats.ms_apply = ms[3];
    // The previous batch's deferred-event demand sizes the next batch's event list.
    m->event_demand = *reinterpret_cast<const uint32_t *>(&m->h_info[1]);
    m->stats_pending = false;
  }
  *stats = m->stats;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

}  // extern "C"

#include "region_io.h"

#include "spill_impl.h"

extern "C" {

int ohmhip_map_mark_dirty(ohmhip_map_t m, const uint32_t *slots, size_t count)
try
{
  OHMHIP_SETTLE(m);
  if (!m || (count && !slots))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  for (size_t i = 0; i < count; ++i)
  {
    if (slots[i] >= m->slot_capacity)
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
  }
  if (count)
  {
    // (the caller rewrote these slots behind the library's back: no use stamp moved, so the background write-back's
    // copies cannot tell -- all of them are void)
    dropPrecleaned(m);
    OHMHIP_CHECK(m->merge_slots.ensure(sizeof(uint32_t) * count, false, m->stream));
    OHMHIP_CHECK(hipMemcpyAsync(m->merge_slots.ptr, slots, sizeof(uint32_t) * count, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_or_at_u32, dim3(64), dim3(256), 0, m->stream, m->d_dirty,
                       static_cast<const uint32_t *>(m->merge_slots.ptr), count, kDirtySync | kDirtyMerge);
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  return hipGetLastError();
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_region_ownership(ohmhip_map_t m, uint32_t world_size, uint32_t rank, int block_shift)
try
{
  if (!m || (world_size > 1 && rank >= world_size) || block_shift < 0 || block_shift > 15)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  ohmhip_map_sync(m);
  if (m->slots_committed != 0)
  {
    return OHMHIP_ERR_INVALID_ARG;  // regions integrated under another partition would be left behind
  }
  m->mc.owner_world = (world_size > 1) ? world_size : 0u;
  m->mc.owner_rank = (world_size > 1) ? rank : 0u;
  m->mc.owner_shift = block_shift;
  m->mc.owner_table = nullptr;  // the block hash deals the regions (a table: ohmhip_map_set_region_partition)
  m->partition.table_host.clear();
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_region_owner(const int16_t *keys_xyz, size_t count, int block_shift, uint32_t world_size, uint32_t *owners)
try
{
  if ((count && (!keys_xyz || !owners)) || block_shift < 0 || block_shift > 15)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  for (size_t i = 0; i < count; ++i)
  {
    owners[i] = (world_size > 1) ? regionOwner(keys_xyz[i * 3], keys_xyz[i * 3 + 1], keys_xyz[i * 3 + 2], block_shift,
                                               world_size)
                                 : 0u;
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_line_keys(ohmhip_map_t m, const double *lines, size_t line_count, uint32_t max_keys_per_line,
                         void *keys_out, uint32_t *counts_out)
try
{
  if (!m || (line_count && (!lines || !keys_out || !counts_out)) || max_keys_per_line == 0)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (line_count == 0)
  {
    return OHMHIP_OK;
  }
  OHMHIP_CHECK(settleLaunch(m));  // (the query shares the map's stream and reads its configuration)
  hipStream_t s = m->stream;
  const size_t key_bytes = sizeof(GpuKeyOut) * line_count * size_t(max_keys_per_line);
  double *d_lines = nullptr;
  GpuKeyOut *d_keys = nullptr;
  uint32_t *d_counts = nullptr;
  int status = OHMHIP_OK;
  auto cleanup = [&]() {
    (void)hipFree(d_lines);
    (void)hipFree(d_keys);
    (void)hipFree(d_counts);
  };
  if ((status = hipMalloc(reinterpret_cast<void **>(&d_lines), sizeof(double) * 6 * line_count)) != 0 ||
      (status = hipMalloc(reinterpret_cast<void **>(&d_keys), key_bytes)) != 0 ||
      (status = hipMalloc(reinterpret_cast<void **>(&d_counts), sizeof(uint32_t) * line_count)) != 0)
  {
    cleanup();
    return status;
  }
  status = hipMemcpyAsync(d_lines, lines, sizeof(double) * 6 * line_count, hipMemcpyHostToDevice, s);
  if (!status)
  {
    hipLaunchKernelGGL(k_line_keys, dim3(uint32_t((line_count + 255) / 256)), dim3(256), 0, s, m->mc, d_lines,
                       uint32_t(line_count), max_keys_per_line, d_keys, d_counts);
    status = hipMemcpyAsync(keys_out, d_keys, key_bytes, hipMemcpyDeviceToHost, s);
  }
  if (!status)
  {
    status = hipMemcpyAsync(counts_out, d_counts, sizeof(uint32_t) * line_count, hipMemcpyDeviceToHost, s);
  }
  if (!status)
  {
    status = hipStreamSynchronize(s);
  }
  cleanup();
  return status;
}
OHMHIP_ABI_CATCH

int ohmhip_map_clear(ohmhip_map_t m)
try
{
  OHMHIP_SETTLE(m);
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  dropPrecleaned(m);
  m->slots_committed = 0;
  m->region_slots.clear();
  m->slot_keys_host.clear();
  for (auto &entry : m->spilled)
  {
    releaseStoreRecord(m, entry.second.record);
  }
  m->spilled.clear();
  return allocPool(m, m->slot_capacity, 0);
}
OHMHIP_ABI_CATCH

}  // extern "C"

#include "merge_impl.h"
#include "partition_impl.h"
#include "tiling_impl.h"
