// ray_entry.h -- the integrate entry points' bodies: device-pointer batches (direct and collected), request validation, host
// staging with the filter evaluated on the host, piecewise upload, integrateRaysHost.
//
// Part of ohmhip_map.hip's translation unit (included there, in order): not a stand-alone header.
#ifndef OHMHIP_RAY_ENTRY_H
#define OHMHIP_RAY_ENTRY_H

namespace
{
int integrateRaysDevice(ohmhip_map_t m, const double *d_rays, size_t element_count, const float *d_intensities,
                        const double *d_timestamps, unsigned ray_flags, size_t *integrated,
                        const unsigned char *d_filter_flags)
{
  // Kernels take MapConst by value at launch: the batch's filter-flag array rides in it for the calls below.
  struct FlagScope
  {
    ohmhip_map_t m;
    ~FlagScope()
    {
      if (m)
      {
        m->mc.batch_filter_flags = nullptr;
      }
    }
  } flag_scope{ m };
  if (m)
  {
    m->mc.batch_filter_flags = d_filter_flags;
  }
  if (integrated)
  {
    *integrated = 0;
  }
  if (!m || (!d_rays && element_count))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_CHECK(validateBatchRequest(m, ray_flags));
  const size_t n_rays = element_count / 2;
  if (n_rays == 0)
  {
    return OHMHIP_OK;
  }
  if (n_rays >= (size_t(1) << (kHitRayBits - 1)))
  {
    return OHMHIP_ERR_INVALID_ARG;  // split larger batches at the caller (29-bit ray index in the hit key)
  }
  if (d_timestamps && m->first_ray_time < 0)
  {
    // OccupancyMap::updateFirstRayTime(*timestamps) (ohm/OccupancyMap.cpp:343-347)
    double first = 0;
    OHMHIP_CHECK(hipMemcpyAsync(&first, d_timestamps, sizeof(double), hipMemcpyDeviceToHost, m->stream));
    OHMHIP_CHECK(hipStreamSynchronize(m->stream));
    m->first_ray_time = first;
  }
  int err = OHMHIP_ERR_UNSUPPORTED;
  m->batch_exceeds_limit = false;
  switch (m->config.mode)
  {
  case OHMHIP_MODE_OCCUPANCY:
    if (!m->layers[OHMHIP_LID_OCCUPANCY])
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    err = integrateBatch(m, d_rays, d_intensities, d_timestamps, uint32_t(n_rays), ray_flags);
    break;
  case OHMHIP_MODE_NDT_OM:
  case OHMHIP_MODE_NDT_TM:
    if (!m->layers[OHMHIP_LID_OCCUPANCY] || !m->layers[OHMHIP_LID_MEAN] || !m->layers[OHMHIP_LID_COVARIANCE])
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    if (m->config.mode == OHMHIP_MODE_NDT_TM && (!m->layers[OHMHIP_LID_INTENSITY] || !m->layers[OHMHIP_LID_HIT_MISS]))
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    err = integrateBatch(m, d_rays, d_intensities, d_timestamps, uint32_t(n_rays), ray_flags);
    break;
  case OHMHIP_MODE_TSDF:
    if (!m->layers[OHMHIP_LID_TSDF])
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    err = integrateBatch(m, d_rays, d_intensities, d_timestamps, uint32_t(n_rays), ray_flags);
    break;
  default:
    break;
  }
  if (err == OHMHIP_ERR_CAPACITY && m->spill_enabled && m->batch_exceeds_limit && n_rays >= 2 &&
      !m->layers[OHMHIP_LID_TRAVERSAL])
  {
    // (only this cause: a full hash, a refused hipMalloc or the slot field's end would cost log2(n) futile attempts,
    // each with a roll-back and a device synchronisation -- ADVICE r5)
    // The batch alone touches more regions than the residency limit leaves room for (even with everything else moved
    // to the host store).  The reference meets a full cache in the middle of a batch by finalising what it has
    // enqueued and carrying on with the rest (ohmgpu/GpuMap.cpp:900-996, enqueueRegions' flush / retry); the
    // counterpart here: the failed attempt left the map as it was, so the batch is presented again as two halves in
    // ray order -- the CPU mappers integrate ray by ray, a batch boundary means nothing to them (except for the
    // traversal layer, whose exit range is carried within a call: such maps keep failing cleanly).
    const size_t half = n_rays / 2;
    size_t done_a = 0, done_b = 0;
    err = integrateRaysDevice(m, d_rays, half * 2, d_intensities, d_timestamps, ray_flags, &done_a, d_filter_flags);
    const ohmhip_batch_stats stats_a = m->stats;
    if (err == OHMHIP_OK)
    {
      err = integrateRaysDevice(m, d_rays + half * 6, (n_rays - half) * 2, d_intensities ? d_intensities + half : nullptr,
                                d_timestamps ? d_timestamps + half : nullptr, ray_flags, &done_b,
                                d_filter_flags ? d_filter_flags + half : nullptr);
      if (err == OHMHIP_OK)
      {
        // the call's statistics cover both halves
        m->stats.rays_in += stats_a.rays_in;
        m->stats.rays_integrated += stats_a.rays_integrated;
        m->stats.voxel_visits += stats_a.voxel_visits;
        m->stats.ray_region_segments += stats_a.ray_region_segments;
        m->stats.regions_touched = std::max(m->stats.regions_touched, stats_a.regions_touched);
      }
    }
    if (integrated)
    {
      // What the first half integrated STAYS integrated if the second fails (include/ohmhip.h): the count says how many
      // leading elements of the caller's arrays must not be presented again.
      *integrated = done_a + done_b;
    }
    return err;
  }
  if (err == OHMHIP_OK && integrated)
  {
    *integrated = size_t(m->stats.rays_integrated) * 2;
  }
  return err;
}
}  // namespace

extern "C" {

int ohmhip_map_integrate_rays_device(ohmhip_map_t m, const double *d_rays, size_t element_count,
                                     const float *d_intensities, const double *d_timestamps, unsigned ray_flags,
                                     size_t *integrated)
try
{
  if (integrated)
  {
    *integrated = 0;
  }
  const size_t n_rays = element_count / 2;
  // Small device-pointer batches (the f4 pipeline: GpuTransformSamples output presented 4096 rays at a time) are
  // collected like small host batches: copied device to device behind the rays already waiting in the filling slot and
  // run as one device batch once coalesce_min_rays have accumulated, or as soon as anything observes the map.  The
  // call's own count of integrated rays comes from a one-workgroup pass of the map's ray filter over its staged rays.
  const bool defer = m && d_rays && n_rays > 0 && m->coalesce_min_rays > 0 && n_rays < m->coalesce_min_rays &&
                     !m->layers[OHMHIP_LID_TRAVERSAL] && !m->spill_enabled;
  if (!defer)
  {
    OHMHIP_SETTLE(m);  // batches presented earlier come first
    return integrateRaysDevice(m, d_rays, element_count, d_intensities, d_timestamps, ray_flags, integrated);
  }
  OHMHIP_CHECK(validateBatchRequest(m, ray_flags));
  if (m->pending_rays &&
      (!m->pending_on_device || m->pending_flags != ray_flags || m->pending_intens != (d_intensities != nullptr) ||
       m->pending_times != (d_timestamps != nullptr) || m->pending_fflags))
  {
    OHMHIP_SETTLE(m);
  }
  ohmhip_map_s::RaySlot &sl = m->ray_slots[m->fill_slot];
  if (m->pending_rays == 0)
  {
    if (sl.in_flight)
    {
      OHMHIP_CHECK(hipEventSynchronize(sl.done));  // the batch before last still owns this slot's buffers
      sl.in_flight = false;
    }
    // room for every call up to the flush (DevBuf::ensure does not keep contents: sized before the first append)
    const size_t cap = 2 * m->coalesce_min_rays;
    OHMHIP_CHECK(sl.d_rays.ensure(cap * 48, false, m->stream));
    if (d_timestamps)
    {
      OHMHIP_CHECK(sl.d_times.ensure(cap * 8, false, m->stream));
    }
    if (d_intensities)
    {
      OHMHIP_CHECK(sl.d_intens.ensure(cap * 4, false, m->stream));
    }
  }
  hipStream_t cs = m->copy_stream;
  double *staged = static_cast<double *>(sl.d_rays.ptr) + m->pending_rays * 6;
  OHMHIP_CHECK(hipMemcpyAsync(staged, d_rays, n_rays * 48, hipMemcpyDeviceToDevice, cs));
  if (d_timestamps)
  {
    OHMHIP_CHECK(hipMemcpyAsync(static_cast<double *>(sl.d_times.ptr) + m->pending_rays, d_timestamps, n_rays * 8,
                                hipMemcpyDeviceToDevice, cs));
    if (m->first_ray_time < 0)
    {
      double first = 0;  // OccupancyMap::updateFirstRayTime(*timestamps) (ohm/OccupancyMap.cpp:343-347)
      OHMHIP_CHECK(hipMemcpyAsync(&first, d_timestamps, sizeof(double), hipMemcpyDeviceToHost, cs));
      OHMHIP_CHECK(hipStreamSynchronize(cs));
      m->first_ray_time = first;
    }
  }
  if (d_intensities)
  {
    OHMHIP_CHECK(hipMemcpyAsync(static_cast<float *>(sl.d_intens.ptr) + m->pending_rays, d_intensities, n_rays * 4,
                                hipMemcpyDeviceToDevice, cs));
  }
  if (integrated)
  {
    // (its own copy of the constants: a batch being launched on the map's thread -- ohmhip_map_set_async_launch -- points
    // m->mc.batch_filter_flags at ITS filter flags while it runs; this call's rays carry none.  ADVICE r3)
    MapConst count_mc = m->mc;
    count_mc.batch_filter_flags = nullptr;
    hipLaunchKernelGGL(k_count_passed, dim3(1), dim3(1024), 0, cs, count_mc, static_cast<const double *>(staged),
                       uint32_t(n_rays), ray_flags, m->h_passed_dev);
    OHMHIP_CHECK(hipEventRecord(m->ev_passed, cs));
    OHMHIP_CHECK(hipEventSynchronize(m->ev_passed));  // (also: the caller's arrays have been copied)
    *integrated = size_t(*m->h_passed) * 2;
  }
  m->pending_flags = ray_flags;
  m->pending_fflags = false;
  m->pending_intens = d_intensities != nullptr;
  m->pending_times = d_timestamps != nullptr;
  m->pending_on_device = true;
  m->pending_rays += n_rays;
  m->pending_calls += 1;
  if (m->pending_rays < m->coalesce_min_rays)
  {
    return OHMHIP_OK;  // deferred
  }
  const int err = flushPendingRays(m);
  if (err != OHMHIP_OK && integrated)
  {
    *integrated = 0;
  }
  return err;
}
OHMHIP_ABI_CATCH

}  // extern "C"

namespace
{
/// What a batch request must satisfy whatever its size: checked when the call is made, also for calls whose rays only
/// run later with a collected batch.
int validateBatchRequest(ohmhip_map_t m, unsigned ray_flags)
{
  // (every RayFlag combination of the CPU mappers is supported since round 3 -- on ONE map)
  switch (m->config.mode)
  {
  case OHMHIP_MODE_OCCUPANCY:
    if ((ray_flags & OHMHIP_RF_STOP_ON_FIRST_OCCUPIED) && m->mc.owner_world > 1u)
    {
      // Where a ray stops depends on every voxel before that point, also those in regions another rank owns: the one
      // flag whose effect is not local to a voxel, hence not available on a region-partitioned map.
      return OHMHIP_ERR_UNSUPPORTED;
    }
    return m->layers[OHMHIP_LID_OCCUPANCY] ? OHMHIP_OK : OHMHIP_ERR_INVALID_ARG;
  case OHMHIP_MODE_NDT_OM:
  case OHMHIP_MODE_NDT_TM:
    if (!m->layers[OHMHIP_LID_OCCUPANCY] || !m->layers[OHMHIP_LID_MEAN] || !m->layers[OHMHIP_LID_COVARIANCE])
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    if (m->config.mode == OHMHIP_MODE_NDT_TM && (!m->layers[OHMHIP_LID_INTENSITY] || !m->layers[OHMHIP_LID_HIT_MISS]))
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    return OHMHIP_OK;
  case OHMHIP_MODE_TSDF:
    if (!m->layers[OHMHIP_LID_TSDF])
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    return OHMHIP_OK;
  default:
    return OHMHIP_ERR_UNSUPPORTED;
  }
}

/// Rays of a host batch the map's ray filter accepts: what the device counts as integrated (k_ray_setup, kRwPassed),
/// computed on the host with the same arithmetic (walk_device.h: filterRay) so that a call can report it without
/// waiting for the device -- or for a batch that has not even been launched yet.
size_t hostFilterCount(const MapConst &mc, const double *rays, size_t n_rays, bool caller_filtered)
{
  if (caller_filtered || mc.filter_mode == OHMHIP_FILTER_NONE)
  {
    return n_rays;
  }
  size_t passed = 0;
  for (size_t i = 0; i < n_rays; ++i)
  {
    const double *r = rays + 6 * i;
    bool good = std::isfinite(r[0]) && std::isfinite(r[1]) && std::isfinite(r[2]) && std::isfinite(r[3]) &&
                std::isfinite(r[4]) && std::isfinite(r[5]);
    if (mc.filter_mode == OHMHIP_FILTER_GOOD)
    {
      const double rx = r[3] - r[0];
      const double ry = r[4] - r[1];
      const double rz = r[5] - r[2];
      const double len2 = (rx * rx + ry * ry) + rz * rz;
      good = good && (mc.filter_range <= 0 || len2 <= mc.filter_range * mc.filter_range);
    }
    passed += good ? 1u : 0u;
  }
  return passed;
}

/// Copy rays and count those of at most `range` length in one loop (the map's default filter, goodRay with a range).
/// A ray with a non-finite coordinate has a NaN or infinite squared length, which fails the comparison against the
/// finite range^2 -- the test for finite coordinates of hostFilterCount is implied and the loop stays at copy speed.
size_t copyRaysCountInRange(double *dst, const double *rays, size_t n_rays, double range2)
{
  size_t passed = 0;
  for (size_t i = 0; i < n_rays; ++i)
  {
    const double *r = rays + 6 * i;
    double *d = dst + 6 * i;
    const double x0 = r[0], y0 = r[1], z0 = r[2], x1 = r[3], y1 = r[4], z1 = r[5];
    d[0] = x0;
    d[1] = y0;
    d[2] = z0;
    d[3] = x1;
    d[4] = y1;
    d[5] = z1;
    const double rx = x1 - x0;
    const double ry = y1 - y0;
    const double rz = z1 - z0;
    const double len2 = (rx * rx + ry * ry) + rz * rz;
    passed += (len2 <= range2) ? 1u : 0u;
  }
  return passed;
}

/// Copy rays [first, last) of a host block into the pinned block at `dst` and count the rays the filter accepts in the
/// same sweep: the range is cut into pieces that stay in the core's L2 between the copy and the count.
size_t stageRayRange(const MapConst &mc, char *dst, const double *rays, size_t first, size_t last, bool caller_filtered)
{
  static constexpr size_t kPiece = 4096;  // rays per copy+count piece (192 KiB)
  const double range2 = mc.filter_range * mc.filter_range;
  if (!caller_filtered && mc.filter_mode == OHMHIP_FILTER_GOOD && mc.filter_range > 0 && std::isfinite(range2))
  {
    return copyRaysCountInRange(reinterpret_cast<double *>(dst + first * 48), rays + first * 6, last - first, range2);
  }
  size_t passed = 0;
  for (size_t at = first; at < last; at += kPiece)
  {
    const size_t n = std::min<size_t>(kPiece, last - at);
    std::memcpy(dst + at * 48, rays + at * 6, n * 48);
    passed += hostFilterCount(mc, rays + at * 6, n, caller_filtered);
  }
  return passed;
}

constexpr unsigned kStageThreads = 8;                  // pool threads of a map (one core copies ~10 GB/s; PCIe Gen5 takes ~55)
constexpr size_t kStagePerThread = size_t(1) << 16;    // rays before another thread is worth waking
constexpr size_t kUploadPiece = size_t(1) << 15;       // rays per host-to-device copy of a staged block (1.5 MiB)

StagePool &stagePool(ohmhip_map_t m)
{
  if (!m->stage_pool)
  {
    m->stage_pool.reset(new StagePool(kStageThreads));
  }
  return *m->stage_pool;
}

/// Stage a host ray block into the pinned slot and count the rays the filter accepts; large blocks are shared between
/// the map's pool threads and the caller.
size_t stageRaysAndCount(ohmhip_map_t m, char *dst, const double *rays, size_t n_rays, bool caller_filtered)
{
  const MapConst &mc = m->mc;
  const unsigned n_workers = unsigned(std::min<size_t>(kStageThreads, n_rays / kStagePerThread));
  if (n_workers <= 1)
  {
    return stageRayRange(mc, dst, rays, 0, n_rays, caller_filtered);
  }
  const size_t n_pieces = (n_rays + kUploadPiece - 1) / kUploadPiece;
  std::atomic<size_t> next(0), passed(0);
  auto work = [&](unsigned) {
    for (size_t p = next.fetch_add(1); p < n_pieces; p = next.fetch_add(1))
    {
      passed.fetch_add(stageRayRange(mc, dst, rays, p * kUploadPiece, std::min(n_rays, (p + 1) * kUploadPiece),
                                     caller_filtered));
    }
  };
  StagePool &pool = stagePool(m);
  pool.start(n_workers - 1, work);
  work(0);
  pool.wait();
  return passed.load();
}

/// The same for a block that is a device batch on its own, with the rays' host-to-device copies queued piece by piece
/// as the pieces are staged: the PCIe transfer runs beside the staging of the rest, not after it (a 1 M-ray call was
/// stage 1.2 ms, then copy 1.0 ms; the flush that follows finds RaySlot::rays_uploaded set).
int stageRaysAndUpload(ohmhip_map_t m, ohmhip_map_s::RaySlot &sl, const double *rays, size_t n_rays,
                       bool caller_filtered, size_t *passed_out)
{
  const MapConst &mc = m->mc;
  OHMHIP_CHECK(sl.d_rays.ensure(n_rays * 48, false, m->stream));
  char *dst = slotRays(sl);
  char *d_dst = static_cast<char *>(sl.d_rays.ptr);
  const size_t n_pieces = (n_rays + kUploadPiece - 1) / kUploadPiece;
  std::unique_ptr<std::atomic<unsigned char>[]> done(new std::atomic<unsigned char>[n_pieces]);
  for (size_t p = 0; p < n_pieces; ++p)
  {
    done[p].store(0, std::memory_order_relaxed);
  }
  std::atomic<size_t> next(0), passed(0);
  auto stage_piece = [&](size_t p) {
    passed.fetch_add(stageRayRange(mc, dst, rays, p * kUploadPiece, std::min(n_rays, (p + 1) * kUploadPiece),
                                   caller_filtered));
    done[p].store(1, std::memory_order_release);
  };
  auto work = [&](unsigned) {
    for (size_t p = next.fetch_add(1); p < n_pieces; p = next.fetch_add(1))
    {
      stage_piece(p);
    }
  };
  StagePool &pool = stagePool(m);
  pool.start(unsigned(std::min<size_t>(kStageThreads, std::max<size_t>(1, n_pieces / 2))), work);
  // The caller sends what is staged, in order, kCopyPieces pieces per copy (a copy call costs ~10 us: 1.5 MiB copies
  // reach 42 GB/s, 6 MiB and more 55; scripts/probes/h2d_probe.hip), and stages pieces itself while it waits.
  constexpr size_t kCopyPieces = 4;
  hipError_t copy_err = hipSuccess;
  for (size_t p = 0; p < n_pieces;)
  {
    const size_t want = std::min(n_pieces, p + kCopyPieces);
    size_t e = p;
    while (e < want && done[e].load(std::memory_order_acquire))
    {
      ++e;
    }
    if (e < want)
    {
      const size_t q = next.fetch_add(1);
      if (q < n_pieces)
      {
        stage_piece(q);
      }
      else
      {
        while (!done[e].load(std::memory_order_acquire))
        {
          std::this_thread::yield();
        }
      }
      continue;
    }
    while (e < n_pieces && e - p < 2 * kCopyPieces && done[e].load(std::memory_order_acquire))
    {
      ++e;
    }
    const size_t first = p * kUploadPiece, last = std::min(n_rays, e * kUploadPiece);
    if (copy_err == hipSuccess)
    {
      copy_err = hipMemcpyAsync(d_dst + first * 48, dst + first * 48, (last - first) * 48, hipMemcpyHostToDevice,
                                m->copy_stream);
    }
    p = e;
  }
  pool.wait();  // (the workers hold references to this frame)
  OHMHIP_CHECK(copy_err);
  sl.rays_uploaded = true;
  *passed_out = passed.load();
  return OHMHIP_OK;
}

int integrateRaysHost(ohmhip_map_t m, const double *rays, size_t element_count, const float *intensities,
                      const double *timestamps, unsigned ray_flags, const unsigned char *filter_flags,
                      size_t *integrated)
{
  if (integrated)
  {
    *integrated = 0;
  }
  if (!m || (!rays && element_count))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_CHECK(validateBatchRequest(m, ray_flags));
  const size_t n_rays = element_count / 2;
  if (n_rays == 0)
  {
    return OHMHIP_OK;
  }
  if (n_rays >= (size_t(1) << (kHitRayBits - 1)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  // Batches only share a device launch when they are integrated the same way; the traversal layer's exit range is
  // carried from ray to ray WITHIN one call (secondary_device.h: lastExitRange), so its batches are never merged.
  const bool coalesce = m->coalesce_min_rays > 0 && !m->layers[OHMHIP_LID_TRAVERSAL];
  if (m->pending_rays &&
      (!coalesce || m->pending_on_device || m->pending_flags != ray_flags || m->pending_intens != (intensities != nullptr) ||
       m->pending_times != (timestamps != nullptr) || m->pending_fflags != (filter_flags != nullptr) ||
       m->pending_rays + n_rays >= (size_t(1) << (kHitRayBits - 1))))
  {
    const int err = flushPendingRays(m);
    if (err != OHMHIP_OK)
    {
      return err;
    }
  }
  ohmhip_map_s::RaySlot &sl = m->ray_slots[m->fill_slot];
  if (m->pending_rays == 0 && sl.in_flight)
  {
    OHMHIP_CHECK(hipEventSynchronize(sl.done));  // the batch before last still owns this slot's buffers
    sl.in_flight = false;
  }
  int err = growRaySlot(m, sl, m->pending_rays + n_rays);
  if (err != OHMHIP_OK)
  {
    return err;
  }
  // A call that is a device batch on its own gets the count from the device (k_ray_setup counts what its filter passes
  // and the batch summary reaches the host inside this call anyway); calls that share a batch are counted here.
  const bool own_batch = m->pending_rays == 0 && (!coalesce || n_rays >= m->coalesce_min_rays);
  const bool device_counts = own_batch && !m->async_launch;  // (a call that hands its batch to the launch thread counts here)
  size_t passed = 0;
  const auto t_stage = std::chrono::steady_clock::now();
  if (own_batch && n_rays >= 4 * kUploadPiece)
  {
    OHMHIP_CHECK(stageRaysAndUpload(m, sl, rays, n_rays, filter_flags != nullptr || device_counts, &passed));
  }
  else
  {
    passed = stageRaysAndCount(m, slotRays(sl) + m->pending_rays * 48, rays, n_rays,
                               filter_flags != nullptr || device_counts);
  }
  if (timestamps)
  {
    std::memcpy(slotTimes(sl) + m->pending_rays * 8, timestamps, n_rays * 8);
    if (m->first_ray_time < 0)
    {
      m->first_ray_time = timestamps[0];  // OccupancyMap::updateFirstRayTime (ohm/OccupancyMap.cpp:343-347)
    }
  }
  if (intensities)
  {
    std::memcpy(slotIntens(sl) + m->pending_rays * 4, intensities, n_rays * 4);
  }
  if (filter_flags)
  {
    std::memcpy(slotFilterFlags(sl) + m->pending_rays, filter_flags, n_rays);
  }
  m->pending_flags = ray_flags;
  m->pending_fflags = filter_flags != nullptr;
  m->pending_intens = intensities != nullptr;
  m->pending_times = timestamps != nullptr;
  m->pending_rays += n_rays;
  m->pending_calls += 1;
  if (integrated)
  {
    *integrated = 2 * passed;
  }
  if (coalesce && m->pending_rays < m->coalesce_min_rays)
  {
    return OHMHIP_OK;  // deferred: runs with the following calls' rays, or as soon as anything observes the map
  }
  const auto t_flush = std::chrono::steady_clock::now();
  err = flushPendingRays(m, device_counts ? integrated : nullptr, true);
  if (m->debug_flags & 2048u)
  {
    const auto t_end = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ohmhip dbg] host batch of %zu rays: staged (+ upload queued) %.3f ms, launched %.3f ms\n", n_rays,
                 std::chrono::duration<double, std::milli>(t_flush - t_stage).count(),
                 std::chrono::duration<double, std::milli>(t_end - t_flush).count());
  }
  if (err != OHMHIP_OK && integrated)
  {
    *integrated = 0;
  }
  return err;
}
}  // namespace


#endif  // OHMHIP_RAY_ENTRY_H
