// batch_run.h -- one integrate call on the device: `struct BatchRun` (set-up + plan, exhaustion handling, binning, walk,
// apply / replay) and integrateBatch (GpuMap::integrateRays / enqueueRegions / finaliseBatch, ohmgpu/GpuMap.cpp:730-1224).
//
// Part of ohmhip_map.hip's translation unit (included there, in order): not a stand-alone header.
#ifndef OHMHIP_BATCH_RUN_H
#define OHMHIP_BATCH_RUN_H

namespace
{
/// One integrate call on the device.  The members are what the phases of a batch share; the phases, in the order
/// integrateBatch runs them:
///   prepare()            launch shapes, map mode, per-batch buffers                      (once)
///   frontHalf()          set-up pass + plan on the front stream, speculative bin + sort, wait for the plan's summary
///   resolveExhaustion()  pool / chunk list full: roll back, grow or spill, ask for a retry (or fail, map untouched)
///   commitRegions()      cache statistics, content of re-admitted regions, undo a wrong speculation
///   sizeBuffers()        segment buffer, event / key buffers
///   binAndOrder()        binning pass and sample ordering (unless the speculative launches stand)
///   walk()               k_region_walk (+ the re-walk when an NDT / TSDF event list overflowed)
///   applyOccupancy() / replayEvents()   ordered replay and count application
///   finish()             events, statistics
struct BatchRun
{
  ohmhip_map_t m;
  const double *d_rays;
  const float *d_intensities;
  const double *d_timestamps;
  uint32_t n_rays;
  unsigned ray_flags;
  hipStream_t s, f;
  hipEvent_t *tev;
  uint32_t ring;  ///< this batch's entry of the timing ring
  // decided once per call
  uint32_t next_info_index = 0;
  bool info_clean = false;
  uint32_t ray_blocks = 0, bin_rays_per_block = 0, bin_threads = 0, bin_blocks = 0, bin_tab_mask = 0;
  uint32_t batch_chunk_segments = 0;
  int mode = 0;
  bool stop_mode = false, occupancy_mode = false, ndt_mode = false, tsdf_mode = false;
  /// Occupancy map without mean / secondary layers: the walk kernel replays the samples of every region it holds in one
  /// chunk itself, and the apply kernels run over k_plan's lists of the regions that are left.
  bool occ_inline = false;
  int ray_shift = 0;
  SecondaryLayers sec;
  // per attempt
  int attempt = 0;
  uint32_t spec_seg_cap = 0, seg_cap = 0;
  bool speculated = false, bucket_hits = false;
  BatchInfo info;
  unsigned long long *keys_a = nullptr, *keys_b = nullptr, *events = nullptr;
  const unsigned long long *sorted = nullptr;
  uint32_t event_capacity = 0, n_events = 0;
  float *direct_occ = nullptr;
  uint32_t direct_segments = 0;
  uint32_t sort_covers = 0;  ///< the per-region sort launched so far orders regions of up to this many samples
  bool batch_end_marked = false;  ///< tev[4] is the stop event of the batch's last kernel already
  /// NDT / TSDF: the event sort and the replay were launched on a speculated event count (the previous batch's plus head
  /// room; k_pad_events); the true count is on its way to the host and is looked at once the replay has been launched.
  bool events_speculated = false;
  uint32_t spec_events = 0;

  /// Samples of a region the walk kernel's shape stages in LDS (WalkFull / WalkHalf).
  uint32_t walkLdsHits() const { return uint32_t(m->walk_half ? WalkHalf::kLdsHits : WalkFull::kLdsHits); }

  /// The batch recorded (or bound to a kernel) event k of its ring entry.
  void mark(int k) { m->tev_mask[ring] = uint8_t(m->tev_mask[ring] | (1u << k)); }

  int prepare()
  {
    // This batch's summary block: the next of the three, zeroed by the previous batch's k_plan if that ran.  (Three: the
    // set-up pass of this batch runs under the previous batch's apply kernels, which still read theirs, and zeroes the
    // following batch's.)
    m->tev_mask[ring] = 0;
    m->tev_pre_walk[ring] = 0;
    m->info_index = (m->info_index + 1u) % 3u;
    next_info_index = (m->info_index + 1u) % 3u;
    info_clean = m->info_clean;
    m->info_clean = false;
    // The other copy of the doubled per-batch scratch.
    m->parity ^= 1u;
    ray_blocks = (n_rays + 255) / 256;
    // Binning launch shape: 1024 rays per 512-thread workgroup for large batches; small batches use smaller workgroups
    // with as many rays as threads so they still cover the CUs.
    bin_rays_per_block = m->bin_rays_per_block;
    bin_threads = kBinThreads;
    while (bin_rays_per_block > 128 && n_rays / bin_rays_per_block < 2 * m->walk_workgroups)
    {
      bin_rays_per_block /= 2;
    }
    bin_threads = std::min<uint32_t>(bin_threads, bin_rays_per_block);
    bin_blocks = (n_rays + bin_rays_per_block - 1) / bin_rays_per_block;
    // LDS region table of the binning workgroups: two entries per ray of the workgroup, at most kLtabSize.
    bin_tab_mask = std::min<uint32_t>(kLtabSize, std::max<uint32_t>(256u, 2u * bin_rays_per_block)) - 1u;
    // Chunk size of this batch: small batches get smaller chunks so the walk still has a few chunks per CU (estimated
    // from the previous batch's segments per ray; results do not depend on it).
    const uint64_t expected_segments = uint64_t(double(n_rays) * m->segments_per_ray);
    batch_chunk_segments = m->chunk_segments;
    // (the floor: 2048 segments, two rounds of the walk workgroup's lanes -- below that a chunk's fixed cost dominates; the
    // batches of a sensor driver's 4096-ray calls are latency bound from end to end and gain 9 % from 512-segment chunks
    // on more CUs, measured in round 6: 123 -> 112 us per call, while 65 536-ray batches LOSE 20 % with that floor)
    const uint32_t chunk_floor = (n_rays <= 16384u) ? std::min<uint32_t>(m->min_chunk_segments, 512u) : m->min_chunk_segments;
    while (batch_chunk_segments > chunk_floor &&
           expected_segments / batch_chunk_segments < 3ull * m->walkSlots())
    {
      batch_chunk_segments /= 2;
    }
    mode = m->config.mode;
    // kRfStopOnFirstOccupied: no counting shortcut exists (replay_kernels.h, k_stop_replay): such a batch takes the
    // general event route of NDT / TSDF -- every visit an event, sorted per voxel -- with its own replay.
    stop_mode = mode == OHMHIP_MODE_OCCUPANCY && (ray_flags & OHMHIP_RF_STOP_ON_FIRST_OCCUPIED) != 0;
    occupancy_mode = mode == OHMHIP_MODE_OCCUPANCY && !stop_mode;
    ndt_mode = mode == OHMHIP_MODE_NDT_OM || mode == OHMHIP_MODE_NDT_TM;
    tsdf_mode = mode == OHMHIP_MODE_TSDF;
    if (ndt_mode)
    {
      // RayMapperNdt honours only kRfEndPointAsFree / kRfExcludeOrigin / kRfExcludeRay (ohm/RayMapperNdt.cpp:238-262).
      ray_flags &= (OHMHIP_RF_END_POINT_AS_FREE | OHMHIP_RF_EXCLUDE_ORIGIN | OHMHIP_RF_EXCLUDE_RAY);
    }
    if (tsdf_mode)
    {
      // RayMapperTsdf ignores the flags and walks start..end inclusive (ohm/RayMapperTsdf.cpp:87-88, 176).
      ray_flags = OHMHIP_RF_END_POINT_AS_FREE;
    }
    ray_shift = occupancy_mode ? 0 : kEvRayShift;
    sec.traversal = tsdf_mode ? nullptr : static_cast<float *>(m->layers[OHMHIP_LID_TRAVERSAL]);
    sec.touch_time = tsdf_mode ? nullptr : static_cast<uint32_t *>(m->layers[OHMHIP_LID_TOUCH_TIME]);
    sec.incident = tsdf_mode ? nullptr : static_cast<uint32_t *>(m->layers[OHMHIP_LID_INCIDENT]);
    sec.timestamps = d_timestamps;
    sec.time_base = m->first_ray_time;
    occ_inline = occupancy_mode && !m->layers[OHMHIP_LID_MEAN] && !sec.traversal && !sec.touch_time && !sec.incident &&
                 !m->layers[OHMHIP_LID_INTENSITY] && !m->layers[OHMHIP_LID_HIT_MISS];

    for (int p = 0; p < 2; ++p)
    {
      // (both parities at once: the next batch's copies would otherwise be allocated -- and the stream drained -- in the
      // middle of a run of batches)
      OHMHIP_CHECK(m->walks_buf[p].ensure(sizeof(RayWalk) * size_t(n_rays), false, s));
      OHMHIP_CHECK(m->wg_regions[p].ensure(sizeof(WgRegion) * size_t(bin_blocks) * kLtabSize, false, s));
      OHMHIP_CHECK(m->wg_region_count[p].ensure(sizeof(uint32_t) * size_t(bin_blocks), false, s));
    }
    if (occupancy_mode)
    {
      OHMHIP_CHECK(m->hit_keys_a.ensure(sizeof(unsigned long long) * size_t(n_rays), false, s));
      OHMHIP_CHECK(m->hit_keys_b.ensure(sizeof(unsigned long long) * size_t(n_rays), false, s));
      OHMHIP_CHECK(m->interval_counts.ensure(sizeof(uint32_t) * size_t(n_rays), true, s));
      size_t sort_bytes = 0;
      OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(nullptr, sort_bytes, static_cast<unsigned long long *>(m->hit_keys_a.ptr),
                                            static_cast<unsigned long long *>(m->hit_keys_b.ptr), size_t(n_rays),
                                            kHitRayBits, sortEndBit(m), s));
      OHMHIP_CHECK(m->sort_temp.ensure(sort_bytes, false, s));
    }
    return OHMHIP_OK;
  }

  void launchBin(bool bucket, uint32_t seg_capacity, unsigned long long *hit_keys)
  {
    // (small batches -- 128-ray workgroups -- run the instantiation with the small LDS table: more workgroups per CU)
    // tev[1], the kernel's own stop event, is what the next batch's set-up pass waits for (see map_state.h: a stop event
    // costs nothing, an event record behind the kernel would idle the queue for microseconds).
    if (bin_tab_mask < kLtabSmall)
    {
      hipExtLaunchKernelGGL(k_ray_bin<kLtabSmall>, dim3(bin_blocks), dim3(bin_threads), 0, s, nullptr, tev[1], 0, m->mc,
                            regionTable(m), batchScratch(m), static_cast<const RayWalk *>(batchWalks(m).ptr), n_rays,
                            static_cast<Segment *>(m->segments.ptr), seg_capacity, hit_keys, m->d_hit_mask, ray_shift,
                            bucket ? 1 : 0, bin_rays_per_block, bin_tab_mask);
    }
    else
    {
      hipExtLaunchKernelGGL(k_ray_bin<kLtabSize>, dim3(bin_blocks), dim3(bin_threads), 0, s, nullptr, tev[1], 0, m->mc,
                            regionTable(m), batchScratch(m), static_cast<const RayWalk *>(batchWalks(m).ptr), n_rays,
                            static_cast<Segment *>(m->segments.ptr), seg_capacity, hit_keys, m->d_hit_mask, ray_shift,
                            bucket ? 1 : 0, bin_rays_per_block, bin_tab_mask);
    }
    mark(1);
    m->tev_pre_walk[ring] = 1;
    m->bin_done_event = tev[1];
  }

  /// Order the regions' sample lists.  Small regions (nearly all) by 256-thread workgroups, the dense ones by the
  /// 1024-thread, 66 KiB instantiation -- launched only when some region needs it, and FIRST: it starts the moment the
  /// binning pass ends, ahead of the next batch's set-up pass (which reaches the device ~15 us later through its event),
  /// whereas launched second its workgroups queued behind that pass for a CU's LDS and held the walk back by 50-90 us.
  /// `dense` = the densest region's sample count (the previous batch's when the launch is speculative: binAndOrder adds
  /// the dense launch once the batch's own summary asks for it).  tev[2], the end of the ordering phase, is the stop event
  /// of the phase's last launch.
  void launchRegionSort(uint32_t dense)
  {
    const unsigned long long *in = static_cast<const unsigned long long *>(m->hit_keys_a.ptr);
    unsigned long long *out = static_cast<unsigned long long *>(m->hit_keys_b.ptr);
    const bool need_dense = kSortSmallHits == 0 || dense > kSortSmallHits;
    if (need_dense)
    {
      hipExtLaunchKernelGGL((k_sort_region_hits<kSortRegionHits, kSortThreads>), dim3(2 * m->walk_workgroups),
                            dim3(kSortThreads), 0, s, nullptr, kSortSmallHits ? nullptr : tev[2], 0, regionTable(m),
                            batchScratch(m), in, out, m->mc.region_voxels, kSortSmallHits);
    }
    if (kSortSmallHits)
    {
      hipExtLaunchKernelGGL((k_sort_region_hits<kSortSmallHits ? kSortSmallHits : 64u, kSortSmallThreads>),
                            dim3(8 * m->walk_workgroups), dim3(kSortSmallThreads), 0, s, nullptr, tev[2], 0,
                            regionTable(m), batchScratch(m), in, out, m->mc.region_voxels, 0u);
    }
    sort_covers = need_dense ? kSortRegionHits : kSortSmallHits;
    mark(2);
    m->tev_pre_walk[ring] = 2;
  }

  void launchDenseSort()
  {
    hipExtLaunchKernelGGL((k_sort_region_hits<kSortRegionHits, kSortThreads>), dim3(2 * m->walk_workgroups),
                          dim3(kSortThreads), 0, s, nullptr, tev[2], 0, regionTable(m), batchScratch(m),
                          static_cast<const unsigned long long *>(m->hit_keys_a.ptr),
                          static_cast<unsigned long long *>(m->hit_keys_b.ptr), m->mc.region_voxels, kSortSmallHits);
    sort_covers = kSortRegionHits;
  }

  int frontHalf()
  {
    // The set-up pass goes to the front stream.  It has to wait for the batch that last used this parity's scratch
    // copy, RayWalk array and workgroup region lists -- the batch before the previous one.  It is also held back until
    // the previous batch's binning pass is done: beside that pass it would only compete for the vector ALUs (measured:
    // no gain), whereas started then k_ray_setup runs beside the previous batch's sample sort (LDS bound, few
    // registers) and k_plan -- one workgroup -- queues behind the persistent walk kernel and runs on the first CU that
    // kernel vacates (C1: 1.06 -> 1.03 ms per batch; holding the pass until the walk has ended loses the gain again,
    // and so does a stream priority above the compute stream's).  In a kernel trace k_plan therefore shows the walk's
    // duration: its dispatch waits for a CU.
    if (m->batch_done_event[m->parity])
    {
      OHMHIP_CHECK(hipStreamWaitEvent(f, m->batch_done_event[m->parity], 0));
    }
    if (m->bin_done_event)
    {
      // (round 5, with the event markers gone: letting the pass start with the previous batch's binning pass instead of
      // behind it was measured again -- 0.951 against 0.883 ms per C1 batch: the two slow each other down by more than
      // the ~15 us the event hand-over costs)
      OHMHIP_CHECK(hipStreamWaitEvent(f, m->bin_done_event, 0));
    }
    if (attempt > 0 || !info_clean)
    {
      OHMHIP_CHECK(hipMemsetAsync(m->d_info + m->info_index, 0, sizeof(BatchInfo), f));
    }
    if (m->phase_timing)
    {
      OHMHIP_CHECK(hipEventRecord(tev[0], f));
      mark(0);
    }
    if (bin_tab_mask < kLtabSmall)
    {
      hipLaunchKernelGGL(k_ray_setup<kLtabSmall>, dim3(bin_blocks), dim3(bin_threads), 0, f, m->mc, regionTable(m),
                         batchScratch(m), d_rays, n_rays, ray_flags, static_cast<RayWalk *>(batchWalks(m).ptr),
                         bin_rays_per_block, bin_tab_mask);
    }
    else
    {
      hipLaunchKernelGGL(k_ray_setup<kLtabSize>, dim3(bin_blocks), dim3(bin_threads), 0, f, m->mc, regionTable(m),
                         batchScratch(m), d_rays, n_rays, ray_flags, static_cast<RayWalk *>(batchWalks(m).ptr),
                         bin_rays_per_block, bin_tab_mask);
    }
    // The plan's stop event, tev[5], is what the compute stream and the host wait for.  (Spill to host: the regions' use
    // stamps reach the host with the summary -- a copy queued behind the plan, so there the wait is on an event
    // recorded behind that copy.)
    hipExtLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, f, nullptr, tev[5], 0, regionTable(m), batchScratch(m),
                          batchChunks(m), m->chunk_capacity, batch_chunk_segments, m->h_info_dev,
                          m->d_info + next_info_index, batchEventCount(m), occ_inline ? walkLdsHits() : 0u);
    mark(5);
    m->info_clean = true;
    hipEvent_t plan_done = tev[5];
    if (m->spill_enabled)
    {
      OHMHIP_CHECK(queueUseStamps(m, f));
      OHMHIP_CHECK(hipEventRecord(m->ev[7], f));
      plan_done = m->ev[7];
    }
    OHMHIP_CHECK(hipStreamWaitEvent(s, plan_done, 0));
    if (m->phase_timing)
    {
      OHMHIP_CHECK(hipEventRecord(tev[6], s));
      mark(6);
    }
    // The host needs the batch summary (segment count, sample distribution, pool state) before it can size and launch
    // the rest -- a round trip during which the device would idle.  In steady state (occupancy, previous batch sorted
    // its samples per region) the binning and the sample sort are launched right away with the buffers of the previous
    // batch; the summary then only confirms the guess, and a wrong guess costs a repeat of the two passes.
    spec_seg_cap = uint32_t(std::min<size_t>(m->segments.bytes / sizeof(Segment), 0xffffffffu));
    speculated = occupancy_mode && m->spec_bucket_ok && attempt == 0 && spec_seg_cap > 0;
    if (speculated)
    {
      launchBin(true, spec_seg_cap, static_cast<unsigned long long *>(m->hit_keys_a.ptr));
      launchRegionSort(m->spec_max_region_hits);
    }
    OHMHIP_CHECK(hipEventSynchronize(plan_done));
    OHMHIP_CHECK(hipGetLastError());
    info = *m->h_info;
    return OHMHIP_OK;
  }

  bool exhausted() const
  {
    return (info.error & (kErrSlotsFull | kErrHashFull)) || info.n_slots > m->slot_capacity ||
           info.n_chunks > m->chunk_capacity;
  }

  /// Pool / chunk list exhausted.  Returns an error when the batch fails (the map is as it was before the call) and
  /// OHMHIP_OK with `retry` set when the attempt is to be repeated.
  int resolveExhaustion(bool &retry)
  {
    retry = false;
    // Pool exhausted: forget what this batch inserted, grow, retry.
    OHMHIP_CHECK(hipStreamSynchronize(s));
    m->spec_bucket_ok = false;
    const int err = rollbackAndGrow(m, info.n_slots);
    if (err)
    {
      // The pool may not grow (memory limit / device memory / slot field): forget what the batch inserted.
      const int rollback_err = rollbackTable(m);
      if (rollback_err)
      {
        return rollback_err;
      }
      if (err == OHMHIP_ERR_CAPACITY && m->spill_enabled && !(info.error & kErrHashFull) &&
          info.n_slots > m->slots_committed)
      {
        // Spill to host: make room by moving the least recently used regions to the host store, then repeat the
        // batch.  (The failed attempt's k_plan stamped the regions this batch touches: they go last.)
        const uint64_t per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
        const uint64_t allowed =
          m->memory_limit ? std::min<uint64_t>(m->memory_limit / per_region, kMaxRegionSlots) : m->slot_capacity;
        const uint64_t wanted = uint64_t(info.n_slots);  // committed + the batch's new regions
        // (the one cause integrateRaysDevice answers by presenting the batch in halves: the batch ALONE does not fit)
        m->batch_exceeds_limit = wanted > allowed && wanted - allowed > m->slots_committed;
        if (wanted > allowed && wanted - allowed <= m->slots_committed)
        {
          const int evict_err = evictColdRegions(m, uint32_t(wanted - allowed));
          if (evict_err == OHMHIP_OK)
          {
            retry = true;
            return OHMHIP_OK;
          }
        }
      }
      return err;  // the batch fails, the map stays as it was
    }
    retry = true;
    return OHMHIP_OK;
  }

  int commitRegions()
  {
    m->cache_misses += info.n_slots - m->slots_committed;
    m->cache_hits += info.n_touched - std::min(info.n_touched, info.n_slots - m->slots_committed);
    if (!m->spilled.empty())
    {
      // Regions this batch created that are waiting in the host store: their content comes back before the binning
      // pass (NDT / TSDF: the replay mask) and the walk see them.
      OHMHIP_CHECK(readmitSpilledSlots(m, m->slots_committed, info.n_slots));
    }
    m->slots_committed = info.n_slots;
    if (speculated && (info.n_segments > spec_seg_cap || info.max_region_hits > kSortRegionHits))
    {
      // Wrong guess (segment buffer too small, or a region too dense for the per-region sort): wait for the two
      // passes, put their cursors back and fall through to the regular launches.
      OHMHIP_CHECK(hipStreamSynchronize(s));
      if (info.n_touched)
      {
        hipLaunchKernelGGL(k_reset_cursors, dim3((info.n_touched + 255) / 256), dim3(256), 0, s, regionTable(m),
                           batchScratch(m));
      }
      speculated = false;
    }
    return OHMHIP_OK;
  }

  int sizeBuffers()
  {
    OHMHIP_CHECK(m->segments.ensure(sizeof(Segment) * size_t(std::max<uint32_t>(info.n_segments, 1u)), false, s));
    seg_cap = uint32_t(std::min<size_t>(m->segments.bytes / sizeof(Segment), 0xffffffffu));

    // Deferred-event list.  Occupancy: sized from the visit count or the previous batch's demand, with an inline
    // fallback in the kernel.  NDT / TSDF: events share one key buffer with the sample keys and are sorted together.
    uint64_t want_events =
      std::max<uint64_t>({ uint64_t(1) << 20, info.visits / 4, uint64_t(m->event_demand) * 5 / 4 });
    want_events = std::min<uint64_t>(want_events, 0xfffffff0ull - n_rays);
    if (occupancy_mode)
    {
      // The visit count is a bound, not an estimate: an occupancy walk resolves deferred misses in LDS and only regions
      // too dense for that use the list (C1: none).  A list that already holds 16 events per ray -- what
      // ohmhip_map_reserve_rays sets aside -- is therefore grown on a measured demand only, not on the bound: growing it
      // is a hipMalloc in the middle of a fresh map's first batch (0.14 ms of idle device, round 5), and a batch that
      // does overflow it falls back to resolving in place for that one batch.
      const uint64_t have = m->events.bytes / sizeof(unsigned long long);
      if (have >= std::max<uint64_t>(uint64_t(1) << 20, 16ull * n_rays))
      {
        want_events = std::max<uint64_t>(std::min(want_events, have), uint64_t(m->event_demand) * 5 / 4);
      }
    }
    if (m->event_limit)
    {
      want_events = std::min<uint64_t>(want_events, m->event_limit);  // (test knob: forces the overflow path)
    }
    keys_a = keys_b = events = nullptr;
    event_capacity = 0;
    if (occupancy_mode)
    {
      OHMHIP_CHECK(m->events.ensure(sizeof(unsigned long long) * size_t(want_events), false, s));
      keys_a = static_cast<unsigned long long *>(m->hit_keys_a.ptr);
      keys_b = static_cast<unsigned long long *>(m->hit_keys_b.ptr);
      events = static_cast<unsigned long long *>(m->events.ptr);
      event_capacity = uint32_t(std::min<size_t>(m->events.bytes / sizeof(unsigned long long), 0xfffffff0u));
    }
    else
    {
      const size_t total = size_t(n_rays) + size_t(want_events);
      OHMHIP_CHECK(m->hit_keys_a.ensure(sizeof(unsigned long long) * total, false, s));
      OHMHIP_CHECK(m->hit_keys_b.ensure(sizeof(unsigned long long) * total, false, s));
      keys_a = static_cast<unsigned long long *>(m->hit_keys_a.ptr);
      keys_b = static_cast<unsigned long long *>(m->hit_keys_b.ptr);
      events = keys_a + n_rays;
      const size_t cap_a = m->hit_keys_a.bytes / sizeof(unsigned long long) - n_rays;
      const size_t cap_b = m->hit_keys_b.bytes / sizeof(unsigned long long) - n_rays;
      event_capacity = uint32_t(std::min<size_t>(std::min(cap_a, cap_b), 0xfffffff0u - n_rays));
      if (m->event_limit)
      {
        event_capacity = std::min(event_capacity, m->event_limit);
      }
    }
    return OHMHIP_OK;
  }

  int binAndOrder()
  {
    // Occupancy: sample keys are bucketed per region and ordered by one workgroup per region in LDS, unless some
    // region holds more samples than that kernel's LDS takes (then: ray-order keys + device-wide radix sort).
    bucket_hits = occupancy_mode && info.max_region_hits <= kSortRegionHits;
    m->spec_bucket_ok = bucket_hits;
    m->spec_max_region_hits = info.max_region_hits;
    if (m->debug_flags & 4096u)
    {
      std::fprintf(stderr, "[ohmhip dbg] batch: %u rays, %u segments, %u chunks, %u regions touched, %u with samples, "
                   "densest %u samples; binned speculatively: %d\n", n_rays, info.n_segments, info.n_chunks,
                   info.n_touched, info.n_hit_regions, info.max_region_hits, int(speculated));
    }
    sorted = keys_b;
    if (speculated && info.max_region_hits > sort_covers)
    {
      launchDenseSort();  // (the speculative launch counted on the previous batch's densest region)
    }
    if (!speculated)
    {
      launchBin(bucket_hits, seg_cap, keys_a);
      if (tsdf_mode)
      {
        hipExtLaunchKernelGGL(k_tsdf_flag, dim3(ray_blocks), dim3(256), 0, s, nullptr, tev[2], 0, m->mc, regionTable(m),
                              static_cast<const RayWalk *>(batchWalks(m).ptr), d_rays, n_rays, m->d_hit_mask);
        mark(2);
        m->tev_pre_walk[ring] = 2;
      }
      if (bucket_hits)
      {
        if (info.n_hit_regions)
        {
          launchRegionSort(info.max_region_hits);
        }
      }
      else if (occupancy_mode)
      {
        size_t temp_bytes = m->sort_temp.bytes;
        // Sample keys are emitted in ray order and the radix sort is stable: sorting on the (slot, voxel) bits alone
        // leaves each voxel's samples in ray order.
        OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(m->sort_temp.ptr, temp_bytes, keys_a, keys_b, size_t(n_rays),
                                                          kHitRayBits, sortEndBit(info.n_slots), s));
        hipExtLaunchKernelGGL(k_hit_bounds, dim3(ray_blocks), dim3(256), 0, s, nullptr, tev[2], 0, sorted,
                              batchScratch(m), m->mc.region_voxels);
        mark(2);
        m->tev_pre_walk[ring] = 2;
      }
    }
    return OHMHIP_OK;
  }

  int walk(int first_attempt = 0)
  {
    // Single-chunk regions are applied by the walk kernel straight from LDS (plain log-odds misses only).
    direct_occ = (occupancy_mode || mode == OHMHIP_MODE_NDT_OM) ? static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]) :
                                                                  nullptr;
    direct_segments = (direct_occ || tsdf_mode) ? batch_chunk_segments : 0u;
    n_events = 0;
    if (info.n_chunks)
    {
      for (int walk_attempt = first_attempt; walk_attempt < 4; ++walk_attempt)
      {
        if (walk_attempt > 0)
        {
          OHMHIP_CHECK(hipMemsetAsync(batchEventCount(m), 0, 2 * sizeof(uint32_t), s));  // (k_plan zeroed them for the first)
        }
        WalkArgs wa;
        wa.mc = m->mc;
        wa.bs = batchScratch(m);
        wa.chunks = batchChunks(m);
        wa.segments = static_cast<const Segment *>(m->segments.ptr);
        wa.walks = static_cast<const RayWalk *>(batchWalks(m).ptr);
        wa.slot_keys = m->d_slot_keys;
        wa.sorted_hits = sorted;
        wa.hit_mask = m->d_hit_mask;
        wa.miss_counts = m->d_miss_counts;
        wa.interval_counts = static_cast<uint32_t *>(m->interval_counts.ptr);
        wa.events = events;
        wa.event_capacity = event_capacity;
        wa.event_count = batchEventCount(m);
        wa.refill_min_idle = m->refill_min_idle;
        wa.dbg = m->debug_flags;
        wa.ray_shift = ray_shift;
        wa.defer_all = occupancy_mode ? 0 : 1;
        wa.occupancy = direct_occ;
        wa.tsdf = tsdf_mode ? static_cast<float *>(m->layers[OHMHIP_LID_TSDF]) : nullptr;
        wa.ray_flags = ray_flags;
        const bool trace = (m->debug_flags & (16u | 64u | 128u)) != 0;
        wa.dbg_counters = trace ? m->d_dbg : nullptr;
        wa.chunk_cursor = batchEventCount(m) + 1;
        wa.n_chunks = info.n_chunks;
        // A repeated walk (NDT / TSDF event list overflow) must not apply anything twice: single-chunk regions were
        // applied straight from LDS by the first launch (the repeat only regenerates their events) and the traversal
        // layer has its sums already.
        wa.rewalk = (walk_attempt > 0 && (direct_occ || tsdf_mode)) ? 1 : 0;
        wa.flag_all = ((tsdf_mode && m->mc.tsdf_dropoff > 0) || stop_mode) ? 1 : 0;
        wa.inline_hits = occ_inline ? 1 : 0;
        // Traversal layer: its own fp64 pass over the chunk list after the count walk (traversal_kernels.h).
        const bool traversal_pass = sec.traversal != nullptr && walk_attempt == 0;
        // The lean instantiation applies unless ray origins are excluded (a first voxel that is not visited).  (An end
        // voxel that is walked -- kRfEndPointAsFree, clipped rays, TSDF -- is simply one more voxel of the ray's last
        // segment.)
        const bool special = (ray_flags & OHMHIP_RF_EXCLUDE_ORIGIN) != 0;
        const dim3 wgrid(std::min<uint32_t>(info.n_chunks, m->walkSlots()));
        const dim3 wblock(m->walk_half ? WalkHalf::kThreads : WalkFull::kThreads);
        const size_t wlds = walkLdsBytes(m->mc, m->chunk_segments, m->walk_half);
        // tev[3] -- the end of the walk phase -- is the stop event of the phase's last kernel.
        hipEvent_t walk_stop = traversal_pass ? nullptr : tev[3];
        if (m->walk_half)
        {
          if (special)
          {
            hipExtLaunchKernelGGL((k_region_walk<true, false, WalkHalf>), wgrid, wblock, wlds, s, nullptr, walk_stop, 0, wa);
          }
          else if (trace)
          {
            hipExtLaunchKernelGGL((k_region_walk<false, true, WalkHalf>), wgrid, wblock, wlds, s, nullptr, walk_stop, 0, wa);
          }
          else
          {
            hipExtLaunchKernelGGL((k_region_walk<false, false, WalkHalf>), wgrid, wblock, wlds, s, nullptr, walk_stop, 0, wa);
          }
        }
        else if (special)
        {
          hipExtLaunchKernelGGL((k_region_walk<true, false, WalkFull>), wgrid, wblock, wlds, s, nullptr, walk_stop, 0, wa);
        }
        else if (trace)
        {
          hipExtLaunchKernelGGL((k_region_walk<false, true, WalkFull>), wgrid, wblock, wlds, s, nullptr, walk_stop, 0, wa);
        }
        else
        {
          hipExtLaunchKernelGGL((k_region_walk<false, false, WalkFull>), wgrid, wblock, wlds, s, nullptr, walk_stop, 0, wa);
        }
        if (traversal_pass)
        {
          TraversalArgs ta;
          ta.mc = m->mc;
          ta.chunks = wa.chunks;
          ta.segments = wa.segments;
          ta.walks = wa.walks;
          ta.slot_keys = m->d_slot_keys;
          ta.traversal_acc = m->d_traversal_acc;
          ta.unit_bits = traversalUnitBits(m->mc.resolution);
          ta.refill_min_idle = 16;  // (8: +8 %, 32: the same, measured on C1)
          ta.chunk_cursor = batchEventCount(m) + 3;  // (k_plan zeroes it; the stop-flag replay that shares the word runs later)
          ta.n_chunks = info.n_chunks;
          hipExtLaunchKernelGGL(k_region_traversal, dim3(std::min<uint32_t>(info.n_chunks, m->walk_workgroups)),
                                dim3(kWalkThreads), traversalLdsBytes(m->mc), s, nullptr, tev[3], 0, ta);
        }
        mark(3);
        if (occupancy_mode)
        {
          // Deferred misses reach the global event list only from regions whose samples do not fit the walk's LDS
          // staging (lds_resolve in k_region_walk): no such region, no list to resolve, no launch.
          if (info.max_region_hits > walkLdsHits())
          {
            hipLaunchKernelGGL(k_flagged_events, dim3(4096), dim3(256), 0, s, batchScratch(m), events, event_capacity,
                               batchEventCount(m), sorted, m->d_miss_counts,
                               static_cast<uint32_t *>(m->interval_counts.ptr), m->mc.region_voxels,
                               reinterpret_cast<uint32_t *>(m->h_info_dev + 1));
          }
          else
          {
            *reinterpret_cast<volatile uint32_t *>(&m->h_info[1]) = 0;  // (the event demand the next batch is sized from)
          }
          break;
        }
        // NDT / TSDF: the event sort is sized by the event count.  A batch that follows another one does not wait for
        // its own: the sort and the replay are launched on the previous batch's count plus head room, the list padded
        // up to that by k_pad_events (which also sends the true count to the host), and the count is checked once the
        // replay has been launched (settleSpeculatedEvents) -- the host round trip idled the device for 26 us per C2
        // batch.  (Not the stop-flag replay: its scans wait for the host anyway.)
        if (!stop_mode && walk_attempt == 0 && m->event_demand > 0 && event_capacity > 0)
        {
          spec_events = uint32_t(std::min<uint64_t>(event_capacity, uint64_t(m->event_demand) + m->event_demand / 8u + 4096u));
          hipExtLaunchKernelGGL(k_pad_events, dim3(256), dim3(256), 0, s, nullptr, m->ev[5], 0, events, batchEventCount(m),
                                spec_events, reinterpret_cast<uint32_t *>(m->h_info_dev + 1));
          events_speculated = true;
          n_events = spec_events;
          break;
        }
        // Otherwise the host needs the count now; an overflowing list is re-walked.
        OHMHIP_CHECK(hipMemcpyAsync(&m->h_info[1], batchEventCount(m), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        OHMHIP_CHECK(hipStreamSynchronize(s));
        n_events = *reinterpret_cast<const uint32_t *>(&m->h_info[1]);
        m->event_demand = n_events;
        if (n_events <= event_capacity)
        {
          break;
        }
        OHMHIP_CHECK(recoverEventOverflow());
        if (walk_attempt == 3)
        {
          return OHMHIP_ERR_INTERNAL;
        }
      }
    }
    else
    {
      OHMHIP_CHECK(hipEventRecord(tev[3], s));
      mark(3);
    }
    return OHMHIP_OK;
  }

  /// The walk produced more events (n_events) than the list holds: undo the count flush, grow the key buffers (sample
  /// keys must be regenerated); the caller walks again.
  int recoverEventOverflow()
  {
    hipLaunchKernelGGL(k_clear_counts, dim3(info.n_touched), dim3(256), 0, s, m->mc, regionTable(m), batchScratch(m),
                       m->d_miss_counts);
    const size_t total = size_t(n_rays) + size_t(n_events) + (size_t(n_events) >> 3) + 1024;
    OHMHIP_CHECK(m->hit_keys_a.ensure(sizeof(unsigned long long) * total, false, s));
    OHMHIP_CHECK(m->hit_keys_b.ensure(sizeof(unsigned long long) * total, false, s));
    keys_a = static_cast<unsigned long long *>(m->hit_keys_a.ptr);
    keys_b = static_cast<unsigned long long *>(m->hit_keys_b.ptr);
    sorted = keys_b;
    events = keys_a + n_rays;
    event_capacity = uint32_t(std::min<size_t>(total - n_rays, 0xfffffff0u - n_rays));
    // k_ray_bin also fills the segment buckets: only the sample keys are rewritten here (cursors already reset).
    OHMHIP_CHECK(hipMemsetAsync(m->d_info + m->info_index, 0, sizeof(BatchInfo), s));
    hipLaunchKernelGGL(k_rekey_samples, dim3(ray_blocks), dim3(256), 0, s, m->mc, regionTable(m),
                       static_cast<const RayWalk *>(batchWalks(m).ptr), n_rays, keys_a, ray_shift);
    return OHMHIP_OK;
  }

  int applyOccupancy()
  {
    // (One launch for both halves -- k_apply_occupancy -- measured slower than the two below: 0.167 vs 0.147 ms for
    // sort + apply in C1; the sample replay wants small workgroups and few registers.)
    // (tev[4], the end of the batch, is the stop event of its last kernel: what the set-up pass of the batch after the
    // next waits for before it reuses this batch's scratch copy)
    float *occ = static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]);
    uint32_t *mean_layer = static_cast<uint32_t *>(m->layers[OHMHIP_LID_MEAN]);
    uint32_t *intervals = static_cast<uint32_t *>(m->interval_counts.ptr);
    const RayWalk *walks = static_cast<const RayWalk *>(batchWalks(m).ptr);
    if (occ_inline)
    {
      // The walk applied every region it held in one chunk: the apply kernels run over k_plan's lists of the others
      // (C1: ~80 of 1243 regions; 46 -> ~20 us), the bookkeeping of all touched regions is k_batch_cleanup's.
      const uint32_t blocks_per_region = std::max<uint32_t>((info.max_region_hits + 255u) / 256u, 1u);
      // (64 bits, and capped: with kRfExcludeRay every sample region is listed, and a skewed multi-million-ray batch then
      // asks for listed regions x tiles of the densest one -- the sample part strides over the pairs instead)
      const unsigned long long hit_tiles = (unsigned long long)info.n_apply_hits * blocks_per_region;
      const uint32_t hit_blocks = uint32_t(std::min<unsigned long long>(hit_tiles, 1ull << 20));
      if (hit_blocks + info.n_apply_counts)
      {
        hipLaunchKernelGGL(k_apply_lists, dim3(hit_blocks + info.n_apply_counts * kApplyListParts), dim3(256), 0, s, m->mc,
                           regionTable(m), batchScratch(m), ray_flags, sorted, intervals, m->d_miss_counts, m->d_hit_mask,
                           d_rays, occ, hit_blocks, blocks_per_region, hit_tiles);
      }
      if (info.n_touched)
      {
        hipExtLaunchKernelGGL(k_batch_cleanup, dim3(std::min<uint32_t>(info.n_touched, 4096u)), dim3(256), 0, s, nullptr,
                              tev[4], 0, m->mc, regionTable(m), batchScratch(m), info.n_touched, m->d_hit_mask);
      }
      else
      {
        OHMHIP_CHECK(hipEventRecord(tev[4], s));
      }
      batch_end_marked = true;
      return OHMHIP_OK;
    }
    // (maps with a mean / secondary layer: every region's samples and every touched region go through the two kernels.
    // One launch with a workgroup per region that receives samples, and the count application shared by eight
    // workgroups per region, were both measured slower in round 5 -- 1.01 and 0.93 against 0.91 ms per C1 batch: short-lived
    // workgroups that leave after three dependent loads cost more than the idle lanes they replace)
    hipExtLaunchKernelGGL(k_apply_hits, dim3(ray_blocks), dim3(256), 0, s, nullptr, info.n_touched ? nullptr : tev[4], 0,
                          m->mc, regionTable(m), batchScratch(m), ray_flags, sorted, intervals, m->d_miss_counts, d_rays,
                          occ, mean_layer, sec, walks);
    if (info.n_touched)
    {
      hipExtLaunchKernelGGL(k_apply_counts, dim3(info.n_touched), dim3(1024), 0, s, nullptr, tev[4], 0, m->mc,
                            regionTable(m), batchScratch(m), ray_flags, m->d_miss_counts, m->d_hit_mask, occ, 1,
                            static_cast<uint32_t *>(nullptr), direct_segments, 0, sec.traversal,
                            sec.traversal ? m->d_traversal_acc : nullptr);
    }
    batch_end_marked = true;
    return OHMHIP_OK;
  }

  /// Order the events (samples included) per voxel in ray order and launch the mode's replay kernel.  With
  /// events_speculated the kernels that consume the list carry the guard (true count, speculated count).
  uint32_t *replay_heads = nullptr;
  uint32_t replay_blocks = 0;
  int sortAndReplay()
  {
    const size_t total = size_t(n_rays) + size_t(n_events);
    const uint32_t *guard_count = events_speculated ? batchEventCount(m) : nullptr;
    const uint32_t guard_limit = events_speculated ? spec_events : 0u;
    // The device-wide one-sweep radix sort.  Round 5 built two replacements that order the keys without it -- bucket by
    // voxel row (histogram, scan, scatter), then an LDS bitonic sort per unit of <= 4096 keys, with the replay either
    // fused into that kernel or run by the kernels below --, both bit exact, neither faster: C2 1.09 ms against 1.05,
    // C3 11.2 against 10.3 (profiles/r05_event_buckets.txt; history 4.2).  Per-key global atomics of the two bucketing
    // passes cost what the sort's passes cost, and the fused replay loses the occupancy the long voxel chains need.
    size_t sort_bytes = 0;
    OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(nullptr, sort_bytes, keys_a, keys_b, total, 0, sortEndBit(m), s));
    OHMHIP_CHECK(m->sort_temp.ensure(sort_bytes, false, s));
    size_t temp_bytes = m->sort_temp.bytes;
    OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(m->sort_temp.ptr, temp_bytes, keys_a, keys_b, total, 0,
                                                      sortEndBit(info.n_slots), s));
    // NDT: one lane per voxel group -- compact the group heads, then replay grid-stride over them (a voxel's event
    // list is long there and the maths heavy; a lane per event with the non-heads exiting ran at a few live lanes
    // per wave).
    replay_heads = nullptr;
    uint32_t *n_heads = batchEventCount(m) + 2;
    replay_blocks = uint32_t((total + 127) / 128);
    if (ndt_mode)
    {
      OHMHIP_CHECK(m->group_heads.ensure(sizeof(uint32_t) * total, false, s));
      replay_heads = static_cast<uint32_t *>(m->group_heads.ptr);
      OHMHIP_CHECK(hipMemsetAsync(n_heads, 0, sizeof(uint32_t), s));
      hipLaunchKernelGGL(k_group_heads, dim3(uint32_t((total + kHeadsPerBlock - 1) / kHeadsPerBlock)), dim3(256), 0, s,
                         sorted, uint32_t(total), replay_heads, n_heads, guard_count, guard_limit);
      replay_blocks = uint32_t(std::min<size_t>(replay_blocks, size_t(m->walk_workgroups) * 32u));
    }
    if (ndt_mode)
    {
      const bool tm = mode == OHMHIP_MODE_NDT_TM;
      hipLaunchKernelGGL(k_replay_ndt, dim3(replay_blocks), dim3(128), 0, s, m->mc, regionTable(m), sorted,
                         uint32_t(total), d_rays, d_intensities,
                         static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]),
                         static_cast<uint32_t *>(m->layers[OHMHIP_LID_MEAN]),
                         static_cast<float *>(m->layers[OHMHIP_LID_COVARIANCE]),
                         tm ? static_cast<float *>(m->layers[OHMHIP_LID_INTENSITY]) : nullptr,
                         tm ? static_cast<uint32_t *>(m->layers[OHMHIP_LID_HIT_MISS]) : nullptr, sec,
                         static_cast<const RayWalk *>(batchWalks(m).ptr), replay_heads, n_heads, guard_count, guard_limit);
    }
    else if (tsdf_mode)
    {
      hipLaunchKernelGGL(k_replay_tsdf, dim3(replay_blocks), dim3(128), 0, s, m->mc, regionTable(m), sorted,
                         uint32_t(total), d_rays, static_cast<float *>(m->layers[OHMHIP_LID_TSDF]), replay_heads, n_heads,
                         guard_count, guard_limit);
    }
    return OHMHIP_OK;
  }

  /// The true event count of a batch whose sort and replay were launched on a speculated one has reached the host (it
  /// left the device right behind the walk, long before the replay ends: the wait costs nothing).  Usually it fits
  /// and nothing remains to do; otherwise the guarded kernels did nothing, and the sort and the replay are repeated with
  /// the exact size -- after a re-walk with a larger list when even that overflowed.
  int settleSpeculatedEvents()
  {
    OHMHIP_CHECK(hipEventSynchronize(m->ev[5]));
    const uint32_t actual = *reinterpret_cast<const volatile uint32_t *>(&m->h_info[1]);
    m->event_demand = actual;
    if (actual <= spec_events)
    {
      return OHMHIP_OK;
    }
    events_speculated = false;
    n_events = actual;
    if (actual > event_capacity)
    {
      OHMHIP_CHECK(recoverEventOverflow());
      OHMHIP_CHECK(walk(1));  // (not speculated: attempts after the first wait for their count)
    }
    return sortAndReplay();
  }

  int replayEvents()
  {
    OHMHIP_CHECK(sortAndReplay());
    if (events_speculated)
    {
      OHMHIP_CHECK(settleSpeculatedEvents());
    }
    const size_t total = size_t(n_rays) + size_t(n_events);
    if (stop_mode)
    {
      // Per-ray stop positions by iteration (k_stop_replay): a scan that moves no ray's stop is the sequential result.
      OHMHIP_CHECK(m->stop_a.ensure(sizeof(uint32_t) * size_t(n_rays), false, s));
      OHMHIP_CHECK(m->stop_b.ensure(sizeof(uint32_t) * size_t(n_rays), false, s));
      uint32_t *stop = static_cast<uint32_t *>(m->stop_a.ptr);
      uint32_t *stop_next = static_cast<uint32_t *>(m->stop_b.ptr);
      OHMHIP_CHECK(hipMemsetAsync(stop, 0xff, sizeof(uint32_t) * size_t(n_rays), s));
      OHMHIP_CHECK(hipMemsetAsync(stop_next, 0xff, sizeof(uint32_t) * size_t(n_rays), s));
      uint32_t *d_changed = batchEventCount(m) + 3;
      const RayWalk *walks = static_cast<const RayWalk *>(batchWalks(m).ptr);
      float *occ = static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]);
      uint32_t *mean_layer = static_cast<uint32_t *>(m->layers[OHMHIP_LID_MEAN]);
      bool settled = false;
      for (uint64_t scan = 0; scan <= uint64_t(n_rays) && !settled; ++scan)
      {
        OHMHIP_CHECK(hipMemsetAsync(d_changed, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL((k_stop_replay<false>), dim3(replay_blocks), dim3(128), 0, s, m->mc, regionTable(m), sorted,
                           uint32_t(total), ray_flags, walks, stop, stop_next, d_rays, occ, mean_layer, sec);
        hipLaunchKernelGGL(k_stop_advance, dim3(ray_blocks), dim3(256), 0, s, stop, stop_next, n_rays, d_changed);
        uint32_t changed = 0;
        OHMHIP_CHECK(hipMemcpyAsync(&changed, d_changed, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        OHMHIP_CHECK(hipStreamSynchronize(s));
        settled = changed == 0;
      }
      if (!settled)
      {
        return OHMHIP_ERR_INTERNAL;  // (cannot happen: every scan fixes at least one more ray)
      }
      hipLaunchKernelGGL((k_stop_replay<true>), dim3(replay_blocks), dim3(128), 0, s, m->mc, regionTable(m), sorted,
                         uint32_t(total), ray_flags, walks, stop, stop_next, d_rays, occ, mean_layer, sec);
      if (info.n_touched)
      {
        // nothing was counted (every visit was an event): this clears the sample mask and the per-batch scratch
        // (with a traversal layer: the ray lengths the walk summed per voxel -- stopped rays keep adding theirs,
        // ohm/RayMapperOccupancy.cpp:166-173 runs for null updates too -- go into the layer here)
        hipLaunchKernelGGL(k_apply_counts, dim3(info.n_touched), dim3(1024), 0, s, m->mc, regionTable(m),
                           batchScratch(m), ray_flags, m->d_miss_counts, m->d_hit_mask, occ, 1,
                           static_cast<uint32_t *>(nullptr), 0u, 1, sec.traversal,
                           sec.traversal ? m->d_traversal_acc : static_cast<unsigned long long *>(nullptr));
      }
    }
    else if (ndt_mode)
    {
      const bool tm = mode == OHMHIP_MODE_NDT_TM;
      if (info.n_touched)
      {
        hipLaunchKernelGGL(k_apply_counts, dim3(info.n_touched), dim3(1024), 0, s, m->mc, regionTable(m),
                           batchScratch(m), 0u, m->d_miss_counts, m->d_hit_mask,
                           static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]), 0,
                           tm ? static_cast<uint32_t *>(m->layers[OHMHIP_LID_HIT_MISS]) : nullptr, direct_segments, 1,
                           sec.traversal, sec.traversal ? m->d_traversal_acc : nullptr);
      }
    }
    else if (info.n_touched)
    {
      hipLaunchKernelGGL(k_apply_counts_tsdf, dim3(info.n_touched * kTsdfApplyParts), dim3(256), 0, s, m->mc, regionTable(m),
                         batchScratch(m), m->d_miss_counts, m->d_hit_mask,
                         static_cast<float *>(m->layers[OHMHIP_LID_TSDF]), direct_segments);
      hipLaunchKernelGGL(k_batch_reset, dim3((info.n_touched + 255u) / 256u), dim3(256), 0, s, batchScratch(m),
                         info.n_touched);
    }
    return OHMHIP_OK;
  }

  int finish()
  {
    if (!batch_end_marked)
    {
      OHMHIP_CHECK(hipEventRecord(tev[4], s));  // (NDT / TSDF / stop-flag batches: their tails end in different kernels)
    }
    mark(4);
    m->batch_done_event[m->parity] = tev[4];
    OHMHIP_CHECK(hipGetLastError());

    m->stats = {};
    m->stats.rays_in = n_rays;
    m->stats.rays_integrated = info.rays_ok;
    m->stats.voxel_visits = info.visits;
    m->stats.ray_region_segments = info.n_segments;
    m->segments_per_ray = std::max(1.0, double(info.n_segments) / double(std::max<uint32_t>(n_rays, 1u)));
    m->stats.regions_touched = info.n_touched;
    m->stats.regions_resident = info.n_slots;
    m->rays_beyond_tiles += info.n_beyond_tiles;
    m->stats_pending = true;
    ++m->batch_seq;
    return OHMHIP_OK;
  }
};

int integrateBatch(ohmhip_map_t m, const double *d_rays, const float *d_intensities, const double *d_timestamps,
                   uint32_t n_rays, unsigned ray_flags)
{
  BatchRun run{ m, d_rays, d_intensities, d_timestamps, n_rays, ray_flags, m->stream, m->front_stream,
                m->tev[m->batch_seq % kTimingRing], uint32_t(m->batch_seq % kTimingRing) };
  OHMHIP_CHECK(run.prepare());
  for (run.attempt = 0; run.attempt < 8; ++run.attempt)
  {
    OHMHIP_CHECK(run.frontHalf());
    if (run.exhausted())
    {
      bool retry = false;
      OHMHIP_CHECK(run.resolveExhaustion(retry));
      if (retry)
      {
        continue;
      }
    }
    // (OHMHIP_DEBUG_FLAGS & 8192: host time of each step of the launch sequence, microseconds)
    const bool host_times = (m->debug_flags & 8192u) != 0;
    auto t_host = std::chrono::steady_clock::now();
    double step_us[6] = { 0, 0, 0, 0, 0, 0 };
    auto lap = [&](int k) {
      if (host_times)
      {
        const auto now = std::chrono::steady_clock::now();
        step_us[k] = std::chrono::duration<double, std::micro>(now - t_host).count();
        t_host = now;
      }
    };
    OHMHIP_CHECK(run.commitRegions());
    lap(0);
    scheduleWriteBack(m, uint32_t(m->batch_seq + 1u));  // (spill to host: keep the next eviction's victims clean)
    OHMHIP_CHECK(run.sizeBuffers());
    lap(1);
    OHMHIP_CHECK(run.binAndOrder());
    lap(2);
    OHMHIP_CHECK(run.walk());
    lap(3);
    OHMHIP_CHECK(run.occupancy_mode ? run.applyOccupancy() : run.replayEvents());
    lap(4);
    const int finish_err = run.finish();
    lap(5);
    if (host_times)
    {
      std::fprintf(stderr, "[ohmhip host] batch %llu: commit %.0f size %.0f bin+order %.0f walk %.0f apply %.0f finish %.0f us\n",
                   (unsigned long long)m->batch_seq, step_us[0], step_us[1], step_us[2], step_us[3], step_us[4], step_us[5]);
    }
    return finish_err;
  }
  return OHMHIP_ERR_CAPACITY;
}

}  // namespace

#endif  // OHMHIP_BATCH_RUN_H
