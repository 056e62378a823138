// writeback_impl.h -- background write-back of the spill-to-host residency manager (include/ohmhip.h, "SPILL TO HOST").
// Included by ohmhip_map.hip ahead of evictColdRegions.
//
// The reference's layer cache overlaps the download of the slot it is about to reuse with the work already queued
// (ohmgpu/GpuLayerCache.cpp:530-584: events per cache entry, two staging buffers).  The counterpart here: while batches
// run, the regions the eviction policy would pick NEXT are copied to records of the pinned host store on the copy
// stream ("pre-cleaned").  An eviction then finds most of its victims clean -- their record is already in the store --
// and only drops them from the pool; the copy-out leaves the batch's critical path.
//
// A pre-cleaned copy is taken of a region the running batch does not touch, behind the event of the batch before it, and
// is VALID for as long as no later batch touches the region: k_plan stamps every region a batch touches (BatchScratch::
// last_use), the stamps of the resident regions travel to the host with every batch's plan summary, and a copy whose
// region shows a newer stamp than the one it was taken at is discarded.  Nothing is ever evicted on the strength of a
// stale copy, so results do not depend on any of this (tests/test_gpu_spill.py compares with an unbounded pool).
#ifndef OHMHIP_WRITEBACK_IMPL_H
#define OHMHIP_WRITEBACK_IMPL_H

namespace
{
/// Eviction order of the resident regions (larger rank = goes earlier): the regions whose NEXT use is expected to be
/// farthest away.  A region that came back after a gap has a period (last use - the use before the gap); while it is on
/// schedule (idle for less than two periods) its next use is predicted at last + period.  Everything else -- never
/// re-used, or overdue -- has no prediction and goes first, least recently used first: a sensor moving through new space
/// sees plain LRU, a sensor sweeping a map larger than the pool again and again (the cyclic access LRU is worst at: it
/// evicts exactly what the next calls need) keeps a fixed part resident and cycles the rest.  Regions stamped `now` are
/// in use by the batch being attempted: rank 0, always last.
/// `prefer_clean`: a region whose content the background write-back has already copied to the store (and which has not
/// been touched since) counts as one batch farther away / older than it is.  The freshest candidates -- the regions of
/// the batch that has only just finished -- are always dirty; the ones used a batch earlier are nearly as good victims
/// and may be clean, and taking them keeps the copy-out off the evicting batch's critical path.  (On the cyclic sweep
/// this shifts the resident set by one batch's worth of regions; the number of evictions is the same.)
void rankForEviction(ohmhip_map_t m, const uint32_t *stamps, uint32_t n, uint32_t now, std::vector<uint64_t> &rank,
                     bool prefer_clean = false)
{
  // A region without a period of its own borrows the one the map's re-admissions show (their median), while it is
  // younger than that: when regions keep coming back after P batches, one used a moment ago is P batches from its next
  // use, one used P - 1 batches ago is about to be needed.
  uint32_t common_period = 0;
  if (m->readmit_periods.size() >= 16)
  {
    std::vector<uint32_t> sorted_periods(m->readmit_periods);
    std::nth_element(sorted_periods.begin(), sorted_periods.begin() + sorted_periods.size() / 2, sorted_periods.end());
    common_period = sorted_periods[sorted_periods.size() / 2];
  }
  rank.resize(n);
  for (uint32_t i = 0; i < n; ++i)
  {
    const uint32_t last = stamps[2 * size_t(i)], prev = stamps[2 * size_t(i) + 1];
    const uint32_t age = now - last;  // (0: in use by the batch being attempted)
    uint32_t period = (prev != 0 && last > prev) ? last - prev : 0;
    if (period < 2 && common_period >= 2 && age < common_period)
    {
      period = common_period;
    }
    uint32_t clean = 0;
    if (prefer_clean && i < m->slot_keys_host.size())
    {
      const auto pre = m->precleaned.find(m->slot_keys_host[i]);
      clean = (pre != m->precleaned.end() && pre->second.last_use == last) ? 1u : 0u;
    }
    if (last == now)
    {
      rank[i] = 0;
    }
    else if (period >= 2 && age < 2 * period)
    {
      const uint32_t next = last + period;               // predicted next use
      // farther away = earlier out; among equals the clean one first
      rank[i] = (uint64_t(1) << 32) | (uint64_t(((next > now) ? next - now : 0u) + clean) << 1) | clean;
    }
    else
    {
      rank[i] = (uint64_t(2) << 32) | (uint64_t(age + clean) << 1) | clean;  // no prediction: before all predicted ones, oldest first
    }
  }
}

/// Copy jobs that move slot `slot`'s block of every layer (+ its replay mask row where that is persistent state) into
/// store record `record`.
void appendSlotToRecordJobs(ohmhip_map_t m, uint32_t slot, char *record, std::vector<CopyJob> &jobs)
{
  const size_t rv = size_t(m->mc.region_voxels);
  const ohmhip_map_s::HostStore &st = m->store;
  for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
  {
    if (m->layers[l])
    {
      const size_t stride = rv * kLayerBytes[l];
      jobs.push_back(CopyJob{ static_cast<const char *>(m->layers[l]) + stride * slot, record + st.layer_offset[l], stride });
    }
  }
  if (m->config.mode != OHMHIP_MODE_OCCUPANCY)  // (transient in occupancy mode: empty between batches)
  {
    jobs.push_back(CopyJob{ reinterpret_cast<const char *>(m->d_hit_mask) + st.mask_bytes * slot, record + st.mask_offset,
                            st.mask_bytes });
  }
  else
  {
    std::memset(record + st.mask_offset, 0, st.mask_bytes);
  }
}

/// Wait for the write-back copies in flight (they read pool slots and write store records).
int drainWriteBack(ohmhip_map_t m)
{
  return m->wb_stream ? int(hipStreamSynchronize(m->wb_stream)) : OHMHIP_OK;
}

/// Forget every pre-cleaned copy (the pool is about to be rebuilt, cleared or destroyed): waits for copies in flight.
void dropPrecleaned(ohmhip_map_t m)
{
  if (m->precleaned.empty() && m->stale_records.empty())
  {
    return;
  }
  (void)drainWriteBack(m);
  for (auto &entry : m->precleaned)
  {
    releaseStoreRecord(m, entry.second.record);
  }
  m->precleaned.clear();
  for (char *rec : m->stale_records)  // (no copy is in flight any more: discarded copies' records are free again)
  {
    releaseStoreRecord(m, rec);
  }
  m->stale_records.clear();
}

/// The pre-cleaned copy of one region is void (an upload rewrote the region, or it left the map).
void dropPrecleanedKey(ohmhip_map_t m, uint64_t key)
{
  const auto it = m->precleaned.find(key);
  if (it != m->precleaned.end())
  {
    (void)drainWriteBack(m);  // (its copy may still be in flight: the record goes back to the free list)
    releaseStoreRecord(m, it->second.record);
    m->precleaned.erase(it);
  }
}

/// Queue the stamps of the resident regions for the host, on `stream` (the stream of the batch's plan: they arrive with
/// its summary).
int queueUseStamps(ohmhip_map_t m, hipStream_t stream)
{
  m->h_use_slots = 0;
  if (!m->spill_enabled || m->slots_committed == 0)
  {
    return OHMHIP_OK;
  }
  const size_t want = 2 * size_t(m->slots_committed);
  if (want > m->h_use_capacity)
  {
    if (m->h_use)
    {
      OHMHIP_CHECK(hipStreamSynchronize(stream));
      OHMHIP_CHECK(hipHostFree(m->h_use));
      m->h_use = nullptr;
      m->h_use_capacity = 0;
    }
    const size_t cap = std::max<size_t>(want + want / 2, 4096);
    OHMHIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&m->h_use), sizeof(uint32_t) * cap, hipHostMallocDefault));
    m->h_use_capacity = cap;
  }
  OHMHIP_CHECK(hipMemcpyAsync(m->h_use, m->d_last_use, sizeof(uint32_t) * want, hipMemcpyDeviceToHost, stream));
  m->h_use_slots = m->slots_committed;
  return OHMHIP_OK;
}

/// After a batch's plan has been accepted: discard pre-cleaned copies of regions that were touched since, and queue the
/// copies of the regions an eviction would take next.  `now`: the running batch's stamp.  Never fails a batch: on any
/// error the write-back simply does not happen.
void scheduleWriteBack(ohmhip_map_t m, uint32_t now)
{
  if (!m->spill_enabled || m->writeback_off || m->h_use_slots == 0 || m->d_merge_base)
  {
    return;
  }
  struct HostTimer
  {
    ohmhip_map_t m;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~HostTimer() { m->wb_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
  } host_timer{ m };
  if (m->slot_keys_host.size() < m->h_use_slots && refreshHostRegionTable(m) != OHMHIP_OK)
  {
    return;
  }
  const uint32_t n = std::min(m->h_use_slots, uint32_t(m->slot_keys_host.size()));
  const uint32_t *stamps = m->h_use;
  // 1. copies of regions touched since they were taken are void
  for (auto it = m->precleaned.begin(); it != m->precleaned.end();)
  {
    const auto slot_it = m->region_slots.find(it->first);
    const bool known = slot_it != m->region_slots.end() && slot_it->second < n;
    if (!known || stamps[2 * size_t(slot_it->second)] != it->second.last_use)
    {
      if (m->debug_flags & 512u)
      {
        std::fprintf(stderr, "[ohmhip writeback] stale at batch %u: known %d copy-stamp %u now-stamp %u\n", now, int(known),
                     it->second.last_use, known ? stamps[2 * size_t(slot_it->second)] : 0u);
      }
      // (the copy kernel may still be writing the record: it is only handed out again behind the copy stream)
      m->stale_records.push_back(it->second.record);
      it = m->precleaned.erase(it);
      ++m->writeback_stale;
    }
    else
    {
      ++it;
    }
  }
  // 2. how many to keep clean ahead of the evictions: what recent evictions took, with a margin
  const uint64_t per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
  const uint64_t allowed = m->memory_limit ? std::min<uint64_t>(m->memory_limit / per_region, kMaxRegionSlots) : m->slot_capacity;
  if (uint64_t(n) * 2 < allowed || m->evictions == 0)
  {
    return;  // the pool is not under pressure (yet): nothing has ever had to leave, or it is half empty
  }
  const uint32_t target = std::min<uint32_t>(n / 2u, std::max<uint32_t>(32u, m->evicted_per_call + m->evicted_per_call / 2u));
  if (m->precleaned.size() >= target)
  {
    return;
  }
  uint32_t want = target - uint32_t(m->precleaned.size());
  // (bounded per batch: the link should carry the write-back beside the batches, not instead of them)
  want = std::min<uint32_t>(want, std::max<uint32_t>(32u, uint32_t((uint64_t(96) << 20) / std::max<uint64_t>(m->store.record_bytes, 1))));
  std::vector<uint64_t> rank;
  rankForEviction(m, stamps, n, now, rank);
  std::vector<uint32_t> order;
  order.reserve(n);
  for (uint32_t i = 0; i < n; ++i)
  {
    if (rank[i] != 0 && m->precleaned.find(m->slot_keys_host[i]) == m->precleaned.end())
    {
      order.push_back(i);
    }
  }
  if (order.empty())
  {
    return;
  }
  want = std::min<uint32_t>(want, uint32_t(order.size()));
  std::partial_sort(order.begin(), order.begin() + want, order.end(),
                    [&](uint32_t a, uint32_t b) {
                      return rank[a] != rank[b] ? rank[a] > rank[b] : m->slot_keys_host[a] < m->slot_keys_host[b];
                    });
  // The job list of a write-back lives in one of a few device buffers used in turn; a buffer whose kernel has not
  // finished yet means the link is still busy with earlier write-backs: skip this round.
  ohmhip_map_s::WritebackRing &ring = m->wb_ring[m->wb_next % ohmhip_map_s::kWritebackRing];
  if (!ring.done && hipEventCreateWithFlags(&ring.done, hipEventDisableTiming) != hipSuccess)
  {
    ring.done = nullptr;
    return;
  }
  if (ring.used && hipEventQuery(ring.done) != hipSuccess)
  {
    (void)hipGetLastError();
    return;
  }
  // records handed back by step 1 may still be written by an earlier copy kernel: recycle them only once the copy
  // stream has run dry
  if (!m->stale_records.empty() && (!m->wb_stream || hipStreamQuery(m->wb_stream) == hipSuccess))
  {
    for (char *rec : m->stale_records)
    {
      releaseStoreRecord(m, rec);
    }
    m->stale_records.clear();
  }
  (void)hipGetLastError();
  if (reserveStoreRecords(m, want) != OHMHIP_OK)
  {
    return;
  }
  std::vector<CopyJob> jobs;
  jobs.reserve(size_t(want) * 3);
  std::vector<std::pair<uint64_t, ohmhip_map_s::Precleaned>> taken;
  for (uint32_t v = 0; v < want; ++v)
  {
    const uint32_t slot = order[v];
    char *rec = takeStoreRecord(m);
    if (!rec)
    {
      break;
    }
    appendSlotToRecordJobs(m, slot, rec, jobs);
    taken.push_back({ m->slot_keys_host[slot], ohmhip_map_s::Precleaned{ rec, stamps[2 * size_t(slot)] } });
  }
  if (taken.empty())
  {
    return;
  }
  // behind the batch BEFORE the running one (the running one does not touch these regions; an earlier one may have)
  if (!m->wb_stream && hipStreamCreateWithFlags(&m->wb_stream, hipStreamNonBlocking) != hipSuccess)
  {
    m->wb_stream = nullptr;
    for (auto &t : taken)
    {
      releaseStoreRecord(m, t.second.record);
    }
    return;
  }
  hipStream_t cs = m->wb_stream;
  bool ok = true;
  if (m->batch_done_event[m->parity ^ 1u])
  {
    ok = hipStreamWaitEvent(cs, m->batch_done_event[m->parity ^ 1u], 0) == hipSuccess;
  }
  if (ok)
  {
    const size_t bytes = sizeof(CopyJob) * jobs.size();
    if (bytes > ring.capacity)
    {
      if (ring.jobs_host)
      {
        (void)hipHostFree(ring.jobs_host);  // (its last kernel has finished: checked above)
        ring.jobs_host = ring.jobs_dev = nullptr;
        ring.capacity = 0;
      }
      const size_t cap = std::max<size_t>(bytes + bytes / 2, size_t(1) << 16);
      ok = hipHostMalloc(&ring.jobs_host, cap, hipHostMallocMapped) == hipSuccess &&
           hipHostGetDevicePointer(&ring.jobs_dev, ring.jobs_host, 0) == hipSuccess;
      ring.capacity = ok ? cap : 0;
    }
    if (ok)
    {
      std::memcpy(ring.jobs_host, jobs.data(), bytes);
    }
  }
  if (ok)
  {
    hipLaunchKernelGGL(k_copy_jobs_few, dim3(m->writeback_workgroups), dim3(256), 0, cs,
                       static_cast<const CopyJob *>(ring.jobs_dev), uint32_t(jobs.size()));
    ok = hipGetLastError() == hipSuccess && hipEventRecord(ring.done, cs) == hipSuccess;
    ring.used = ok;
    ++m->wb_next;
  }
  if (!ok)
  {
    (void)hipStreamSynchronize(cs);
    for (auto &t : taken)
    {
      releaseStoreRecord(m, t.second.record);
    }
    return;
  }
  for (auto &t : taken)
  {
    m->precleaned[t.first] = t.second;
  }
  m->writebacks += taken.size();
  if (m->debug_flags & 512u)
  {
    uint32_t lo = 0xffffffffu, hi = 0;
    for (auto &t : taken)
    {
      lo = std::min(lo, t.second.last_use);
      hi = std::max(hi, t.second.last_use);
    }
    std::fprintf(stderr, "[ohmhip writeback] batch %u: resident %u, %zu copies queued (stamps %u..%u), %zu held, target %u\n", now,
                 n, taken.size(), lo, hi, m->precleaned.size(), target);
  }
}
}  // namespace

#endif  // OHMHIP_WRITEBACK_IMPL_H
