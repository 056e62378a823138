// spill_impl.h -- residency management: dropping resident regions (slot compaction), eviction to the pinned host store by
// predicted next use, pool growth for named regions, re-admission (ohmgpu/GpuLayerCache.cpp:530-584).
//
// Part of ohmhip_map.hip's translation unit (included there, in order): not a stand-alone header.
#ifndef OHMHIP_SPILL_IMPL_H
#define OHMHIP_SPILL_IMPL_H

/// Drop resident regions from the pool (ohmhip_map_remove_regions; also the second half of an eviction).
static int removeResidentRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, size_t *removed)
{
  if (removed)
  {
    *removed = 0;
  }
  hipStream_t s = m->stream;
  OHMHIP_CHECK(hipStreamSynchronize(s));
  if (!m->precleaned.empty() || !m->stale_records.empty())
  {
    // background write-back copies read the slots that are about to move
    OHMHIP_CHECK(drainWriteBack(m));
  }
  int err = refreshHostRegionTable(m);
  if (err)
  {
    return err;
  }
  const uint32_t n = m->slots_committed;
  std::vector<uint8_t> drop(n, 0);
  uint32_t k = 0;
  for (size_t i = 0; i < count; ++i)
  {
    const uint64_t packed = packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]);
    const auto it = m->region_slots.find(packed);
    if (it != m->region_slots.end() && !drop[it->second])
    {
      drop[it->second] = 1;
      ++k;
      const auto pre = m->precleaned.find(packed);
      if (pre != m->precleaned.end())
      {
        releaseStoreRecord(m, pre->second.record);  // (the copy stream is drained: nothing writes the record any more)
        m->precleaned.erase(pre);
      }
    }
  }
  if (removed)
  {
    *removed = k;
  }
  if (k == 0)
  {
    return OHMHIP_OK;
  }
  // Slots stay dense: the survivors at the tail move into the holes the removed regions leave further down, the vacated
  // tail goes back to the pristine state every unassigned slot is in, and the hash table is rebuilt from the slot keys.
  const uint32_t new_n = n - k;
  const size_t rv = size_t(m->mc.region_voxels);
  const size_t mask_row = ((rv + 31) / 32) * sizeof(uint32_t);
  uint32_t src = new_n;
  std::vector<CopyJob> jobs;
  for (uint32_t dst = 0; dst < new_n; ++dst)
  {
    if (!drop[dst])
    {
      continue;
    }
    while (drop[src])
    {
      ++src;
    }
    // (source slots lie in the tail [new_n, n), destinations below new_n: no job reads what another writes)
    for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
    {
      if (m->layers[l])
      {
        const size_t stride = rv * kLayerBytes[l];
        jobs.push_back(CopyJob{ static_cast<const char *>(m->layers[l]) + stride * src,
                                static_cast<char *>(m->layers[l]) + stride * dst, stride });
      }
    }
    jobs.push_back(CopyJob{ reinterpret_cast<const char *>(m->d_hit_mask) + mask_row * src,
                            reinterpret_cast<char *>(m->d_hit_mask) + mask_row * dst, mask_row });
    jobs.push_back(CopyJob{ reinterpret_cast<const char *>(m->d_dirty + src), reinterpret_cast<char *>(m->d_dirty + dst),
                            sizeof(uint32_t) });
    jobs.push_back(CopyJob{ reinterpret_cast<const char *>(m->d_last_use + 2 * size_t(src)),
                            reinterpret_cast<char *>(m->d_last_use + 2 * size_t(dst)), 2 * sizeof(uint32_t) });
    if (m->d_merge_base)
    {
      jobs.push_back(CopyJob{ reinterpret_cast<const char *>(m->d_merge_base + rv * src),
                              reinterpret_cast<char *>(m->d_merge_base + rv * dst), sizeof(float) * rv });
    }
    m->slot_keys_host[dst] = m->slot_keys_host[src];
    ++src;
  }
  OHMHIP_CHECK(launchCopyJobs(m, jobs, s));
  for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
  {
    if (!m->layers[l])
    {
      continue;
    }
    const size_t stride = rv * kLayerBytes[l];
    char *tail = static_cast<char *>(m->layers[l]) + stride * new_n;
    if (l == OHMHIP_LID_OCCUPANCY)
    {
      hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, s, reinterpret_cast<uint32_t *>(tail), 0x7f800000u,
                         stride * k / 4);
    }
    else
    {
      OHMHIP_CHECK(hipMemsetAsync(tail, 0, stride * k, s));
    }
  }
  OHMHIP_CHECK(hipMemsetAsync(reinterpret_cast<char *>(m->d_hit_mask) + mask_row * new_n, 0, mask_row * k, s));
  OHMHIP_CHECK(hipMemsetAsync(m->d_dirty + new_n, 0, sizeof(uint32_t) * k, s));
  OHMHIP_CHECK(hipMemsetAsync(m->d_last_use + 2 * size_t(new_n), 0, sizeof(uint32_t) * 2 * k, s));
  if (m->d_merge_base)
  {
    hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, s, reinterpret_cast<uint32_t *>(m->d_merge_base + rv * new_n),
                       0x7f800000u, rv * k);
  }
  m->slot_keys_host.resize(new_n);
  m->region_slots.clear();
  for (uint32_t i = 0; i < new_n; ++i)
  {
    m->region_slots[m->slot_keys_host[i]] = i;
  }
  OHMHIP_CHECK(hipMemsetAsync(m->d_slot_keys, 0, sizeof(uint64_t) * n, s));
  if (new_n)
  {
    OHMHIP_CHECK(hipMemcpyAsync(m->d_slot_keys, m->slot_keys_host.data(), sizeof(uint64_t) * new_n,
                                hipMemcpyHostToDevice, s));
  }
  OHMHIP_CHECK(hipMemsetAsync(m->d_keys, 0, sizeof(unsigned long long) * m->hash_capacity, s));
  OHMHIP_CHECK(hipMemcpyAsync(m->d_n_slots, &new_n, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  if (new_n)
  {
    hipLaunchKernelGGL(k_rehash, dim3((new_n + 255) / 256), dim3(256), 0, s, regionTable(m), new_n);
  }
  OHMHIP_CHECK(hipStreamSynchronize(s));
  OHMHIP_CHECK(hipGetLastError());
  m->slots_committed = new_n;
  m->spec_bucket_ok = false;  // per-slot sample ranges of the previous batch no longer describe these slots
  return OHMHIP_OK;
}

/// Spill to host, first half: copy the least recently used resident regions into the host store and drop them from the
/// pool, so that at least `want_free` slots become free (a quarter of the pool at a time, so evictions are rare).
/// Regions the current batch attempt touched carry the newest stamp (k_plan) and go last.
#include "writeback_impl.h"

static int evictColdRegions(ohmhip_map_t m, uint32_t want_free, uint32_t max_evict)
{
  const auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&](int slot, std::chrono::steady_clock::time_point &from) {
    const auto now = std::chrono::steady_clock::now();
    m->spill_ms[slot] += std::chrono::duration<double, std::milli>(now - from).count();
    from = now;
  };
  auto t_mark = t_begin;
  hipStream_t s = m->stream;
  OHMHIP_CHECK(hipStreamSynchronize(s));
  OHMHIP_CHECK(refreshHostRegionTable(m));
  const uint32_t n = m->slots_committed;
  if (n == 0 || m->d_merge_base)
  {
    return OHMHIP_ERR_CAPACITY;  // nothing to evict / replica-merge maps keep a base copy per region: not spilled
  }
  const uint32_t k = std::min(std::min(n, std::max(want_free, n / 4u)), std::max(want_free, max_evict));
  std::vector<uint32_t> stamps(2 * size_t(n)), dirty(n);
  OHMHIP_CHECK(hipMemcpy(stamps.data(), m->d_last_use, sizeof(uint32_t) * 2 * n, hipMemcpyDeviceToHost));
  OHMHIP_CHECK(hipMemcpy(dirty.data(), m->d_dirty, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
  // Who goes: the regions whose NEXT use is expected to be farthest away (rankForEviction, writeback_impl.h).
  const uint32_t now = uint32_t(m->batch_seq + 1u);
  std::vector<uint64_t> rank;
  rankForEviction(m, stamps.data(), n, now, rank, true);
  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; ++i)
  {
    order[i] = i;
  }
  // (ties -- regions last used by the same batch -- go by region key: slot numbers are handed out by atomics in the
  // set-up kernel and differ from run to run, and which regions leave should not)
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return rank[a] != rank[b] ? rank[a] > rank[b] : m->slot_keys_host[a] < m->slot_keys_host[b];
  });
  // The victims' content goes straight from the pool into pinned store records, all regions and layers by ONE kernel
  // that writes the mapped host memory itself (k_copy_jobs); the compute stream is idle here -- it was drained above.
  lap(0, t_mark);
  {
    // records for the victims that do not have a valid pre-cleaned copy in the store already
    uint32_t fresh = 0;
    for (uint32_t v = 0; v < k; ++v)
    {
      const auto pre = m->precleaned.find(m->slot_keys_host[order[v]]);
      fresh += (pre != m->precleaned.end() && pre->second.last_use == stamps[2 * size_t(order[v])]) ? 0u : 1u;
    }
    OHMHIP_CHECK(reserveStoreRecords(m, fresh));
  }
  lap(5, t_mark);
  std::vector<int16_t> victim_keys(3 * size_t(k));
  std::vector<ohmhip_map_s::SpilledRegion> content(k);
  std::vector<uint64_t> precleaned_used;  // victims whose content the background write-back had copied already
  auto giveBack = [&]() {
    // (records of pre-cleaned victims stay with the write-back's bookkeeping: the regions are still resident)
    std::unordered_map<uint64_t, char> kept;
    for (uint64_t key : precleaned_used)
    {
      kept.emplace(key, 1);
    }
    for (uint32_t v = 0; v < k; ++v)
    {
      const bool pre = content[v].record && v < uint32_t(order.size()) && kept.count(m->slot_keys_host[order[v]]) != 0;
      if (!pre)
      {
        releaseStoreRecord(m, content[v].record);
      }
      content[v].record = nullptr;
    }
  };
  std::vector<CopyJob> jobs;
  jobs.reserve(size_t(k) * 2);
  for (uint32_t v = 0; v < k; ++v)
  {
    const uint32_t slot = order[v];
    unpackRegionKey(m->slot_keys_host[slot], &victim_keys[3 * size_t(v)]);
    content[v].dirty = dirty[slot];
    content[v].last_use = stamps[2 * size_t(slot)];
    // Pre-cleaned by the background write-back and not touched since: its record is in the store already.
    const auto pre = m->precleaned.find(m->slot_keys_host[slot]);
    if (pre != m->precleaned.end())
    {
      if (pre->second.last_use == stamps[2 * size_t(slot)])
      {
        content[v].record = pre->second.record;
        precleaned_used.push_back(pre->first);
        ++m->writeback_hits;
        continue;
      }
      m->stale_records.push_back(pre->second.record);  // (recycled once the copy stream has passed its copy)
      m->precleaned.erase(pre);
      ++m->writeback_stale;
    }
    content[v].record = takeStoreRecord(m);
    if (!content[v].record)
    {
      giveBack();
      return OHMHIP_ERR_CAPACITY;
    }
    appendSlotToRecordJobs(m, slot, content[v].record, jobs);
  }
  {
    const int err = launchCopyJobs(m, jobs, m->copy_stream);
    if (err)
    {
      (void)hipStreamSynchronize(m->copy_stream);
      giveBack();
      return err;
    }
  }
  {
    int err = int(hipStreamSynchronize(m->copy_stream));
    if (!err && !precleaned_used.empty())
    {
      err = drainWriteBack(m);  // (a pre-cleaned victim's record must be complete before it stands for the region)
    }
    if (err)
    {
      giveBack();
      return err;
    }
  }
  lap(1, t_mark);
  size_t removed = 0;
  {
    // (the records of pre-cleaned victims are the spilled regions' from here on: out of the write-back's bookkeeping
    // before the removal, which would otherwise release them with the regions)
    std::vector<std::pair<uint64_t, ohmhip_map_s::Precleaned>> moved;
    for (uint64_t key : precleaned_used)
    {
      const auto it = m->precleaned.find(key);
      moved.push_back({ key, it->second });
      m->precleaned.erase(it);
    }
    const int err = removeResidentRegions(m, victim_keys.data(), k, &removed);
    if (err)
    {
      for (auto &e : moved)
      {
        m->precleaned[e.first] = e.second;
      }
      giveBack();  // the regions are still resident: nothing is lost
      return err;
    }
  }
  lap(2, t_mark);
  for (uint32_t v = 0; v < k; ++v)
  {
    m->spilled[packRegionKey(victim_keys[3 * size_t(v)], victim_keys[3 * size_t(v) + 1], victim_keys[3 * size_t(v) + 2])] =
      content[v];
  }
  m->evictions += removed;
  m->evicted_per_call = k;
  return OHMHIP_OK;
}

/// Pool growth on behalf of regions created by name (ohmhip_map_write_regions / ohmhip_map_ensure_regions): the same
/// budget rules as a batch's growth (rollbackAndGrow) -- the map's memory limit and the device's free memory.
static int growPoolForNamedRegions(ohmhip_map_t m, uint32_t total, uint32_t keep)
{
  uint32_t cap = 0;
  if (!grownCapacity(m->slot_capacity, total, cap))
  {
    return OHMHIP_ERR_CAPACITY;
  }
  const size_t per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
  if (m->memory_limit)
  {
    const uint64_t allowed = m->memory_limit / per_region;
    if (allowed < total)
    {
      return OHMHIP_ERR_CAPACITY;
    }
    cap = uint32_t(std::min<uint64_t>(cap, allowed));
  }
  size_t free_b = 0, total_b = 0;
  OHMHIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  if (per_region * size_t(cap) > free_b)
  {
    return OHMHIP_ERR_CAPACITY;
  }
  ++m->cache_full;
  return allocPool(m, cap, keep);
}

/// Before regions are created by name under a memory limit: if the named keys that are not resident yet would push the
/// pool past the limit, the least recently used OTHER regions go to the host store first (spill to host) -- or the
/// call fails with OHMHIP_ERR_CAPACITY and changes nothing.
static int makeRoomForNamedRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count)
{
  if (!m->memory_limit || count == 0)
  {
    return OHMHIP_OK;
  }
  OHMHIP_CHECK(refreshHostRegionTable(m));
  const uint64_t per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
  const uint64_t allowed = std::min<uint64_t>(m->memory_limit / per_region, kMaxRegionSlots);
  std::vector<uint32_t> named_resident;
  std::vector<uint64_t> fresh;
  for (size_t i = 0; i < count; ++i)
  {
    const uint64_t key = packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]);
    const auto it = m->region_slots.find(key);
    if (it != m->region_slots.end())
    {
      named_resident.push_back(it->second);
    }
    else if (std::find(fresh.begin(), fresh.end(), key) == fresh.end())
    {
      fresh.push_back(key);
    }
  }
  const uint64_t wanted = uint64_t(m->slots_committed) + fresh.size();
  if (wanted <= allowed)
  {
    return OHMHIP_OK;
  }
  const uint64_t need = wanted - allowed;
  std::sort(named_resident.begin(), named_resident.end());
  named_resident.erase(std::unique(named_resident.begin(), named_resident.end()), named_resident.end());
  const uint64_t evictable = uint64_t(m->slots_committed) - named_resident.size();
  if (!m->spill_enabled || m->d_merge_base || need > evictable)
  {
    return OHMHIP_ERR_CAPACITY;
  }
  if (!named_resident.empty())
  {
    // the named regions are in use now: newest stamp, so the eviction below takes others
    OHMHIP_CHECK(m->merge_slots.ensure(sizeof(uint32_t) * named_resident.size(), false, m->stream));
    OHMHIP_CHECK(hipMemcpy(m->merge_slots.ptr, named_resident.data(), sizeof(uint32_t) * named_resident.size(),
                           hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_touch_use_at, dim3(64), dim3(256), 0, m->stream, m->d_last_use,
                       static_cast<const uint32_t *>(m->merge_slots.ptr), named_resident.size(),
                       uint32_t(m->batch_seq + 1u));
    OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  }
  return evictColdRegions(m, uint32_t(need), uint32_t(evictable));
}

namespace
{
/// Queue the copies that put stored regions back into pool slots (each slot holds a fresh, unobserved region of the
/// same key): layers and mask rows straight from the pinned records on the copy stream, the dirty bits OR-ed in by one
/// small kernel per bit pattern behind them.  Returns with everything QUEUED; the caller waits for the copy stream.
int queueReadmission(ohmhip_map_t m, const std::vector<std::pair<uint32_t, ohmhip_map_s::SpilledRegion>> &back)
{
  const size_t rv = size_t(m->mc.region_voxels);
  const ohmhip_map_s::HostStore &st = m->store;
  const bool keep_mask = m->config.mode != OHMHIP_MODE_OCCUPANCY;
  std::vector<uint32_t> dirty_slots[4];
  std::vector<uint32_t> use_pairs;  // (slot, stamp of the region's last use before it left the pool)
  use_pairs.reserve(back.size() * 2);
  std::vector<CopyJob> jobs;
  jobs.reserve(back.size() * 2);
  for (const auto &entry : back)
  {
    const uint32_t slot = entry.first;
    const char *record = entry.second.record;
    use_pairs.push_back(slot);
    use_pairs.push_back(entry.second.last_use);
    if (entry.second.last_use != 0)
    {
      const uint32_t gap = uint32_t(m->batch_seq + 1u) - entry.second.last_use;
      if (m->readmit_periods.size() < 256)
      {
        m->readmit_periods.push_back(gap);
      }
      else
      {
        m->readmit_periods[m->readmit_period_at++ % 256] = gap;
      }
    }
    for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
    {
      if (m->layers[l])
      {
        const size_t stride = rv * kLayerBytes[l];
        jobs.push_back(CopyJob{ record + st.layer_offset[l], static_cast<char *>(m->layers[l]) + stride * slot, stride });
      }
    }
    if (keep_mask)
    {
      jobs.push_back(CopyJob{ record + st.mask_offset, reinterpret_cast<char *>(m->d_hit_mask) + st.mask_bytes * slot,
                              st.mask_bytes });
    }
    dirty_slots[entry.second.dirty & (kDirtySync | kDirtyMerge)].push_back(slot);
  }
  OHMHIP_CHECK(launchCopyJobs(m, jobs, m->copy_stream));
  // The use history comes back with the content: the slot's "use before the gap" is the region's last use before it
  // left (the slot itself is new: its own last-use stamp is this batch's, or is set by the caller).
  OHMHIP_CHECK(m->use_scratch.ensure(sizeof(uint32_t) * use_pairs.size(), false, m->copy_stream));
  OHMHIP_CHECK(hipMemcpy(m->use_scratch.ptr, use_pairs.data(), sizeof(uint32_t) * use_pairs.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_set_prev_use, dim3(64), dim3(256), 0, m->copy_stream, m->d_last_use,
                     static_cast<const uint32_t *>(m->use_scratch.ptr), back.size());
  // (k_plan may be OR-ing this batch's bits into the same words: atomic ORs, from a persistent index scratch)
  size_t n_index = dirty_slots[1].size() + dirty_slots[2].size() + dirty_slots[3].size();
  if (n_index)
  {
    OHMHIP_CHECK(m->merge_slots.ensure(sizeof(uint32_t) * n_index, false, m->copy_stream));
    uint32_t *d_index = static_cast<uint32_t *>(m->merge_slots.ptr);
    for (uint32_t bits = 1; bits < 4; ++bits)
    {
      if (dirty_slots[bits].empty())
      {
        continue;
      }
      OHMHIP_CHECK(hipMemcpy(d_index, dirty_slots[bits].data(), sizeof(uint32_t) * dirty_slots[bits].size(),
                             hipMemcpyHostToDevice));  // (blocking: the vector goes out of scope; a few hundred bytes)
      hipLaunchKernelGGL(k_or_at_u32, dim3(64), dim3(256), 0, m->copy_stream, m->d_dirty, d_index,
                         dirty_slots[bits].size(), bits);
      d_index += dirty_slots[bits].size();
    }
  }
  return hipGetLastError();
}
}  // namespace

/// Spill to host, second half: a batch's set-up pass has just created the slots [first_slot, end_slot); those whose key
/// is in the host store get their content back before anything reads or updates the layers.  Entries leave the store
/// only once their content is safely back in the pool (a failure on the way must not lose a region).
static int readmitSpilledSlots(ohmhip_map_t m, uint32_t first_slot, uint32_t end_slot)
{
  if (m->spilled.empty() || end_slot <= first_slot)
  {
    return OHMHIP_OK;
  }
  std::vector<uint64_t> keys(end_slot - first_slot);
  OHMHIP_CHECK(hipMemcpy(keys.data(), m->d_slot_keys + first_slot, sizeof(uint64_t) * keys.size(), hipMemcpyDeviceToHost));
  // (slot, stored content) of the new slots that have content waiting, in slot order
  std::vector<std::pair<uint32_t, ohmhip_map_s::SpilledRegion>> back;
  for (size_t i = 0; i < keys.size(); ++i)
  {
    const auto it = m->spilled.find(keys[i]);
    if (it != m->spilled.end())
    {
      back.emplace_back(first_slot + uint32_t(i), it->second);
    }
  }
  if (back.empty())
  {
    return OHMHIP_OK;
  }
  const auto t_begin = std::chrono::steady_clock::now();
  int err = queueReadmission(m, back);
  const int sync_err = int(hipStreamSynchronize(m->copy_stream));
  err = err ? err : sync_err;
  m->spill_ms[3] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (err)
  {
    return err;  // the store still holds every region; the batch fails and is rolled back by the caller
  }
  for (size_t i = 0; i < keys.size(); ++i)
  {
    const auto it = m->spilled.find(keys[i]);
    if (it != m->spilled.end())
    {
      releaseStoreRecord(m, it->second.record);
      m->spilled.erase(it);
    }
  }
  m->readmissions += back.size();
  return OHMHIP_OK;
}

/// Bring stored regions back for an upload / a caller that wants their slots (ohmhip_map_write_regions,
/// ohmhip_map_ensure_regions): afterwards the keys are ordinary resident regions.
static int readmitSpilledKeys(ohmhip_map_t m, const int16_t *keys_xyz, size_t count)
{
  if (m->spilled.empty())
  {
    return OHMHIP_OK;
  }
  std::vector<int16_t> wanted;
  std::vector<uint64_t> wanted_packed;
  for (size_t i = 0; i < count; ++i)
  {
    const uint64_t packed = packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]);
    if (m->spilled.count(packed) && std::find(wanted_packed.begin(), wanted_packed.end(), packed) == wanted_packed.end())
    {
      wanted.insert(wanted.end(), keys_xyz + 3 * i, keys_xyz + 3 * i + 3);
      wanted_packed.push_back(packed);
    }
  }
  if (wanted.empty())
  {
    return OHMHIP_OK;
  }
  // Take the entries out of the store while ensure_regions runs (it would otherwise come straight back here); they go
  // back in if anything fails before their content is in the pool.
  std::vector<ohmhip_map_s::SpilledRegion> content(wanted_packed.size());
  for (size_t i = 0; i < content.size(); ++i)
  {
    const auto it = m->spilled.find(wanted_packed[i]);
    content[i] = it->second;
    m->spilled.erase(it);
  }
  auto putBack = [&]() {
    for (size_t i = 0; i < content.size(); ++i)
    {
      m->spilled[wanted_packed[i]] = content[i];
    }
  };
  std::vector<uint32_t> slots(content.size());
  int err = ohmhip_map_ensure_regions(m, wanted.data(), content.size(), slots.data());
  if (err)
  {
    // ensure_regions created some of the regions fresh before it failed: those must not shadow the stored content
    size_t removed = 0;
    (void)removeResidentRegions(m, wanted.data(), content.size(), &removed);
    putBack();
    return err;
  }
  std::vector<std::pair<uint32_t, ohmhip_map_s::SpilledRegion>> back;
  for (size_t i = 0; i < content.size(); ++i)
  {
    back.emplace_back(slots[i], content[i]);
  }
  err = queueReadmission(m, back);
  const int sync_err = int(hipStreamSynchronize(m->copy_stream));
  err = err ? err : sync_err;
  if (err)
  {
    size_t removed = 0;
    (void)removeResidentRegions(m, wanted.data(), content.size(), &removed);
    putBack();
    return err;
  }
  // re-admitted by name: they are in use NOW -- stamp them so the next eviction does not pick them first
  {
    OHMHIP_CHECK(m->merge_slots.ensure(sizeof(uint32_t) * slots.size(), false, m->stream));
    OHMHIP_CHECK(hipMemcpy(m->merge_slots.ptr, slots.data(), sizeof(uint32_t) * slots.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_touch_use_at, dim3(64), dim3(256), 0, m->stream, m->d_last_use,
                       static_cast<const uint32_t *>(m->merge_slots.ptr), slots.size(), uint32_t(m->batch_seq + 1u));
    OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  }
  for (auto &c : content)
  {
    releaseStoreRecord(m, c.record);
  }
  m->readmissions += content.size();
  return OHMHIP_OK;
}


#endif  // OHMHIP_SPILL_IMPL_H
