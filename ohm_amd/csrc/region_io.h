// region_io.h -- the entry points that list, read, write, create and remove regions (GpuLayerCache sync / upload,
// ohmgpu/GpuLayerCache.cpp:172-182, 429-633; MapRegionCache::remove).
//
// Part of ohmhip_map.hip's translation unit (included there, in order): not a stand-alone header.
#ifndef OHMHIP_REGION_IO_H
#define OHMHIP_REGION_IO_H

extern "C" {

int ohmhip_map_region_count(ohmhip_map_t m, size_t *count)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !count)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (tiledBoundary(m))
  {
    return tiledListRegions(m, false, nullptr, 0, count);
  }
  *count = size_t(m->slots_committed) + m->spilled.size();  // (regions in the host store are part of the map)
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_regions(ohmhip_map_t m, int16_t *keys_xyz, size_t capacity, size_t *count)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !count)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (tiledBoundary(m))
  {
    return tiledListRegions(m, false, keys_xyz, capacity, count);
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  const int err = refreshHostRegionTable(m);
  if (err)
  {
    return err;
  }
  *count = m->slot_keys_host.size() + m->spilled.size();
  size_t at = 0;
  for (; at < m->slot_keys_host.size() && at < capacity && keys_xyz; ++at)
  {
    unpackRegionKey(m->slot_keys_host[at], keys_xyz + 3 * at);
  }
  for (auto it = m->spilled.begin(); it != m->spilled.end() && at < capacity && keys_xyz; ++it, ++at)
  {
    unpackRegionKey(it->first, keys_xyz + 3 * at);
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_dirty_regions(ohmhip_map_t m, int16_t *keys_xyz, size_t capacity, size_t *count)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !count)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (tiledBoundary(m))
  {
    return tiledListRegions(m, true, keys_xyz, capacity, count);
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  int err = refreshHostRegionTable(m);
  if (err)
  {
    return err;
  }
  std::vector<uint32_t> dirty(m->slots_committed);
  if (!dirty.empty())
  {
    OHMHIP_CHECK(hipMemcpy(dirty.data(), m->d_dirty, sizeof(uint32_t) * dirty.size(), hipMemcpyDeviceToHost));
  }
  size_t n = 0;
  for (size_t i = 0; i < dirty.size(); ++i)
  {
    if (dirty[i] & kDirtySync)
    {
      if (keys_xyz && n < capacity)
      {
        unpackRegionKey(m->slot_keys_host[i], keys_xyz + 3 * n);
      }
      ++n;
    }
  }
  for (const auto &entry : m->spilled)
  {
    if (entry.second.dirty & kDirtySync)
    {
      if (keys_xyz && n < capacity)
      {
        unpackRegionKey(entry.first, keys_xyz + 3 * n);
      }
      ++n;
    }
  }
  *count = n;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_clear_dirty(ohmhip_map_t m)
try
{
  OHMHIP_SETTLE(m);
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(k_and_u32, dim3(256), dim3(256), 0, m->stream, m->d_dirty, ~kDirtySync, size_t(m->slot_capacity));
  for (auto &entry : m->spilled)
  {
    entry.second.dirty &= ~kDirtySync;
  }
  return hipGetLastError();
}
OHMHIP_ABI_CATCH

int ohmhip_map_region_slot(ohmhip_map_t m, const int16_t key_xyz[3], uint32_t *slot)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !key_xyz || !slot)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (tiledBoundary(m))
  {
    return OHMHIP_ERR_UNSUPPORTED;  // a region cut into tiles has no single slot (zero-copy views: 32^3 regions)
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  const int err = refreshHostRegionTable(m);
  if (err)
  {
    return err;
  }
  const auto it = m->region_slots.find(packRegionKey(key_xyz[0], key_xyz[1], key_xyz[2]));
  if (it == m->region_slots.end())
  {
    return OHMHIP_ERR_NOT_FOUND;
  }
  *slot = it->second;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_device_layer_ptr(ohmhip_map_t m, int layer_id, void **device_ptr, size_t *region_stride_bytes)
try
{
  OHMHIP_SETTLE(m);
  if (!m || layer_id < 0 || layer_id >= OHMHIP_LID_COUNT || !device_ptr)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (!m->layers[layer_id])
  {
    return OHMHIP_ERR_NOT_FOUND;
  }
  *device_ptr = m->layers[layer_id];
  if (region_stride_bytes)
  {
    *region_stride_bytes = size_t(m->mc.region_voxels) * kLayerBytes[layer_id];
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_read_regions(ohmhip_map_t m, int layer_id, const int16_t *keys_xyz, size_t count, void *const *dsts)
try
{
  OHMHIP_SETTLE(m);
  if (!m || layer_id < 0 || layer_id >= OHMHIP_LID_COUNT || (count && (!keys_xyz || !dsts)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (!m->layers[layer_id])
  {
    return OHMHIP_ERR_NOT_FOUND;
  }
  if (tiledBoundary(m))
  {
    return tiledReadRegions(m, layer_id, keys_xyz, count, dsts);
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));  // fence: all queued integration done
  int err = refreshHostRegionTable(m);
  if (err)
  {
    return err;
  }
  const size_t stride = size_t(m->mc.region_voxels) * kLayerBytes[layer_id];
  std::vector<int16_t> resident_keys;
  std::vector<void *> resident_dsts;
  if (!m->spilled.empty())
  {
    // Regions in the host store are copied straight from there; the rest goes through the device path below.
    for (size_t k = 0; k < count; ++k)
    {
      const int16_t *key = keys_xyz + 3 * k;
      const auto it = m->spilled.find(packRegionKey(key[0], key[1], key[2]));
      if (it != m->spilled.end())
      {
        std::memcpy(dsts[k], it->second.record + m->store.layer_offset[layer_id], stride);
      }
      else
      {
        resident_keys.insert(resident_keys.end(), key, key + 3);
        resident_dsts.push_back(dsts[k]);
      }
    }
    keys_xyz = resident_keys.data();
    dsts = resident_dsts.data();
    count = resident_dsts.size();
  }
  // Pinned double-buffered staging on the copy stream, 64 regions per burst.
  const size_t burst = 64;
  err = ensureStage(m, std::max(m->h_stage_bytes, 2 * burst * stride));
  if (err)
  {
    return err;
  }
  char *stage[2] = { static_cast<char *>(m->h_stage), static_cast<char *>(m->h_stage) + burst * stride };
  hipEvent_t done[2] = { m->ev[6], nullptr };
  // Requests in pool-slot order: consecutive slots are one contiguous device range and travel as ONE copy (a first
  // sync of a freshly built map is a handful of large copies instead of one small copy per region).
  std::vector<std::pair<uint32_t, size_t>> order(count);
  for (size_t k = 0; k < count; ++k)
  {
    const int16_t *key = keys_xyz + 3 * k;
    const auto it = m->region_slots.find(packRegionKey(key[0], key[1], key[2]));
    if (it == m->region_slots.end())
    {
      return OHMHIP_ERR_NOT_FOUND;
    }
    order[k] = { it->second, k };
  }
  std::sort(order.begin(), order.end());
  OHMHIP_CHECK(hipEventCreate(&done[1]));
  size_t pending_base[2] = { 0, 0 };
  size_t pending_n[2] = { 0, 0 };
  int status = OHMHIP_OK;
  auto scatter = [&](int b, size_t first, size_t last) {
    for (size_t k = first; k < last; ++k)
    {
      std::memcpy(dsts[order[pending_base[b] + k].second], stage[b] + k * stride, stride);
    }
  };
  auto drain = [&](int b) -> int {
    if (pending_n[b])
    {
      OHMHIP_CHECK(hipEventSynchronize(done[b]));
      // The host-side scatter into the callers' blocks is memory-bandwidth work: the map's pool threads share a large
      // burst (one core copies ~10 GB/s, the link delivers ~50).
      const size_t n = pending_n[b];
      const size_t workers = std::min<size_t>(kStageThreads, (n * stride) >> 20);
      if (workers <= 1)
      {
        scatter(b, 0, n);
      }
      else
      {
        std::atomic<size_t> next(0);
        auto work = [&](unsigned) {
          for (size_t k = next.fetch_add(1); k < n; k = next.fetch_add(1))
          {
            scatter(b, k, k + 1);
          }
        };
        StagePool &pool = stagePool(m);
        pool.start(unsigned(workers - 1), work);
        work(0);
        pool.wait();
      }
      pending_n[b] = 0;
    }
    return OHMHIP_OK;
  };
  int b = 0;
  for (size_t base = 0; base < count && status == OHMHIP_OK; base += burst, b ^= 1)
  {
    status = drain(b);
    if (status)
    {
      break;
    }
    const size_t n = std::min(burst, count - base);
    for (size_t k = 0; k < n && status == OHMHIP_OK;)
    {
      size_t run = 1;
      while (k + run < n && order[base + k + run].first == order[base + k].first + uint32_t(run))
      {
        ++run;
      }
      const char *src = static_cast<const char *>(m->layers[layer_id]) + size_t(order[base + k].first) * stride;
      const hipError_t e =
        hipMemcpyAsync(stage[b] + k * stride, src, run * stride, hipMemcpyDeviceToHost, m->copy_stream);
      if (e != hipSuccess)
      {
        status = int(e);
      }
      k += run;
    }
    if (status == OHMHIP_OK)
    {
      const hipError_t e = hipEventRecord(done[b], m->copy_stream);
      if (e != hipSuccess)
      {
        status = int(e);
      }
      pending_base[b] = base;
      pending_n[b] = n;
    }
  }
  if (status == OHMHIP_OK)
  {
    status = drain(0);
  }
  if (status == OHMHIP_OK)
  {
    status = drain(1);
  }
  (void)hipStreamSynchronize(m->copy_stream);
  (void)hipEventDestroy(done[1]);
  return status;
}
OHMHIP_ABI_CATCH

int ohmhip_map_write_regions(ohmhip_map_t m, int layer_id, const int16_t *keys_xyz, size_t count,
                             const void *const *srcs)
try
{
  OHMHIP_SETTLE(m);
  if (!m || layer_id < 0 || layer_id >= OHMHIP_LID_COUNT || (count && (!keys_xyz || !srcs)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (!m->layers[layer_id])
  {
    return OHMHIP_ERR_NOT_FOUND;
  }
  if (tiledBoundary(m))
  {
    return tiledWriteRegions(m, layer_id, keys_xyz, count, srcs);
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  for (size_t i = 0; i < count && !m->precleaned.empty(); ++i)
  {
    // (the write-back's copy of a region that is being rewritten is void)
    dropPrecleanedKey(m, packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]));
  }
  OHMHIP_CHECK(readmitSpilledKeys(m, keys_xyz, count));  // (an upload edits the region where it lives: in the pool)
  OHMHIP_CHECK(makeRoomForNamedRegions(m, keys_xyz, count));
  int err = refreshHostRegionTable(m);
  if (err)
  {
    return err;
  }
  const size_t stride = size_t(m->mc.region_voxels) * kLayerBytes[layer_id];
  // Create any regions which are not resident yet (host-side insert, then rebuild the device hash).
  std::vector<uint64_t> new_keys;
  for (size_t i = 0; i < count; ++i)
  {
    const uint64_t key = packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]);
    if (m->region_slots.find(key) == m->region_slots.end())
    {
      m->region_slots[key] = uint32_t(m->slot_keys_host.size());
      m->slot_keys_host.push_back(key);
      new_keys.push_back(key);
    }
  }
  if (!new_keys.empty())
  {
    const uint32_t total = uint32_t(m->slot_keys_host.size());
    const uint32_t old = m->slots_committed;
    if (total > m->slot_capacity)
    {
      err = growPoolForNamedRegions(m, total, old);
      if (err)
      {
        dropHostRegions(m, old);  // the device never saw them
        return err;
      }
    }
    OHMHIP_CHECK(hipMemcpy(m->d_slot_keys + old, m->slot_keys_host.data() + old, sizeof(uint64_t) * (total - old),
                           hipMemcpyHostToDevice));
    // Rebuild the hash from slot_keys (cheap: one lane per region).
    OHMHIP_CHECK(hipMemsetAsync(m->d_keys, 0, sizeof(unsigned long long) * m->hash_capacity, m->stream));
    OHMHIP_CHECK(hipMemcpyAsync(m->d_n_slots, &total, sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_rehash, dim3((total + 255) / 256), dim3(256), 0, m->stream, regionTable(m), total);
    OHMHIP_CHECK(hipStreamSynchronize(m->stream));
    m->slots_committed = total;
  }
  const size_t burst = 64;
  err = ensureStage(m, std::max(m->h_stage_bytes, burst * stride));
  if (err)
  {
    return err;
  }
  for (size_t base = 0; base < count; base += burst)
  {
    const size_t n = std::min(burst, count - base);
    for (size_t k = 0; k < n; ++k)
    {
      const int16_t *key = keys_xyz + 3 * (base + k);
      const uint32_t slot = m->region_slots[packRegionKey(key[0], key[1], key[2])];
      std::memcpy(static_cast<char *>(m->h_stage) + k * stride, srcs[base + k], stride);
      OHMHIP_CHECK(hipMemcpyAsync(static_cast<char *>(m->layers[layer_id]) + size_t(slot) * stride,
                                  static_cast<char *>(m->h_stage) + k * stride, stride, hipMemcpyHostToDevice,
                                  m->copy_stream));
    }
    OHMHIP_CHECK(hipStreamSynchronize(m->copy_stream));
  }
  // NDT / TSDF keep a persistent per-voxel "ordered replay" mask derived from the stored state: rebuild it for the
  // uploaded regions when the layer that defines it was written.
  const bool ndt = m->config.mode == OHMHIP_MODE_NDT_OM || m->config.mode == OHMHIP_MODE_NDT_TM;
  const bool tsdf = m->config.mode == OHMHIP_MODE_TSDF;
  if ((ndt && layer_id == OHMHIP_LID_MEAN) || (tsdf && layer_id == OHMHIP_LID_TSDF))
  {
    for (size_t i = 0; i < count; ++i)
    {
      const uint32_t slot = m->region_slots[packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2])];
      hipLaunchKernelGGL(k_rebuild_mask, dim3(4), dim3(256), 0, m->stream, m->mc, slot,
                         ndt ? static_cast<const uint32_t *>(m->layers[OHMHIP_LID_MEAN]) : nullptr,
                         tsdf ? static_cast<const float *>(m->layers[OHMHIP_LID_TSDF]) : nullptr, m->d_hit_mask);
    }
    OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_ensure_regions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, uint32_t *slots)
try
{
  OHMHIP_SETTLE(m);
  if (!m || (count && !keys_xyz))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (tiledBoundary(m))
  {
    return OHMHIP_ERR_UNSUPPORTED;  // slots are per tile: the zero-copy / merge plumbing is for one-tile regions
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  OHMHIP_CHECK(readmitSpilledKeys(m, keys_xyz, count));
  OHMHIP_CHECK(makeRoomForNamedRegions(m, keys_xyz, count));
  int err = refreshHostRegionTable(m);
  if (err)
  {
    return err;
  }
  const uint32_t old = m->slots_committed;
  for (size_t i = 0; i < count; ++i)
  {
    const uint64_t key = packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]);
    auto it = m->region_slots.find(key);
    if (it == m->region_slots.end())
    {
      it = m->region_slots.emplace(key, uint32_t(m->slot_keys_host.size())).first;
      m->slot_keys_host.push_back(key);
    }
    if (slots)
    {
      slots[i] = it->second;
    }
  }
  const uint32_t total = uint32_t(m->slot_keys_host.size());
  if (total > old)
  {
    if (total > m->slot_capacity)
    {
      err = growPoolForNamedRegions(m, total, old);
      if (err)
      {
        dropHostRegions(m, old);  // the device never saw them
        return err;
      }
    }
    OHMHIP_CHECK(hipMemcpy(m->d_slot_keys + old, m->slot_keys_host.data() + old, sizeof(uint64_t) * (total - old),
                           hipMemcpyHostToDevice));
    OHMHIP_CHECK(hipMemsetAsync(m->d_keys, 0, sizeof(unsigned long long) * m->hash_capacity, m->stream));
    OHMHIP_CHECK(hipMemcpyAsync(m->d_n_slots, &total, sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_rehash, dim3((total + 255) / 256), dim3(256), 0, m->stream, regionTable(m), total);
    OHMHIP_CHECK(hipStreamSynchronize(m->stream));
    m->slots_committed = total;
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_remove_regions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, size_t *removed)
try
{
  if (removed)
  {
    *removed = 0;
  }
  if (!m || (count && !keys_xyz))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_SETTLE(m);
  if (tiledBoundary(m))
  {
    return tiledRemoveRegions(m, keys_xyz, count, removed);
  }
  // Regions held in the host store (spill to host) are simply forgotten.
  size_t forgotten = 0;
  for (size_t i = 0; i < count && !m->spilled.empty(); ++i)
  {
    const auto it = m->spilled.find(packRegionKey(keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2]));
    if (it != m->spilled.end())
    {
      releaseStoreRecord(m, it->second.record);
      m->spilled.erase(it);
      ++forgotten;
    }
  }
  size_t resident_removed = 0;
  const int err = removeResidentRegions(m, keys_xyz, count, &resident_removed);
  if (removed)
  {
    *removed = resident_removed + forgotten;
  }
  return err;
}
OHMHIP_ABI_CATCH

}  // extern "C"

#endif  // OHMHIP_REGION_IO_H
