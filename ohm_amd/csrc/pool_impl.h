// pool_impl.h -- the region pool: per-batch scratch views, pool allocation / growth / roll-back, the host mirror of the
// region table, the pinned host store's records, copy jobs, LDS sizing, value configuration.
//
// Part of ohmhip_map.hip's translation unit (included there, in order): not a stand-alone header.
#ifndef OHMHIP_POOL_IMPL_H
#define OHMHIP_POOL_IMPL_H

namespace
{
RegionTable regionTable(ohmhip_map_t m)
{
  RegionTable rt;
  rt.keys = m->d_keys;
  rt.vals = m->d_vals;
  rt.slot_keys = m->d_slot_keys;
  rt.n_slots = m->d_n_slots;
  rt.hash_mask = m->hash_capacity - 1;
  rt.slot_capacity = m->slot_capacity;
  return rt;
}

BatchScratch batchScratch(ohmhip_map_t m)
{
  // The counters a batch's set-up pass writes exist twice (allocPool makes the arrays twice as long): batch N+1 sets up
  // in the other half while batch N's walk / apply kernels still read theirs.
  const size_t h = size_t(m->parity) * m->hash_capacity;
  const size_t c = size_t(m->parity) * m->slot_capacity;
  BatchScratch bs;
  bs.seg_count = m->d_seg_count + h;
  bs.seg_cursor = m->d_seg_cursor + h;
  bs.seg_offset = m->d_seg_offset + h;
  bs.touched_flag = m->d_touched_flag + h;
  bs.touched = m->d_touched + h;
  bs.hit_count = m->d_hit_count + h;
  // (three lists per parity in one allocation: regions receiving samples, regions whose counts / whose samples the apply
  // kernels still have to process -- k_plan)
  bs.sort_list = m->d_sort_list + 3 * h;
  bs.apply_counts_list = bs.sort_list + m->hash_capacity;
  bs.apply_hits_list = bs.sort_list + 2 * size_t(m->hash_capacity);
  bs.voxel_first_hit = m->d_voxel_first_hit;
  bs.hit_begin = m->d_hit_begin + c;
  bs.hit_end = m->d_hit_end + c;
  bs.dirty = m->d_dirty;
  bs.last_use = m->d_last_use;
  bs.stamp = uint32_t(m->batch_seq + 1u);
  bs.info = m->d_info + m->info_index;
  bs.wg_regions = static_cast<WgRegion *>(m->wg_regions[m->parity].ptr);
  bs.wg_region_count = static_cast<uint32_t *>(m->wg_region_count[m->parity].ptr);
  return bs;
}

inline Chunk *batchChunks(ohmhip_map_t m) { return m->d_chunks + size_t(m->parity) * m->chunk_capacity; }
inline uint32_t *batchEventCount(ohmhip_map_t m) { return m->d_event_count + 4u * m->parity; }
inline DevBuf &batchWalks(ohmhip_map_t m) { return m->walks_buf[m->parity]; }

__global__ void k_rehash(RegionTable rt, uint32_t n)
{
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n)
  {
    return;
  }
  const uint64_t key = rt.slot_keys[slot];
  uint32_t idx = hashRegionKey(key, rt.hash_mask);
  while (true)
  {
    const unsigned long long prev = atomicCAS(&rt.keys[idx], 0ull, (unsigned long long)key);
    if (prev == 0)
    {
      rt.vals[idx] = slot;
      return;
    }
    idx = (idx + 1) & rt.hash_mask;
  }
}

uint32_t nextPow2(uint32_t v)
{
  uint32_t p = 1;
  while (p < v)
  {
    p <<= 1;
  }
  return p;
}

size_t bytesPerRegionAllLayers(const ohmhip_map_config &c, int region_voxels)
{
  size_t b = 0;
  for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
  {
    if (c.layers & (1u << l))
    {
      b += kLayerBytes[l] * size_t(region_voxels);
    }
  }
  // + miss count layer + hit mask
  b += 4 * size_t(region_voxels) + size_t((region_voxels + 31) / 32) * 4;
  // + first-sample table (occupancy mode), traversal accumulator (traversal layer)
  b += (c.mode == OHMHIP_MODE_OCCUPANCY) ? 4 * size_t(region_voxels) : 0;
  b += (c.layers & (1u << OHMHIP_LID_TRAVERSAL)) ? 8 * size_t(region_voxels) : 0;
  return b;
}

void freePool(ohmhip_map_t m)
{
  for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
  {
    if (m->layers[l])
    {
      (void)hipFree(m->layers[l]);
      m->layers[l] = nullptr;
    }
  }
  void *ptrs[] = { m->d_keys,       m->d_vals,        m->d_slot_keys, m->d_seg_count, m->d_seg_cursor,
                   m->d_seg_offset, m->d_touched_flag, m->d_touched,   m->d_voxel_first_hit, m->d_hit_begin, m->d_hit_end,
                   m->d_dirty,      m->d_miss_counts,  m->d_hit_mask,  m->d_chunks,    m->d_hit_count, m->d_sort_list,
                   m->d_last_use };
  for (void *p : ptrs)
  {
    if (p)
    {
      (void)hipFree(p);
    }
  }
  m->d_keys = nullptr;
  m->d_vals = nullptr;
  m->d_slot_keys = nullptr;
  m->d_seg_count = m->d_seg_cursor = m->d_seg_offset = m->d_touched_flag = m->d_touched = nullptr;
  m->d_voxel_first_hit = m->d_hit_begin = m->d_hit_end = m->d_dirty = nullptr;
  m->d_last_use = nullptr;
  m->d_miss_counts = m->d_hit_mask = nullptr;
  m->d_chunks = nullptr;
  m->d_hit_count = m->d_sort_list = nullptr;
  if (m->d_merge_base)
  {
    (void)hipFree(m->d_merge_base);
    m->d_merge_base = nullptr;
  }
  if (m->d_traversal_acc)
  {
    (void)hipFree(m->d_traversal_acc);
    m->d_traversal_acc = nullptr;
  }
}

/// (Re)allocate the region pool for `capacity` regions, preserving the first `keep` slots' contents.  Everything new is
/// allocated before anything old is released: a failed allocation leaves the map exactly as it was.
int allocPool(ohmhip_map_t m, uint32_t capacity, uint32_t keep)
{
  const size_t rv = size_t(m->mc.region_voxels);
  const uint32_t hash_cap = nextPow2(std::max<uint32_t>(1024u, capacity * 2u));
  hipStream_t s = m->stream;
  if (!m->precleaned.empty() || !m->stale_records.empty())
  {
    OHMHIP_CHECK(drainWriteBack(m));  // background write-back copies read the pool being replaced
  }

  std::vector<void *> fresh;  // released again if any step fails
  auto alloc = [&](void **p, size_t bytes) -> int {
    *p = nullptr;
    const int err = int(hipMalloc(p, std::max<size_t>(bytes, 4)));
    if (err == 0)
    {
      fresh.push_back(*p);
    }
    return err;
  };
  auto zalloc = [&](void **p, size_t bytes) -> int {
    OHMHIP_CHECK(alloc(p, bytes));
    OHMHIP_CHECK(hipMemsetAsync(*p, 0, std::max<size_t>(bytes, 4), s));
    return OHMHIP_OK;
  };
  void *new_layers[OHMHIP_LID_COUNT] = {};
  uint64_t *new_slot_keys = nullptr;
  uint32_t *new_mask = nullptr, *new_dirty = nullptr, *new_last_use = nullptr;
  unsigned long long *n_keys = nullptr;
  uint32_t *n_vals = nullptr, *n_seg_count = nullptr, *n_seg_cursor = nullptr, *n_hit_count = nullptr,
           *n_sort_list = nullptr, *n_seg_offset = nullptr, *n_touched_flag = nullptr, *n_touched = nullptr,
           *n_first_hit = nullptr, *n_hit_begin = nullptr, *n_hit_end = nullptr, *n_miss_counts = nullptr;
  Chunk *n_chunks = nullptr;
  float *n_merge_base = nullptr;
  unsigned long long *n_traversal_acc = nullptr;
  const uint32_t chunk_capacity = capacity + (1u << 16);
  // The per-voxel mask is persistent state for NDT / TSDF (voxels that take the ordered replay path): it moves with
  // the regions it describes.
  const size_t mask_row = ((rv + 31) / 32) * sizeof(uint32_t);
  auto build = [&]() -> int {
    for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
    {
      if (!(m->config.layers & (1u << l)))
      {
        continue;
      }
      const size_t stride = rv * kLayerBytes[l];
      OHMHIP_CHECK(alloc(&new_layers[l], stride * capacity));
      if (keep && m->layers[l])
      {
        OHMHIP_CHECK(hipMemcpyAsync(new_layers[l], m->layers[l], stride * keep, hipMemcpyDeviceToDevice, s));
      }
      char *tail = static_cast<char *>(new_layers[l]) + stride * keep;
      const size_t tail_bytes = stride * (capacity - keep);
      if (l == OHMHIP_LID_OCCUPANCY)
      {
        // Occupancy clears to +inf == unobserved (ohm/DefaultLayer.cpp:87-91, ohm/VoxelOccupancy.h:42-45).
        const size_t count = tail_bytes / 4;
        if (count)
        {
          hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, s, reinterpret_cast<uint32_t *>(tail), 0x7f800000u,
                             count);
        }
      }
      else if (tail_bytes)
      {
        OHMHIP_CHECK(hipMemsetAsync(tail, 0, tail_bytes, s));
      }
    }
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&new_slot_keys), sizeof(uint64_t) * capacity));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&new_mask), mask_row * capacity));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&new_dirty), sizeof(uint32_t) * capacity));
    if (keep && m->d_slot_keys)
    {
      OHMHIP_CHECK(hipMemcpyAsync(new_slot_keys, m->d_slot_keys, sizeof(uint64_t) * keep, hipMemcpyDeviceToDevice, s));
    }
    if (keep && m->d_hit_mask)
    {
      OHMHIP_CHECK(hipMemcpyAsync(new_mask, m->d_hit_mask, mask_row * keep, hipMemcpyDeviceToDevice, s));
    }
    if (keep && m->d_dirty)
    {
      OHMHIP_CHECK(hipMemcpyAsync(new_dirty, m->d_dirty, sizeof(uint32_t) * keep, hipMemcpyDeviceToDevice, s));
    }
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&new_last_use), sizeof(uint32_t) * 2 * capacity));
    if (keep && m->d_last_use)
    {
      OHMHIP_CHECK(hipMemcpyAsync(new_last_use, m->d_last_use, sizeof(uint32_t) * 2 * keep, hipMemcpyDeviceToDevice, s));
    }
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_keys), sizeof(unsigned long long) * hash_cap));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_vals), sizeof(uint32_t) * hash_cap));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_seg_count), sizeof(uint32_t) * 2 * hash_cap));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_seg_cursor), sizeof(uint32_t) * 2 * hash_cap));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_hit_count), sizeof(uint32_t) * 2 * hash_cap));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_sort_list), sizeof(uint32_t) * 6 * hash_cap));  // (3 lists x 2 parities)
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_seg_offset), sizeof(uint32_t) * 2 * hash_cap));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_touched_flag), sizeof(uint32_t) * 2 * hash_cap));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_touched), sizeof(uint32_t) * 2 * hash_cap));
    if (m->config.mode == OHMHIP_MODE_OCCUPANCY)
    {
      OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_first_hit), sizeof(uint32_t) * rv * capacity));
    }
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_hit_begin), sizeof(uint32_t) * 2 * capacity));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_hit_end), sizeof(uint32_t) * 2 * capacity));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_miss_counts), sizeof(uint32_t) * rv * capacity));
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_chunks), sizeof(Chunk) * 2 * chunk_capacity));
    if (m->config.layers & (1u << OHMHIP_LID_TRAVERSAL))
    {
      OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_traversal_acc), sizeof(unsigned long long) * rv * capacity));
    }
    if (m->d_merge_base)
    {
      // replica-merge base (merge_impl.h): moves with the regions; a new region's base is "unobserved"
      OHMHIP_CHECK(alloc(reinterpret_cast<void **>(&n_merge_base), sizeof(float) * rv * capacity));
      if (keep)
      {
        OHMHIP_CHECK(hipMemcpyAsync(n_merge_base, m->d_merge_base, sizeof(float) * rv * keep, hipMemcpyDeviceToDevice, s));
      }
      if (capacity > keep)
      {
        hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, s, reinterpret_cast<uint32_t *>(n_merge_base + rv * keep),
                           0x7f800000u, rv * (capacity - keep));
      }
    }
    OHMHIP_CHECK(hipStreamSynchronize(s));
    return OHMHIP_OK;
  };
  const int build_err = build();
  if (build_err)
  {
    (void)hipStreamSynchronize(s);
    for (void *p : fresh)
    {
      (void)hipFree(p);
    }
    (void)hipGetLastError();
    return (build_err == int(hipErrorOutOfMemory)) ? int(OHMHIP_ERR_CAPACITY) : build_err;
  }

  // Swap in.
  freePool(m);
  for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
  {
    m->layers[l] = new_layers[l];
  }
  m->d_slot_keys = new_slot_keys;
  m->d_hit_mask = new_mask;
  m->d_dirty = new_dirty;
  m->d_last_use = new_last_use;
  m->d_keys = n_keys;
  m->d_vals = n_vals;
  m->d_seg_count = n_seg_count;
  m->d_seg_cursor = n_seg_cursor;
  m->d_hit_count = n_hit_count;
  m->d_sort_list = n_sort_list;
  m->d_seg_offset = n_seg_offset;
  m->d_touched_flag = n_touched_flag;
  m->d_touched = n_touched;
  m->d_voxel_first_hit = n_first_hit;
  m->d_hit_begin = n_hit_begin;
  m->d_hit_end = n_hit_end;
  m->d_miss_counts = n_miss_counts;
  m->d_chunks = n_chunks;
  m->d_merge_base = n_merge_base;
  m->d_traversal_acc = n_traversal_acc;
  m->chunk_capacity = chunk_capacity;
  m->slot_capacity = capacity;
  m->hash_capacity = hash_cap;
  OHMHIP_CHECK(hipMemcpyAsync(m->d_n_slots, &keep, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  if (keep)
  {
    hipLaunchKernelGGL(k_rehash, dim3((keep + 255) / 256), dim3(256), 0, s, regionTable(m), keep);
  }
  OHMHIP_CHECK(hipStreamSynchronize(s));
  OHMHIP_CHECK(hipGetLastError());
  return OHMHIP_OK;
}

/// Largest region pool the 20-bit slot field of the sample / event sort keys can address.
constexpr uint32_t kMaxRegionSlots = (1u << 20) - 2u;

/// Pool capacity for `needed` regions: doubling, clamped to what the sort keys can address.  False when `needed` itself
/// is beyond that (the caller reports OHMHIP_ERR_CAPACITY: a larger slot would be truncated in the keys and alias
/// another region).
bool grownCapacity(uint32_t current, uint32_t needed, uint32_t &capacity)
{
  if (needed > kMaxRegionSlots)
  {
    return false;
  }
  uint64_t cap = std::max<uint32_t>(current, 1u);
  while (cap < needed)
  {
    cap *= 2;
  }
  capacity = uint32_t(std::min<uint64_t>(cap, kMaxRegionSlots));
  return true;
}

/// Forget the regions a failed write_regions / ensure_regions call added to the host table.
void dropHostRegions(ohmhip_map_t m, size_t keep)
{
  for (size_t i = keep; i < m->slot_keys_host.size(); ++i)
  {
    m->region_slots.erase(m->slot_keys_host[i]);
  }
  m->slot_keys_host.resize(keep);
}

/// Forget what a failed batch's set-up pass left in the region table and the per-batch scratch, without touching the
/// pool: the hash is rebuilt from the committed slots.  (Used when the pool may not grow.)
int rollbackTable(ohmhip_map_t m)
{
  hipStream_t s = m->stream;
  const size_t hash_words = m->hash_capacity;
  OHMHIP_CHECK(hipStreamSynchronize(s));
  OHMHIP_CHECK(hipMemsetAsync(m->d_keys, 0, sizeof(unsigned long long) * hash_words, s));
  OHMHIP_CHECK(hipMemsetAsync(m->d_vals, 0, sizeof(uint32_t) * hash_words, s));
  uint32_t *per_hash[] = { m->d_seg_count,  m->d_seg_cursor,   m->d_hit_count, m->d_seg_offset,
                           m->d_touched_flag, m->d_touched };
  for (uint32_t *p : per_hash)
  {
    OHMHIP_CHECK(hipMemsetAsync(p, 0, sizeof(uint32_t) * 2 * hash_words, s));  // (both parities)
  }
  OHMHIP_CHECK(hipMemsetAsync(m->d_sort_list, 0, sizeof(uint32_t) * 6 * hash_words, s));  // (3 lists x 2 parities)
  const uint32_t keep = m->slots_committed;
  if (m->slot_capacity > keep)
  {
    OHMHIP_CHECK(hipMemsetAsync(m->d_slot_keys + keep, 0, sizeof(uint64_t) * (m->slot_capacity - keep), s));
  }
  if (m->slot_capacity > keep)
  {
    // the slots the failed batch handed out go back to the pristine state: no modified flags, no use stamp
    OHMHIP_CHECK(hipMemsetAsync(m->d_dirty + keep, 0, sizeof(uint32_t) * (m->slot_capacity - keep), s));
    OHMHIP_CHECK(hipMemsetAsync(m->d_last_use + 2 * size_t(keep), 0, sizeof(uint32_t) * 2 * (m->slot_capacity - keep), s));
  }
  OHMHIP_CHECK(hipMemcpyAsync(m->d_n_slots, &keep, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  if (keep)
  {
    hipLaunchKernelGGL(k_rehash, dim3((keep + 255) / 256), dim3(256), 0, s, regionTable(m), keep);
  }
  OHMHIP_CHECK(hipMemsetAsync(m->d_info, 0, 3 * sizeof(BatchInfo), s));
  m->info_clean = false;
  m->spec_bucket_ok = false;
  OHMHIP_CHECK(hipStreamSynchronize(s));
  return hipGetLastError();
}

/// Restore the region table after a batch that overflowed the pool: drop regions the failed batch inserted.
/// `needed`: the slots the batch must have; the pool is at least doubled beyond that where it may (amortised growth).
int rollbackAndGrow(ohmhip_map_t m, uint32_t needed)
{
  uint32_t cap = 0;
  const uint32_t wish = std::max(needed, std::min(m->slot_capacity * 2u, kMaxRegionSlots));
  if (!grownCapacity(m->slot_capacity, needed, cap))
  {
    return OHMHIP_ERR_CAPACITY;
  }
  uint32_t wished_cap = cap;
  if (grownCapacity(m->slot_capacity, wish, wished_cap))
  {
    cap = wished_cap;
  }
  // Check memory budget: refuse if the new pool cannot fit in free device memory, or in the map's own limit (the
  // largest pool the limit allows is still tried when doubling overshoots it).
  const size_t per_region = bytesPerRegionAllLayers(m->config, m->mc.region_voxels);
  if (m->memory_limit)
  {
    const uint64_t allowed = m->memory_limit / per_region;
    if (allowed < needed)
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
  return allocPool(m, cap, m->slots_committed);
}

int refreshHostRegionTable(ohmhip_map_t m)
{
  const uint32_t n = m->slots_committed;
  if (m->slot_keys_host.size() == n)
  {
    return OHMHIP_OK;
  }
  const size_t old = m->slot_keys_host.size();
  m->slot_keys_host.resize(n);
  if (n > old)
  {
    OHMHIP_CHECK(hipMemcpy(m->slot_keys_host.data() + old, m->d_slot_keys + old, sizeof(uint64_t) * (n - old),
                           hipMemcpyDeviceToHost));
    for (size_t i = old; i < n; ++i)
    {
      m->region_slots[m->slot_keys_host[i]] = uint32_t(i);
    }
  }
  return OHMHIP_OK;
}

int ensureStage(ohmhip_map_t m, size_t bytes)
{
  if (bytes <= m->h_stage_bytes)
  {
    return OHMHIP_OK;
  }
  if (m->h_stage)
  {
    OHMHIP_CHECK(hipHostFree(m->h_stage));
    m->h_stage = nullptr;
    m->h_stage_bytes = 0;
  }
  OHMHIP_CHECK(hipHostMalloc(&m->h_stage, bytes, hipHostMallocDefault));
  m->h_stage_bytes = bytes;
  return OHMHIP_OK;
}

/// Lay out the host store's records for this map's layer set (once) and make sure at least `records` are free.
int reserveStoreRecords(ohmhip_map_t m, size_t records)
{
  ohmhip_map_s::HostStore &st = m->store;
  if (st.record_bytes == 0)
  {
    const size_t rv = size_t(m->mc.region_voxels);
    size_t at = 0;
    for (int l = 0; l < OHMHIP_LID_COUNT; ++l)
    {
      st.layer_offset[l] = at;
      if (m->layers[l])
      {
        at += (rv * kLayerBytes[l] + 255) & ~size_t(255);
      }
    }
    st.mask_offset = at;
    st.mask_bytes = ((rv + 31) / 32) * sizeof(uint32_t);
    at += (st.mask_bytes + 255) & ~size_t(255);
    st.record_bytes = at;
  }
  while (st.free_records.size() < records)
  {
    // slabs of about 64 MiB, at least the shortfall (one pinning call for a large reservation)
    const size_t want = std::max<size_t>(records - st.free_records.size(), (size_t(64) << 20) / st.record_bytes + 1);
    void *slab = nullptr;
    if (hipHostMalloc(&slab, want * st.record_bytes, hipHostMallocDefault) != hipSuccess)
    {
      (void)hipGetLastError();
      return OHMHIP_ERR_CAPACITY;
    }
    st.slabs.push_back(slab);
    for (size_t i = 0; i < want; ++i)
    {
      st.free_records.push_back(static_cast<char *>(slab) + i * st.record_bytes);
    }
    st.records_total += want;
  }
  return OHMHIP_OK;
}

char *takeStoreRecord(ohmhip_map_t m)
{
  if (m->store.free_records.empty() && reserveStoreRecords(m, 1) != OHMHIP_OK)
  {
    return nullptr;
  }
  char *rec = m->store.free_records.back();
  m->store.free_records.pop_back();
  return rec;
}

void releaseStoreRecord(ohmhip_map_t m, char *record)
{
  if (record)
  {
    m->store.free_records.push_back(record);
  }
}

void freeHostStore(ohmhip_map_t m)
{
  for (void *slab : m->store.slabs)
  {
    (void)hipHostFree(slab);
  }
  m->store = ohmhip_map_s::HostStore{};
}

/// Run a list of byte copies as one kernel on `stream` (k_copy_jobs); returns with the launch queued.
int launchCopyJobs(ohmhip_map_t m, const std::vector<CopyJob> &jobs, hipStream_t stream)
{
  if (jobs.empty())
  {
    return OHMHIP_OK;
  }
  OHMHIP_CHECK(m->copy_jobs.ensure(sizeof(CopyJob) * jobs.size(), false, stream));
  OHMHIP_CHECK(hipMemcpy(m->copy_jobs.ptr, jobs.data(), sizeof(CopyJob) * jobs.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_copy_jobs, dim3(uint32_t(jobs.size()) * kCopyBlocksPerJob), dim3(256), 0, stream,
                     static_cast<const CopyJob *>(m->copy_jobs.ptr), uint32_t(jobs.size()));
  return hipGetLastError();
}

/// Highest key bit the sorts need: the slot field only uses log2(slots) + 1 bits (invalid keys are all ones).  `slots`:
/// the pool's capacity when sizing buffers, the slots actually in use when sorting (fewer 8-bit passes for a map that
/// occupies a small part of a large pool).
unsigned sortEndBit(uint32_t slots)
{
  unsigned bits = 1;
  while ((1u << bits) <= slots)
  {
    ++bits;
  }
  return std::min<unsigned>(64u, unsigned(kHitSlotShift) + bits + 1u);
}
unsigned sortEndBit(ohmhip_map_t m) { return sortEndBit(m->slot_capacity); }

/// rocPRIM falls back to a 20-launch merge sort for up to 2^20 keys by default; the one-sweep radix path is several
/// times faster on the 1M-key sample lists of a typical batch.
using SortConfig =
  rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, size_t(1) << 15>;

constexpr size_t kDbgWords = 16 + size_t(kTraceChunks) * kTraceWords;

template <typename G>
size_t walkLdsBytesOf(const MapConst &mc, uint32_t chunk_segments)
{
  // [count tile, padded to 16 B][per-wave queues][staged sample keys][interval counters][cursor + pad]
  // [length histogram][segment order, u16 each]
  const size_t count_words = (size_t((mc.region_voxels + 1) / 2) + 31u) & ~size_t(31);  // whole 32-word rows (tileWord)
  return (count_words + size_t(2 * G::kWaves * G::kQueueCap) + size_t(2 * G::kLdsHits) + size_t(G::kLdsHits / 2) +
          kWalkCursorWords + 64 + (kIndexBuckets + 2) / 2 +
          kLengthClasses + (chunk_segments + 1) / 2) *
         sizeof(uint32_t);
}

size_t walkLdsBytes(const MapConst &mc, uint32_t chunk_segments, bool half = false)
{
  return half ? walkLdsBytesOf<WalkHalf>(mc, chunk_segments) : walkLdsBytesOf<WalkFull>(mc, chunk_segments);
}

__global__ void k_clear_counts(MapConst mc, RegionTable rt, BatchScratch bs, uint32_t *__restrict__ miss_counts)
{
  const uint32_t slot = rt.vals[bs.touched[blockIdx.x]];
  const size_t base = size_t(slot) * size_t(mc.region_voxels);
  for (uint32_t vi = threadIdx.x; vi < uint32_t(mc.region_voxels); vi += blockDim.x)
  {
    miss_counts[base + vi] = 0;
  }
}

/// One ray batch through the pipeline (all map modes).  d_rays: device pointer to 6 doubles per ray.
/// The value half of the configuration (probabilities, clamps, filter, NDT / TSDF parameters): everything a host map can
/// change between batches.  Geometry, mode and the layer set are fixed at creation.
void applyValueConfig(ohmhip_map_t m)
{
  MapConst &mc = m->mc;
  mc.hit_value = m->config.hit_value;
  mc.miss_value = m->config.miss_value;
  mc.threshold_value = m->config.threshold_value;
  mc.min_value = m->config.min_value;
  mc.max_value = m->config.max_value;
  // ohm/RayMapperOccupancy.cpp:92-93
  mc.sat_min = m->config.saturate_at_min ? mc.min_value : std::numeric_limits<float>::lowest();
  mc.sat_max = m->config.saturate_at_max ? mc.max_value : std::numeric_limits<float>::max();
  mc.filter_mode = m->config.ray_filter;
  mc.filter_range = m->config.ray_filter_range;
  mc.sensor_noise = m->config.ndt_sensor_noise;
  mc.sample_threshold = m->config.ndt_sample_threshold;
  mc.adaptation_rate = m->config.ndt_adaptation_rate;
  mc.reinit_threshold = m->config.ndt_reinit_threshold;
  mc.reinit_count = m->config.ndt_reinit_count;
  mc.initial_intensity_cov = m->config.ndt_initial_intensity_cov;
  mc.tsdf_max_weight = m->config.tsdf_max_weight;
  mc.tsdf_trunc = m->config.tsdf_trunc;
  mc.tsdf_dropoff = m->config.tsdf_dropoff;
  mc.tsdf_sparsity = m->config.tsdf_sparsity;

}

}  // namespace

#endif  // OHMHIP_POOL_IMPL_H
