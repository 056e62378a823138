// merge_impl.h -- replica merge of the occupancy layer across GPUs (include/ohmhip.h, "Replica merge").
// Included at the end of ohmhip_map.hip (it needs the map object's internals).  RCCL is linked directly: the key
// all-gather and the tile all-reduce are enqueued on the map's compute stream.
#ifndef OHMHIP_MERGE_IMPL_H
#define OHMHIP_MERGE_IMPL_H

#include <rccl/rccl.h>

#include <chrono>

struct ohmhip_comm_s
{
  ncclComm_t comm = nullptr;
  int world = 1;
  int rank = 0;
  int64_t *d_words = nullptr;    ///< [world + 2] small words of ohmhip_map_merge_replicas: counts, own count, status pair
  uint32_t *d_counts = nullptr;  ///< [world + world * world] scratch of ohmhip_comm_exchange_counts (partition_impl.h)
};

namespace
{
inline int ncclStatus(ncclResult_t r)
{
  return (r == ncclSuccess) ? OHMHIP_OK : OHMHIP_ERR_INTERNAL;
}

/// delta / observer payload of `count` regions: delta = value - base with unobserved (+inf) read as 0, 0 for a voxel
/// this replica has not observed; observers = 1 where it has.
__global__ void __launch_bounds__(256)
  k_merge_pack(const uint32_t *__restrict__ slots, uint32_t region_voxels, const float *__restrict__ occupancy,
               const float *__restrict__ base, float *__restrict__ delta, unsigned char *__restrict__ observers)
{
  const size_t in = size_t(slots[blockIdx.x]) * region_voxels;
  const size_t out = size_t(blockIdx.x) * region_voxels;
  const float inf = __int_as_float(0x7f800000);
  for (uint32_t v = threadIdx.x; v < region_voxels; v += blockDim.x)
  {
    const float x = occupancy[in + v];
    const float b = base[in + v];
    const bool observed = x != inf;
    delta[out + v] = observed ? (x - ((b != inf) ? b : 0.0f)) : 0.0f;
    observers[out + v] = observed ? 1 : 0;
  }
}

/// merged = clamp(base + sum of deltas, min, max) where any rank observed the voxel; becomes value and base.
__global__ void __launch_bounds__(256)
  k_merge_apply(const uint32_t *__restrict__ slots, uint32_t region_voxels, float *__restrict__ occupancy,
                float *__restrict__ base, const float *__restrict__ delta_sum,
                const unsigned char *__restrict__ observer_sum, float min_value, float max_value,
                uint32_t *__restrict__ dirty)
{
  const size_t out = size_t(slots[blockIdx.x]) * region_voxels;
  const size_t in = size_t(blockIdx.x) * region_voxels;
  const float inf = __int_as_float(0x7f800000);
  for (uint32_t v = threadIdx.x; v < region_voxels; v += blockDim.x)
  {
    float merged = inf;
    if (observer_sum[in + v])
    {
      const float b = base[out + v];
      merged = fminf(fmaxf(((b != inf) ? b : 0.0f) + delta_sum[in + v], min_value), max_value);
    }
    occupancy[out + v] = merged;
    base[out + v] = merged;
  }
  if (threadIdx.x == 0)
  {
    // the merged tile differs from what the host last saw; it IS the new base
    dirty[slots[blockIdx.x]] = (dirty[slots[blockIdx.x]] | kDirtySync) & ~kDirtyMerge;
  }
}

int64_t packSortable(const int16_t *k)
{
  return ((int64_t(k[0]) + 32768) << 32) | ((int64_t(k[1]) + 32768) << 16) | (int64_t(k[2]) + 32768);
}

void unpackSortable(int64_t p, int16_t *k)
{
  k[0] = int16_t((p >> 32) - 32768);
  k[1] = int16_t(((p >> 16) & 0xffff) - 32768);
  k[2] = int16_t((p & 0xffff) - 32768);
}

/// Slots of `count` regions (created when missing), uploaded to the device scratch `d_slots`.
int mergeSlots(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, DevBuf &d_slots)
{
  std::vector<uint32_t> slots(count);
  OHMHIP_CHECK(ohmhip_map_ensure_regions(m, keys_xyz, count, slots.data()));
  OHMHIP_CHECK(d_slots.ensure(sizeof(uint32_t) * std::max<size_t>(count, 1), false, m->stream));
  OHMHIP_CHECK(hipMemcpyAsync(d_slots.ptr, slots.data(), sizeof(uint32_t) * count, hipMemcpyHostToDevice, m->stream));
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));  // `slots` goes out of scope
  return OHMHIP_OK;
}
}  // namespace

extern "C" {

int ohmhip_comm_unique_id(unsigned char id[OHMHIP_COMM_ID_BYTES])
try
{
  static_assert(sizeof(ncclUniqueId) <= OHMHIP_COMM_ID_BYTES, "unique id does not fit the ABI's buffer");
  if (!id)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  ncclUniqueId uid;
  OHMHIP_CHECK(ncclStatus(ncclGetUniqueId(&uid)));
  std::memset(id, 0, OHMHIP_COMM_ID_BYTES);
  std::memcpy(id, &uid, sizeof(uid));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_comm_init_rank(ohmhip_comm_t *comm, const unsigned char id[OHMHIP_COMM_ID_BYTES], int world_size, int rank)
try
{
  if (!comm || !id || world_size < 1 || rank < 0 || rank >= world_size)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
  {
    (void)hipGetLastError();
    return OHMHIP_ERR_NO_DEVICE;
  }
  ohmhip_comm_t c = new (std::nothrow) ohmhip_comm_s;
  if (!c)
  {
    return OHMHIP_ERR_INTERNAL;
  }
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  c->world = world_size;
  c->rank = rank;
  const int err = ncclStatus(ncclCommInitRank(&c->comm, world_size, uid, rank));
  if (err)
  {
    delete c;
    return err;
  }
  // (the small device words of the collectives are allocated here, with the communicator: an allocation that fails on
  // one rank in the middle of a collective call would leave its peers blocked)
  if (hipMalloc(&c->d_words, sizeof(int64_t) * size_t(world_size + 2)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&c->d_counts), sizeof(uint32_t) * size_t(world_size) * size_t(world_size + 1)) != hipSuccess)
  {
    if (c->d_words)
    {
      (void)hipFree(c->d_words);
    }
    (void)hipGetLastError();
    (void)ncclCommDestroy(c->comm);
    delete c;
    return OHMHIP_ERR_CAPACITY;
  }
  *comm = c;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_comm_destroy(ohmhip_comm_t comm)
try
{
  if (comm)
  {
    if (comm->comm)
    {
      (void)ncclCommDestroy(comm->comm);
    }
    if (comm->d_words)
    {
      (void)hipFree(comm->d_words);
    }
    if (comm->d_counts)
    {
      (void)hipFree(comm->d_counts);
    }
    delete comm;
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_enable_merge(ohmhip_map_t m)
try
{
  OHMHIP_SETTLE(m);
  if (!m)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (m->mc.tile_split[1] > 1 || m->mc.tile_split[2] > 1)
  {
    return OHMHIP_ERR_UNSUPPORTED;  // the replica merge exchanges whole regions by slot: one-tile regions only
  }
  if (m->config.mode != OHMHIP_MODE_OCCUPANCY || !m->layers[OHMHIP_LID_OCCUPANCY])
  {
    return OHMHIP_ERR_UNSUPPORTED;  // NDT / TSDF state is not additive: replicas or region ownership
  }
  if (m->d_merge_base)
  {
    return OHMHIP_OK;
  }
  if (m->spill_enabled || !m->spilled.empty())
  {
    return OHMHIP_ERR_UNSUPPORTED;  // a merging map keeps a base copy per resident region: not combined with spilling
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  const size_t bytes = size_t(m->mc.region_voxels) * sizeof(float) * m->slot_capacity;
  if (hipMalloc(reinterpret_cast<void **>(&m->d_merge_base), bytes) != hipSuccess)
  {
    (void)hipGetLastError();
    m->d_merge_base = nullptr;
    return OHMHIP_ERR_CAPACITY;
  }
  // What the map holds now is what all replicas are taken to share.
  OHMHIP_CHECK(hipMemcpyAsync(m->d_merge_base, m->layers[OHMHIP_LID_OCCUPANCY], bytes, hipMemcpyDeviceToDevice,
                              m->stream));
  hipLaunchKernelGGL(k_and_u32, dim3(256), dim3(256), 0, m->stream, m->d_dirty, ~kDirtyMerge, size_t(m->slot_capacity));
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_merge_keys(ohmhip_map_t m, int16_t *keys_xyz, size_t capacity, size_t *count)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !count || (capacity && !keys_xyz))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (!m->d_merge_base)
  {
    return OHMHIP_ERR_INVALID_ARG;  // ohmhip_map_enable_merge first
  }
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  OHMHIP_CHECK(refreshHostRegionTable(m));
  std::vector<uint32_t> dirty(m->slots_committed);
  if (!dirty.empty())
  {
    OHMHIP_CHECK(hipMemcpy(dirty.data(), m->d_dirty, sizeof(uint32_t) * dirty.size(), hipMemcpyDeviceToHost));
  }
  size_t n = 0;
  for (size_t i = 0; i < dirty.size(); ++i)
  {
    if (dirty[i] & kDirtyMerge)
    {
      if (n < capacity)
      {
        unpackRegionKey(m->slot_keys_host[i], keys_xyz + 3 * n);
      }
      ++n;
    }
  }
  *count = n;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_merge_pack(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, float *d_delta,
                          unsigned char *d_observers)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !m->d_merge_base || (count && (!keys_xyz || !d_delta || !d_observers)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (count == 0)
  {
    return OHMHIP_OK;
  }
  OHMHIP_CHECK(mergeSlots(m, keys_xyz, count, m->merge_slots));
  hipLaunchKernelGGL(k_merge_pack, dim3(uint32_t(count)), dim3(256), 0, m->stream,
                     static_cast<const uint32_t *>(m->merge_slots.ptr), uint32_t(m->mc.region_voxels),
                     static_cast<const float *>(m->layers[OHMHIP_LID_OCCUPANCY]), m->d_merge_base, d_delta, d_observers);
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  return hipGetLastError();
}
OHMHIP_ABI_CATCH

int ohmhip_map_merge_apply(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, const float *d_delta_sum,
                           const unsigned char *d_observer_sum)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !m->d_merge_base || (count && (!keys_xyz || !d_delta_sum || !d_observer_sum)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (count == 0)
  {
    return OHMHIP_OK;
  }
  OHMHIP_CHECK(mergeSlots(m, keys_xyz, count, m->merge_slots));
  hipLaunchKernelGGL(k_merge_apply, dim3(uint32_t(count)), dim3(256), 0, m->stream,
                     static_cast<const uint32_t *>(m->merge_slots.ptr), uint32_t(m->mc.region_voxels),
                     static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]), m->d_merge_base, d_delta_sum,
                     d_observer_sum, m->mc.min_value, m->mc.max_value, m->d_dirty);
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  return hipGetLastError();
}
OHMHIP_ABI_CATCH

int ohmhip_map_merge_finish(ohmhip_map_t m)
try
{
  OHMHIP_SETTLE(m);
  if (!m || !m->d_merge_base)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  // Nothing is rebased here (round 3): `base` must stay the state ALL replicas share, so a region only this
  // rank modified keeps its shared base -- and stays pending -- until it has been exchanged (k_merge_apply clears
  // the flag).  Rebasing it locally made `base` rank-private and later exchanges of the region wrong on the peers.
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_set_merge_mode(ohmhip_map_t m, int mode)
try
{
  if (!m || (mode != OHMHIP_MERGE_SHARED_ONLY && mode != OHMHIP_MERGE_FULL_UNION))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  m->merge_mode = mode;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_merge_replicas(ohmhip_map_t m, ohmhip_comm_t comm, ohmhip_merge_stats *stats)
try
{
  const auto t_start = std::chrono::steady_clock::now();
  if (stats)
  {
    *stats = ohmhip_merge_stats{};
  }
  if (!m || !comm || !comm->comm)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  hipStream_t s = m->stream;
  const int world = comm->world;
  // One-word agreement: 0 when the step succeeded on every rank.  Every rank calls it at the same points whatever
  // happened locally (a rank that left early would leave its peers blocked in the next collective).
  auto anyRankFailed = [&](int local_err, int32_t *any_failed) -> int {
    int32_t *d_status = reinterpret_cast<int32_t *>(comm->d_words + world + 1);
    const int32_t failed = local_err ? 1 : 0;
    *any_failed = 0;
    OHMHIP_CHECK(hipMemcpyAsync(d_status, &failed, sizeof(failed), hipMemcpyHostToDevice, s));
    OHMHIP_CHECK(ncclStatus(ncclAllReduce(d_status, d_status + 1, 1, ncclInt32, ncclMax, comm->comm, s)));
    OHMHIP_CHECK(hipMemcpyAsync(any_failed, d_status + 1, sizeof(*any_failed), hipMemcpyDeviceToHost, s));
    return hipStreamSynchronize(s);
  };
  // 1. this rank's pending regions.  A failure here does not leave the call: it travels as a negative count in the
  //    first collective, so that every rank returns together (a rank that returned early would leave its peers blocked).
  size_t n_local = 0;
  int keys_err = ohmhip_map_merge_keys(m, nullptr, 0, &n_local);
  std::vector<int16_t> local_keys(3 * std::max<size_t>(keys_err ? 0 : n_local, 1));
  if (!keys_err)
  {
    keys_err = ohmhip_map_merge_keys(m, local_keys.data(), n_local, &n_local);
  }
  if (keys_err)
  {
    n_local = 0;
  }
  // 2. all-gather the key lists: counts first (one word per rank), then the lists padded to the longest.  The small
  //    words of this call live in the communicator's own scratch (allocated with it), so that no allocation -- nothing
  //    that can fail on one rank alone -- sits between two collectives without the ranks agreeing on its outcome.
  int64_t *d_counts = comm->d_words;
  const int64_t my_count = keys_err ? int64_t(-1) : int64_t(n_local);
  OHMHIP_CHECK(hipMemcpyAsync(d_counts + world, &my_count, sizeof(int64_t), hipMemcpyHostToDevice, s));
  OHMHIP_CHECK(ncclStatus(ncclAllGather(d_counts + world, d_counts, 1, ncclInt64, comm->comm, s)));
  std::vector<int64_t> counts(size_t(world), 0);
  OHMHIP_CHECK(hipMemcpyAsync(counts.data(), d_counts, sizeof(int64_t) * size_t(world), hipMemcpyDeviceToHost, s));
  OHMHIP_CHECK(hipStreamSynchronize(s));
  if (*std::min_element(counts.begin(), counts.end()) < 0)
  {
    return keys_err ? keys_err : OHMHIP_ERR_PEER;  // some rank could not list its regions: nobody merges
  }
  const size_t longest = size_t(*std::max_element(counts.begin(), counts.end()));
  size_t n_union = 0;
  std::vector<int16_t> shared_keys;
  if (longest)
  {
    std::vector<int64_t> packed(longest, -1);
    for (size_t i = 0; i < n_local; ++i)
    {
      packed[i] = packSortable(&local_keys[3 * i]);
    }
    const int alloc_err = m->merge_keys_dev.ensure(sizeof(int64_t) * longest * size_t(world + 1), false, s);
    int32_t alloc_failed = 0;
    OHMHIP_CHECK(anyRankFailed(alloc_err, &alloc_failed));
    if (alloc_failed)
    {
      return alloc_err ? alloc_err : OHMHIP_ERR_PEER;
    }
    int64_t *d_all = static_cast<int64_t *>(m->merge_keys_dev.ptr);
    int64_t *d_mine = d_all + longest * size_t(world);
    OHMHIP_CHECK(hipMemcpyAsync(d_mine, packed.data(), sizeof(int64_t) * longest, hipMemcpyHostToDevice, s));
    OHMHIP_CHECK(ncclStatus(ncclAllGather(d_mine, d_all, longest, ncclInt64, comm->comm, s)));
    std::vector<int64_t> all(longest * size_t(world));
    OHMHIP_CHECK(hipMemcpyAsync(all.data(), d_all, sizeof(int64_t) * all.size(), hipMemcpyDeviceToHost, s));
    OHMHIP_CHECK(hipStreamSynchronize(s));
    // 3. the regions more than one rank modified, in key order (the same list on every rank)
    std::vector<int64_t> valid;
    valid.reserve(all.size());
    for (int r = 0; r < world; ++r)
    {
      valid.insert(valid.end(), all.begin() + size_t(r) * longest, all.begin() + size_t(r) * longest + size_t(counts[r]));
    }
    std::sort(valid.begin(), valid.end());
    for (size_t i = 0; i < valid.size();)
    {
      size_t j = i;
      while (j < valid.size() && valid[j] == valid[i])
      {
        ++j;
      }
      ++n_union;
      if (j - i > 1 || m->merge_mode == OHMHIP_MERGE_FULL_UNION)
      {
        shared_keys.resize(shared_keys.size() + 3);
        unpackSortable(valid[i], &shared_keys[shared_keys.size() - 3]);
      }
      i = j;
    }
  }
  const size_t n_shared = shared_keys.size() / 3;
  const size_t voxels = n_shared * size_t(m->mc.region_voxels);
  // 4. the fallible local work (buffers, making the regions resident, pack), then the ranks AGREE on its outcome with
  //    a one-word all-reduce before the payload collective: a rank that failed must not leave its peers blocked in
  //    the tile all-reduce.  Every rank takes part in both collectives whatever happened locally.
  int local_err = OHMHIP_OK;
  float *d_delta = nullptr;
  unsigned char *d_obs = nullptr;
  if (n_shared)
  {
    local_err = m->merge_delta.ensure(sizeof(float) * voxels, false, s);
    if (!local_err)
    {
      local_err = m->merge_observers.ensure(voxels, false, s);
    }
    if (!local_err)
    {
      d_delta = static_cast<float *>(m->merge_delta.ptr);
      d_obs = static_cast<unsigned char *>(m->merge_observers.ptr);
      local_err = ohmhip_map_merge_pack(m, shared_keys.data(), n_shared, d_delta, d_obs);
    }
  }
  int32_t any_failed = 0;
  OHMHIP_CHECK(anyRankFailed(local_err, &any_failed));
  if (any_failed)
  {
    return local_err ? local_err : OHMHIP_ERR_PEER;  // every rank leaves here together; nothing was applied
  }
  if (n_shared)
  {
    // 5. all-reduce (float deltas summed; observer flags by max: only non-zero matters and a u8 sum would wrap at 256
    //    ranks) -> apply, all on the map's stream
    OHMHIP_CHECK(ncclStatus(ncclGroupStart()));
    OHMHIP_CHECK(ncclStatus(ncclAllReduce(d_delta, d_delta, voxels, ncclFloat, ncclSum, comm->comm, s)));
    OHMHIP_CHECK(ncclStatus(ncclAllReduce(d_obs, d_obs, voxels, ncclUint8, ncclMax, comm->comm, s)));
    OHMHIP_CHECK(ncclStatus(ncclGroupEnd()));
    OHMHIP_CHECK(ohmhip_map_merge_apply(m, shared_keys.data(), n_shared, d_delta, d_obs));
  }
  // 6. regions only this rank modified are left pending on their shared base (see ohmhip_map_merge_finish)
  OHMHIP_CHECK(ohmhip_map_merge_finish(m));
  if (stats)
  {
    stats->regions_local = uint32_t(n_local);
    stats->regions_union = uint32_t(n_union);
    stats->regions_shared = uint32_t(n_shared);
    stats->payload_bytes = uint64_t(voxels) * 5u;
    stats->key_bytes = uint64_t(sizeof(int64_t)) * (uint64_t(longest) + 1u);
    stats->ms_total =
      std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

}  // extern "C"

#endif  // OHMHIP_MERGE_IMPL_H
