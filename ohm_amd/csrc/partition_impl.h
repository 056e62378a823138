// partition_impl.h -- the region-partitioned map across GPUs (include/ohmhip.h, "Partitioned map"; no
// reference equivalent: ohm is single device).  Included at the end of ohmhip_map.hip, after merge_impl.h (it needs the
// map object's internals and the RCCL communicator).
//
// Every rank owns a territory of region blocks (a table dealt by the host, or the block hash).  A rank's rays are
// ROUTED: for each ray the exact set of regions its walk touches is enumerated with the very functions the integration
// uses (setupRay + forEachSegment), the owners of those regions are the ray's destinations, and the rays are compacted
// per destination in ray order.  The ranks exchange the routed rays (48 B each -- voxels never travel) and every rank
// integrates the rays addressed to it in (source rank, ray) order with the ownership filter on.  Because a voxel's
// update sequence depends only on the rays that reach it, in order, the union of the ranks' regions is bit-identical to
// ONE map integrating rank 0's batch, then rank 1's, ... (tests/test_gpu_partitioned.py, C4 at full size).
#ifndef OHMHIP_PARTITION_IMPL_H
#define OHMHIP_PARTITION_IMPL_H

namespace
{
constexpr int kRouteThreads = 256;
constexpr uint32_t kRouteMaxWorld = 64;  ///< destinations of a ray are a 64-bit mask

/// Destination mask per ray + per-workgroup destination counts.
__global__ void __launch_bounds__(kRouteThreads)
  k_route_mask(MapConst mc, const double *__restrict__ rays, uint32_t n_rays, unsigned ray_flags,
               unsigned long long *__restrict__ masks, uint32_t *__restrict__ block_counts, uint32_t world,
               unsigned long long *__restrict__ visits_out)
{
  __shared__ uint32_t s_counts[kRouteMaxWorld];
  __shared__ unsigned long long s_visits;
  if (threadIdx.x < kRouteMaxWorld)
  {
    s_counts[threadIdx.x] = 0;
  }
  if (threadIdx.x == 0)
  {
    s_visits = 0;
  }
  __syncthreads();
  const uint32_t ray = blockIdx.x * kRouteThreads + threadIdx.x;
  unsigned long long mask = 0;
  unsigned long long visits = 0;
  if (ray < n_rays)
  {
    RayWalk rw;
    double start[3], end[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      start[a] = rays[size_t(ray) * 6 + a];
      end[a] = rays[size_t(ray) * 6 + 3 + a];
    }
    setupRay(mc, start, end, ray_flags, rw, ray);
    if (rw.flags & kRwValid)
    {
      // every region the walk touches, whoever owns it (the enumeration's own ownership filter switched off)
      MapConst all = mc;
      all.owner_world = 0;
      forEachSegment(all, rw, false, [&](uint64_t key, const SegmentEntry &) { mask |= 1ull << regionOwnerOf(mc, key); });
      if (rw.flags & kRwApplySample)
      {
        uint64_t key;
        uint32_t vi;
        sampleVoxel(mc, rw, key, vi);
        mask |= 1ull << regionOwnerOf(mc, key);
      }
      // the ray's voxel visits (what k_ray_setup accumulates into ohmhip_batch_stats::visits)
      const int manhattan = rw.total[0] + rw.total[1] + rw.total[2];
      if (rw.flags & kRwWalk)
      {
        visits += (unsigned long long)manhattan;
        visits -= ((rw.flags & kRwExcludeStart) && manhattan > 0) ? 1u : 0u;
        visits += (rw.flags & kRwIncludeEnd) ? 1u : 0u;
      }
      visits += (rw.flags & kRwApplySample) ? 1u : 0u;
    }
    masks[ray] = mask;
  }
  for (uint32_t d = 0; d < world; ++d)
  {
    const unsigned long long b = __ballot((mask >> d) & 1ull);
    if ((threadIdx.x & 63u) == 0 && b)
    {
      atomicAdd(&s_counts[d], uint32_t(__popcll(b)));
    }
  }
  if (visits)
  {
    atomicAdd(&s_visits, visits);
  }
  __syncthreads();
  if (threadIdx.x < world)
  {
    block_counts[size_t(threadIdx.x) * gridDim.x + blockIdx.x] = s_counts[threadIdx.x];
  }
  if (threadIdx.x == 0 && s_visits)
  {
    atomicAdd(visits_out, s_visits);
  }
}

/// Exclusive scan of one destination's per-workgroup counts (one workgroup per destination); the destination's total
/// goes to device and (pinned) host memory.
__global__ void __launch_bounds__(1024)
  k_route_scan(uint32_t *__restrict__ block_counts, uint32_t n_blocks, uint32_t *__restrict__ totals,
               uint32_t *__restrict__ host_totals)
{
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_base;
  uint32_t *counts = block_counts + size_t(blockIdx.x) * n_blocks;
  if (threadIdx.x == 0)
  {
    s_base = 0;
  }
  __syncthreads();
  for (uint32_t first = 0; first < n_blocks; first += 1024)
  {
    const uint32_t i = first + threadIdx.x;
    const uint32_t c = (i < n_blocks) ? counts[i] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t up = __shfl_up(incl, d);
      incl += (int(threadIdx.x & 63u) >= d) ? up : 0u;
    }
    if ((threadIdx.x & 63u) == 63u)
    {
      s_wave[threadIdx.x >> 6] = incl;
    }
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w)
    {
      const uint32_t v = s_wave[w];
      before += (w < (threadIdx.x >> 6)) ? v : 0u;
      all += v;
    }
    if (i < n_blocks)
    {
      counts[i] = s_base + before + incl - c;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
      s_base += all;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    totals[blockIdx.x] = s_base;
    host_totals[blockIdx.x] = s_base;
    __threadfence_system();
  }
}

/// Compact the rays per destination, in ray order: destination d's rays at [sum of totals[< d], ...).
__global__ void __launch_bounds__(kRouteThreads)
  k_route_scatter(const double *__restrict__ rays, uint32_t n_rays, const unsigned long long *__restrict__ masks,
                  const uint32_t *__restrict__ block_offsets, const uint32_t *__restrict__ totals, uint32_t world,
                  double *__restrict__ out_rays, uint32_t *__restrict__ out_index, uint32_t capacity)
{
  __shared__ uint32_t s_wave[kRouteThreads / 64];
  const uint32_t ray = blockIdx.x * kRouteThreads + threadIdx.x;
  const unsigned long long mask = (ray < n_rays) ? masks[ray] : 0ull;
  double r[6] = { 0, 0, 0, 0, 0, 0 };
  if (mask)
  {
#pragma unroll
    for (int a = 0; a < 6; ++a)
    {
      r[a] = rays[size_t(ray) * 6 + a];
    }
  }
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = threadIdx.x >> 6;
  uint32_t dest_base = 0;
  for (uint32_t d = 0; d < world; ++d)
  {
    const bool mine = ((mask >> d) & 1ull) != 0;
    const unsigned long long b = __ballot(mine);
    if (lane == 0)
    {
      s_wave[wave] = uint32_t(__popcll(b));
    }
    __syncthreads();
    uint32_t before = 0;
    for (unsigned w = 0; w < wave; ++w)
    {
      before += s_wave[w];
    }
    __syncthreads();
    if (mine)
    {
      const uint32_t pos = dest_base + block_offsets[size_t(d) * gridDim.x + blockIdx.x] + before +
                           uint32_t(__popcll(b & ((1ull << lane) - 1ull)));
      if (pos < capacity)
      {
#pragma unroll
        for (int a = 0; a < 6; ++a)
        {
          out_rays[size_t(pos) * 6 + a] = r[a];
        }
        if (out_index)
        {
          out_index[pos] = ray;
        }
      }
    }
    dest_base += totals[d];
  }
}

/// The partition table of `m` on the host (empty: hash rule).
uint32_t hostRegionOwner(ohmhip_map_t m, const int16_t *key)
{
  const uint32_t world = std::max(1u, m->mc.owner_world);
  if (world <= 1u)
  {
    return 0u;
  }
  return partitionOwner(m->partition.table_host.empty() ? nullptr : m->partition.table_host.data(),
                        m->mc.owner_grid_origin, m->mc.owner_grid_dims, m->mc.owner_shift, world, key[0], key[1], key[2]);
}
}  // namespace

extern "C" {

int ohmhip_map_set_region_partition(ohmhip_map_t m, const ohmhip_partition *p)
try
{
  if (!m || !p)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const bool on = p->world_size > 1;
  const size_t cells = size_t(p->grid_dims[0]) * size_t(p->grid_dims[1]) * size_t(p->grid_dims[2]);
  if (on && (p->rank >= p->world_size || p->world_size > kRouteMaxWorld || p->block_shift < 0 || p->block_shift > 15 ||
             (cells != 0 && !p->owners) || cells > (size_t(1) << 26) ||
             (cells == 0 && p->owners != nullptr)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (on && p->owners)
  {
    for (size_t i = 0; i < cells; ++i)
    {
      if (p->owners[i] >= p->world_size)
      {
        return OHMHIP_ERR_INVALID_ARG;
      }
    }
  }
  OHMHIP_CHECK(ohmhip_map_sync(m));
  if (m->slots_committed != 0 || !m->spilled.empty())
  {
    return OHMHIP_ERR_INVALID_ARG;  // regions integrated under another partition would be left behind
  }
  m->partition.table_host.clear();
  m->mc.owner_table = nullptr;
  m->mc.owner_world = on ? p->world_size : 0u;
  m->mc.owner_rank = on ? p->rank : 0u;
  m->mc.owner_shift = on ? p->block_shift : 0;
  for (int a = 0; a < 3; ++a)
  {
    m->mc.owner_grid_origin[a] = 0;
    m->mc.owner_grid_dims[a] = 0;
  }
  if (on && cells)
  {
    OHMHIP_CHECK(m->partition.table_dev.ensure(cells, false, m->stream));
    OHMHIP_CHECK(hipMemcpy(m->partition.table_dev.ptr, p->owners, cells, hipMemcpyHostToDevice));
    m->partition.table_host.assign(p->owners, p->owners + cells);
    m->mc.owner_table = static_cast<const unsigned char *>(m->partition.table_dev.ptr);
    for (int a = 0; a < 3; ++a)
    {
      m->mc.owner_grid_origin[a] = p->grid_origin[a];
      m->mc.owner_grid_dims[a] = int(p->grid_dims[a]);
    }
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_partition_owners(const ohmhip_partition *p, const int16_t *keys_xyz, size_t count, uint32_t *owners)
try
{
  if (!p || (count && (!keys_xyz || !owners)) || p->block_shift < 0 || p->block_shift > 15)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const size_t cells = size_t(p->grid_dims[0]) * size_t(p->grid_dims[1]) * size_t(p->grid_dims[2]);
  if ((cells != 0) != (p->owners != nullptr))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const int dims[3] = { int(p->grid_dims[0]), int(p->grid_dims[1]), int(p->grid_dims[2]) };
  for (size_t i = 0; i < count; ++i)
  {
    owners[i] = (p->world_size > 1) ? partitionOwner(p->owners, p->grid_origin, dims, p->block_shift, p->world_size,
                                                      keys_xyz[3 * i], keys_xyz[3 * i + 1], keys_xyz[3 * i + 2])
                                    : 0u;
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_region_owners(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, uint32_t *owners)
try
{
  if (!m || (count && (!keys_xyz || !owners)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  for (size_t i = 0; i < count; ++i)
  {
    owners[i] = hostRegionOwner(m, keys_xyz + 3 * i);
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_map_route_rays(ohmhip_map_t m, const double *d_rays, size_t ray_count, unsigned ray_flags, double *d_routed,
                          uint32_t *d_routed_index, size_t capacity, uint32_t *counts, uint64_t *visits)
try
{
  if (!m || !counts || (ray_count && !d_rays) || ray_count > 0xfffffff0ull || (capacity && !d_routed))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const uint32_t world = std::max(1u, m->mc.owner_world);
  if (world > kRouteMaxWorld)
  {
    return OHMHIP_ERR_UNSUPPORTED;
  }
  if (visits)
  {
    *visits = 0;
  }
  for (uint32_t d = 0; d < world; ++d)
  {
    counts[d] = 0;
  }
  if (ray_count == 0)
  {
    return OHMHIP_OK;
  }
  OHMHIP_CHECK(hipSetDevice(m->device));
  ohmhip_map_s::PartitionState &ps = m->partition;
  if (!ps.h_totals)
  {
    OHMHIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ps.h_totals), sizeof(uint32_t) * kRouteMaxWorld + sizeof(uint64_t),
                               hipHostMallocMapped));
    OHMHIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&ps.h_totals_dev), ps.h_totals, 0));
  }
  // A stream of its own: routing reads the rays, the map's constants and the territory table -- nothing a batch writes --
  // so the routing of the next batch runs beside the batches still in flight (the call itself stays synchronous).
  if (!ps.route_stream)
  {
    OHMHIP_CHECK(hipStreamCreateWithFlags(&ps.route_stream, hipStreamNonBlocking));
  }
  hipStream_t s = ps.route_stream;
  const uint32_t n = uint32_t(ray_count);
  const uint32_t blocks = (n + kRouteThreads - 1) / kRouteThreads;
  OHMHIP_CHECK(ps.masks.ensure(sizeof(unsigned long long) * size_t(n), false, s));
  OHMHIP_CHECK(ps.block_counts.ensure(sizeof(uint32_t) * size_t(blocks) * world, false, s));
  OHMHIP_CHECK(ps.totals.ensure(sizeof(uint32_t) * kRouteMaxWorld + sizeof(unsigned long long), false, s));
  // (the visit counter lives behind the totals)
  unsigned long long *d_visits = reinterpret_cast<unsigned long long *>(static_cast<uint32_t *>(ps.totals.ptr) + kRouteMaxWorld);
  OHMHIP_CHECK(hipMemsetAsync(d_visits, 0, sizeof(unsigned long long), s));
  // A batch the caller filtered (ohmhip_map_integrate_rays_filtered) carries per-ray flags that do not travel with routed
  // rays: the routing applies the map's own ray filter, like the integration of the routed rays will.
  // (The map's launch thread -- ohmhip_map_set_async_launch -- writes per-batch fields of m->mc while it launches: the
  // copy is taken once that thread is idle; its status stays with the map for the next settling call.  ADVICE r4.)
  if (m->launch_busy)
  {
    m->launch_thread->wait();
  }
  MapConst mc = m->mc;
  mc.batch_filter_flags = nullptr;
  hipLaunchKernelGGL(k_route_mask, dim3(blocks), dim3(kRouteThreads), 0, s, mc, d_rays, n, ray_flags,
                     static_cast<unsigned long long *>(ps.masks.ptr), static_cast<uint32_t *>(ps.block_counts.ptr), world,
                     d_visits);
  hipLaunchKernelGGL(k_route_scan, dim3(world), dim3(1024), 0, s, static_cast<uint32_t *>(ps.block_counts.ptr), blocks,
                     static_cast<uint32_t *>(ps.totals.ptr), ps.h_totals_dev);
  hipLaunchKernelGGL(k_route_scatter, dim3(blocks), dim3(kRouteThreads), 0, s, d_rays, n,
                     static_cast<const unsigned long long *>(ps.masks.ptr), static_cast<const uint32_t *>(ps.block_counts.ptr),
                     static_cast<const uint32_t *>(ps.totals.ptr), world, d_routed, d_routed_index,
                     uint32_t(std::min<size_t>(capacity, 0xffffffffu)));
  unsigned long long h_visits = 0;
  OHMHIP_CHECK(hipMemcpyAsync(&h_visits, d_visits, sizeof(h_visits), hipMemcpyDeviceToHost, s));
  OHMHIP_CHECK(hipStreamSynchronize(s));
  OHMHIP_CHECK(hipGetLastError());
  uint64_t total = 0;
  for (uint32_t d = 0; d < world; ++d)
  {
    counts[d] = ps.h_totals[d];
    total += counts[d];
  }
  if (visits)
  {
    *visits = h_visits;
  }
  return (total > capacity) ? OHMHIP_ERR_CAPACITY : OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_comm_exchange_counts(ohmhip_comm_t comm, const uint32_t *send_counts, uint32_t *recv_counts,
                                ohmhip_stream_t stream)
try
{
  if (!comm || !send_counts || !recv_counts)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const int world = comm->world;
  hipStream_t s = stream ? stream->stream : nullptr;
  uint32_t *row = comm->d_counts;
  uint32_t *matrix = comm->d_counts + world;
  OHMHIP_CHECK(hipMemcpyAsync(row, send_counts, sizeof(uint32_t) * world, hipMemcpyHostToDevice, s));
  OHMHIP_CHECK(ncclStatus(ncclAllGather(row, matrix, size_t(world), ncclUint32, comm->comm, s)));
  std::vector<uint32_t> host(size_t(world) * world);
  OHMHIP_CHECK(hipMemcpyAsync(host.data(), matrix, sizeof(uint32_t) * host.size(), hipMemcpyDeviceToHost, s));
  OHMHIP_CHECK(hipStreamSynchronize(s));
  // A rank whose local step failed (routing error, no memory ...) reports OHMHIP_COUNT_FAILED for every destination
  // instead of counts: every rank sees it here, in the same call, and none goes on to the payload exchange (ADVICE r4:
  // the peers of a failing rank used to block in the next collective).
  for (size_t i = 0; i < host.size(); ++i)
  {
    if (host[i] == OHMHIP_COUNT_FAILED)
    {
      for (int src = 0; src < world; ++src)
      {
        recv_counts[src] = 0;
      }
      return OHMHIP_ERR_PEER;
    }
  }
  for (int src = 0; src < world; ++src)
  {
    recv_counts[src] = host[size_t(src) * world + comm->rank];  // what rank `src` addressed to this rank
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

}  // extern "C"

namespace
{
/// dst row i = src row index[i] (rows of `row_bytes` bytes, 4-byte granularity when the size allows).
__global__ void __launch_bounds__(256)
  k_gather_rows(const unsigned char *__restrict__ src, const uint32_t *__restrict__ index, uint32_t count,
                uint32_t row_bytes, unsigned char *__restrict__ dst)
{
  const uint32_t words = (row_bytes % 4u == 0u) ? row_bytes / 4u : 0u;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  if (words)
  {
    const size_t total = size_t(count) * words;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride)
    {
      const uint32_t row = uint32_t(i / words), w = uint32_t(i % words);
      reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[size_t(index[row]) * words + w];
    }
    return;
  }
  const size_t total = size_t(count) * row_bytes;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride)
  {
    const uint32_t row = uint32_t(i / row_bytes), b = uint32_t(i % row_bytes);
    dst[i] = src[size_t(index[row]) * row_bytes + b];
  }
}

/// The all-to-all of one per-ray array (rays: 48 bytes per ray; time stamps: 8; intensities: 4) over ncclSend / ncclRecv
/// in one group.  Everything that can be wrong with the arguments is found BEFORE the group starts: a rank
/// that has entered the collective goes through with it.
int exchangeBytes(ohmhip_comm_t comm, const unsigned char *d_send, const uint32_t *send_counts, unsigned char *d_recv,
                  const uint32_t *recv_counts, size_t bytes_per_ray, hipStream_t s)
{
  if (!comm || !send_counts || !recv_counts || bytes_per_ray == 0)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const int world = comm->world;
  size_t send_total = 0, recv_total = 0;
  for (int peer = 0; peer < world; ++peer)
  {
    send_total += send_counts[peer];
    recv_total += recv_counts[peer];
  }
  if ((send_total && !d_send) || (recv_total && !d_recv) || send_counts[comm->rank] != recv_counts[comm->rank])
  {
    return OHMHIP_ERR_INVALID_ARG;  // (the block a rank addresses to itself never leaves the device: same size both ways)
  }
  size_t send_at = 0, recv_at = 0, self_send = 0, self_recv = 0;
  OHMHIP_CHECK(ncclStatus(ncclGroupStart()));
  int err = OHMHIP_OK;
  for (int peer = 0; peer < world && err == OHMHIP_OK; ++peer)
  {
    if (peer == comm->rank)
    {
      self_send = send_at;
      self_recv = recv_at;
    }
    else
    {
      if (send_counts[peer])
      {
        err = ncclStatus(ncclSend(d_send + send_at * bytes_per_ray, size_t(send_counts[peer]) * bytes_per_ray, ncclUint8,
                                  peer, comm->comm, s));
      }
      if (err == OHMHIP_OK && recv_counts[peer])
      {
        err = ncclStatus(ncclRecv(d_recv + recv_at * bytes_per_ray, size_t(recv_counts[peer]) * bytes_per_ray, ncclUint8,
                                  peer, comm->comm, s));
      }
    }
    send_at += send_counts[peer];
    recv_at += recv_counts[peer];
  }
  const int end_err = ncclStatus(ncclGroupEnd());
  OHMHIP_CHECK(err);
  OHMHIP_CHECK(end_err);
  if (send_counts[comm->rank])
  {
    OHMHIP_CHECK(hipMemcpyAsync(d_recv + self_recv * bytes_per_ray, d_send + self_send * bytes_per_ray,
                                bytes_per_ray * size_t(send_counts[comm->rank]), hipMemcpyDeviceToDevice, s));
  }
  return OHMHIP_OK;
}
}  // namespace

extern "C" {

int ohmhip_comm_exchange_rays(ohmhip_comm_t comm, const double *d_send, const uint32_t *send_counts, double *d_recv,
                              const uint32_t *recv_counts, ohmhip_stream_t stream)
try
{
  return exchangeBytes(comm, reinterpret_cast<const unsigned char *>(d_send), send_counts,
                       reinterpret_cast<unsigned char *>(d_recv), recv_counts, 6 * sizeof(double),
                       stream ? stream->stream : nullptr);
}
OHMHIP_ABI_CATCH

int ohmhip_comm_exchange_side(ohmhip_comm_t comm, const void *d_send, const uint32_t *send_counts, void *d_recv,
                              const uint32_t *recv_counts, uint32_t bytes_per_ray, ohmhip_stream_t stream)
try
{
  return exchangeBytes(comm, static_cast<const unsigned char *>(d_send), send_counts, static_cast<unsigned char *>(d_recv),
                       recv_counts, bytes_per_ray, stream ? stream->stream : nullptr);
}
OHMHIP_ABI_CATCH

int ohmhip_gather_rows(const void *d_src, const uint32_t *d_index, size_t count, uint32_t bytes_per_row, void *d_dst,
                       ohmhip_stream_t stream)
try
{
  if (count == 0)
  {
    return OHMHIP_OK;
  }
  if (!d_src || !d_index || !d_dst || bytes_per_row == 0 || count > 0xffffffffull)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  const size_t units = count * size_t((bytes_per_row % 4u == 0u) ? bytes_per_row / 4u : bytes_per_row);
  const uint32_t blocks = uint32_t(std::min<size_t>((units + 255) / 256, 65535));
  hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, stream ? stream->stream : nullptr,
                     static_cast<const unsigned char *>(d_src), d_index, uint32_t(count), bytes_per_row,
                     static_cast<unsigned char *>(d_dst));
  return int(hipGetLastError());
}
OHMHIP_ABI_CATCH

}  // extern "C"

#endif  // OHMHIP_PARTITION_IMPL_H
