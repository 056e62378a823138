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

namespace
{
const size_t kLayerBytes[OHMHIP_LID_COUNT] = { 4, 8, 24, 4, 4, 4, 8, 8, 8 };

struct DevBuf
{
  void *ptr = nullptr;
  size_t bytes = 0;

  int ensure(size_t want, bool zero, hipStream_t stream)
  {
    if (want <= bytes)
    {
      return OHMHIP_OK;
    }
    if (ptr)
    {
      OHMHIP_CHECK(hipStreamSynchronize(stream));
      OHMHIP_CHECK(hipFree(ptr));
      ptr = nullptr;
      bytes = 0;
    }
    // Grow geometrically so steady-state batches never reallocate.
    size_t alloc = std::max(want, size_t(1) << 16);
    alloc = (alloc + (alloc >> 2) + 255) & ~size_t(255);
    OHMHIP_CHECK(hipMalloc(&ptr, alloc));
    bytes = alloc;
    if (zero)
    {
      OHMHIP_CHECK(hipMemsetAsync(ptr, 0, alloc, stream));
    }
    return OHMHIP_OK;
  }

  void release()
  {
    if (ptr)
    {
      (void)hipFree(ptr);
    }
    ptr = nullptr;
    bytes = 0;
  }
};
}  // namespace

constexpr uint32_t kTimingRing = 32;
constexpr uint32_t kDirtySync = 1u;   ///< d_dirty bit: modified since the last syncVoxels() (ohmhip_map_clear_dirty)
constexpr uint32_t kDirtyMerge = 2u;  ///< d_dirty bit: modified since the last replica merge (merge_impl.h)

/// A few host threads that stay around for the life of a map: staging a large host ray block into pinned memory is a
/// memcpy one core cannot do at PCIe speed, and starting threads per call costs as much as the copy of a small batch.
class StagePool
{
public:
  explicit StagePool(unsigned n_threads)
  {
    for (unsigned i = 0; i < n_threads; ++i)
    {
      threads_.emplace_back([this, i] { loop(i); });
    }
  }
  ~StagePool()
  {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto &t : threads_)
    {
      t.join();
    }
  }
  unsigned size() const { return unsigned(threads_.size()); }
  /// Start job(worker index) on the first `n_workers` threads; returns at once.
  void start(unsigned n_workers, std::function<void(unsigned)> job)
  {
    std::lock_guard<std::mutex> lock(mu_);
    job_ = std::move(job);
    active_ = std::min<unsigned>(n_workers, size());
    running_ = active_;
    ++generation_;
    cv_work_.notify_all();
  }
  /// Block until every worker of the last start() has returned.
  void wait()
  {
    std::unique_lock<std::mutex> lock(mu_);
    cv_done_.wait(lock, [this] { return running_ == 0; });
  }

private:
  void loop(unsigned index)
  {
    uint64_t seen = 0;
    for (;;)
    {
      std::function<void(unsigned)> job;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_work_.wait(lock, [&] { return stop_ || generation_ != seen; });
        if (stop_)
        {
          return;
        }
        seen = generation_;
        if (index >= active_)
        {
          continue;
        }
        job = job_;
      }
      job(index);
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (--running_ == 0)
        {
          cv_done_.notify_all();
        }
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::function<void(unsigned)> job_;
  uint64_t generation_ = 0;
  unsigned active_ = 0, running_ = 0;
  bool stop_ = false;
};

struct ohmhip_map_s
{
  ohmhip_map_config config;
  MapConst mc;
  int device = 0;
  hipStream_t stream = nullptr;       ///< compute stream
  hipStream_t copy_stream = nullptr;  ///< side stream for region upload/download
  /// Stream of a batch's set-up pass (k_ray_setup, k_plan).  It reads the rays and the region table only, and writes
  /// per-batch scratch that exists twice (see `parity`), so the set-up of batch N+1 runs beside the sample sort of batch
  /// N and in the CUs its walk kernel vacates.  It is idle whenever no batch call is in progress: every call waits for
  /// its own plan summary.
  hipStream_t front_stream = nullptr;
  hipEvent_t ev_batch_done[2] = { nullptr, nullptr };  ///< per parity: the batch that last used this scratch copy is done
  bool batch_done_recorded[2] = { false, false };
  hipEvent_t ev_bin_done = nullptr;  ///< the latest k_ray_bin has finished
  bool bin_done_recorded = false;
  uint32_t parity = 0;  ///< which copy of the doubled per-batch scratch (RayWalk array, per-hash / per-slot counters,
                        ///< chunk list, event counters) the current batch uses
  hipEvent_t ev[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  /// Timing events of the last kTimingRing batches (start, binned, samples ordered, walked, done): reading a batch's
  /// phase times does not have to synchronise the host with every batch.
  hipEvent_t tev[kTimingRing][7] = {};  // ([5]: set-up pass done, [6]: binning starts)
  uint64_t batch_seq = 0;

  uint32_t slot_capacity = 0;
  uint32_t hash_capacity = 0;
  uint32_t slots_committed = 0;  ///< slots in use after the last successful batch / upload

  void *layers[OHMHIP_LID_COUNT] = {};
  unsigned long long *d_keys = nullptr;
  uint32_t *d_vals = nullptr;
  uint64_t *d_slot_keys = nullptr;
  uint32_t *d_n_slots = nullptr;
  // scratch
  uint32_t *d_hit_count = nullptr, *d_sort_list = nullptr;
  uint32_t *d_seg_count = nullptr, *d_seg_cursor = nullptr, *d_seg_offset = nullptr, *d_touched_flag = nullptr,
           *d_touched = nullptr;
  uint32_t *d_voxel_first_hit = nullptr, *d_hit_begin = nullptr, *d_hit_end = nullptr, *d_dirty = nullptr;
  /// [2 x slot_capacity] per slot: the stamp of the batch that used the region last, and the stamp of the last use before
  /// the current run of consecutive batches (0: none) -- what the spill policy predicts a region's next use from
  /// (touchRegionUse, evictColdRegions); moves with the slot
  uint32_t *d_last_use = nullptr;
  BatchInfo *d_info = nullptr;   ///< three summaries used in turn: k_plan of one batch zeroes the next batch's
  BatchInfo *h_info = nullptr;   ///< pinned, device visible: [0] batch summary (written by k_plan), [1] event count
  BatchInfo *h_info_dev = nullptr;  ///< device address of h_info
  uint32_t info_index = 0;
  bool info_clean = false;       ///< d_info[next index] was zeroed by the previous batch's k_plan
  uint32_t *d_miss_counts = nullptr;
  uint32_t *d_hit_mask = nullptr;
  Chunk *d_chunks = nullptr;
  uint32_t chunk_capacity = 0;

  DevBuf walks_buf[2], hit_keys_a, hit_keys_b, interval_counts, segments, sort_temp, events;
  DevBuf wg_regions[2], wg_region_count[2], group_heads;  // (workgroup region lists: per parity)
  /// Replica merge (merge_impl.h): base copy of the occupancy layer (null until ohmhip_map_enable_merge) and scratch.
  float *d_merge_base = nullptr;
  int merge_mode = 0;  ///< OHMHIP_MERGE_SHARED_ONLY / OHMHIP_MERGE_FULL_UNION
  /// Traversal layer only: per-voxel fixed-point sum of a batch's ray lengths (zero between batches).
  unsigned long long *d_traversal_acc = nullptr;
  DevBuf merge_slots, merge_keys_dev, merge_delta, merge_observers;
  /// Regions cut into tiles (tiling_impl.h): > 0 while the translation layer calls back into the entry points with tile
  /// keys.
  int tile_passthrough = 0;
  /// Partitioned map (partition_impl.h): the owner table of ohmhip_map_set_region_partition (host copy for
  /// ohmhip_map_region_owners, device copy behind MapConst::owner_table) and the scratch of ohmhip_map_route_rays.
  struct PartitionState
  {
    std::vector<unsigned char> table_host;
    DevBuf table_dev, masks, block_counts, totals;
    uint32_t *h_totals = nullptr;      ///< pinned, device visible: rays per destination of the last routing
    uint32_t *h_totals_dev = nullptr;
  } partition;
  DevBuf use_scratch;  ///< (slot, stamp) pairs of re-admitted regions (queueReadmission)
  /// After how many batches the regions re-admitted lately came back (ring of the last 256): their median stands in as
  /// the period of regions that have no history of their own yet (evictColdRegions).
  std::vector<uint32_t> readmit_periods;
  size_t readmit_period_at = 0;
  DevBuf copy_jobs;  ///< job list of k_copy_jobs (spill to host, compaction)
  DevBuf stop_a, stop_b;  ///< kRfStopOnFirstOccupied: per-ray stop positions (current / candidate)
  uint32_t *d_event_count = nullptr;  ///< per parity: [0] deferred event count, [1] walk kernel chunk cursor, [2] replay group count, [3] stop iteration flag
  uint32_t walk_workgroups = 256;     ///< persistent walk workgroups: one per CU
  unsigned long long *d_dbg = nullptr;  ///< 8 debug counters (OHMHIP_DEBUG_FLAGS & 64)
  double first_ray_time = -1.0;  ///< OccupancyMap::firstRayTime() (ohm/OccupancyMap.cpp:343-347)
  uint32_t event_demand = 0;
  uint32_t event_limit = 0;  ///< OHMHIP_EVENT_LIMIT (tests): cap of the NDT / TSDF event list's first sizing
  bool spec_bucket_ok = false;  ///< the previous occupancy batch used the per-region sample sort: bin speculatively
  double segments_per_ray = 10.0;            ///< running estimate (previous batch) used to size the next batch's chunks
  uint32_t bin_rays_per_block = kBinRaysPerBlock;  ///< tunable (OHMHIP_BIN_RAYS): rays per binning workgroup, large batches
  uint32_t min_chunk_segments = 2048;  ///< tunable (OHMHIP_MIN_CHUNK_SEGMENTS): floor of the small-batch chunk size (two rounds of the walk workgroup's 1024 lanes)
  uint32_t chunk_segments = kChunkSegments;  ///< tunable (OHMHIP_CHUNK_SEGMENTS), <= kMaxChunkSegments (15-bit LDS counters)
  /// OHMHIP_DEBUG_FLAGS (development only): 16 = walk kernel refills lanes but does not walk (timing experiments,
  /// breaks results); 64 = per-chunk timing trace of the walk kernel (OHMHIP_DEBUG_TRACE=<file>, scripts/
  /// analyse_trace.py); 128 = iteration / visit / refill counters (hot-address atomics: distorts timing); 256 = phase
  /// timeline of the last three batches printed by ohmhip_map_sync; 512 = spill path timers; 2048 = where a host batch's
  /// call spends its time; 4096 = one line per batch: segments, chunks, regions, densest region.
  unsigned debug_flags = 0;
  int refill_min_idle = kRefillMinIdle;      ///< tunable (OHMHIP_REFILL_MIN_IDLE)  ///< events the previous batch produced (sizes the next batch's list)
  void *h_stage = nullptr;  ///< pinned staging for region copies
  size_t h_stage_bytes = 0;

  /// Host-pointer ray batches go through one of two staging slots (pinned host block + device copies), so the host
  /// copy and the H2D transfer of batch N+1 overlap the device work of batch N.  With coalescing on, consecutive small
  /// batches with the same flags accumulate in the filling slot and run as one device batch.
  struct RaySlot
  {
    char *h = nullptr;            ///< pinned: capacity x 48 B rays, x 8 B timestamps, x 4 B intensities, x 1 B filter flags
    size_t capacity = 0;          ///< rays
    DevBuf d_rays, d_times, d_intens, d_fflags;
    hipEvent_t uploaded = nullptr;  ///< H2D copies done (copy stream)
    hipEvent_t done = nullptr;      ///< the batch reading the device copies has finished (compute stream)
    bool in_flight = false;
    bool rays_uploaded = false;     ///< the rays' H2D copies were queued piece by piece while the block was staged
  } ray_slots[2];
  std::unique_ptr<StagePool> stage_pool;  ///< created by the first large host batch
  /// ohmhip_map_set_async_launch: a host batch's device launch sequence (with its host round trip for the plan) runs on
  /// this one thread while the caller returns and stages its next block.
  bool async_launch = false;
  std::unique_ptr<StagePool> launch_thread;
  bool launch_busy = false;
  int launch_result = OHMHIP_OK;
  int fill_slot = 0;
  size_t pending_rays = 0;
  size_t pending_calls = 0;
  unsigned pending_flags = 0;
  bool pending_intens = false, pending_times = false, pending_fflags = false;
  /// The pending rays were presented through the device-pointer entry point: they sit in the filling slot's DEVICE
  /// buffers already (copied there device to device), the pinned block is not used.
  bool pending_on_device = false;
  uint32_t *h_passed = nullptr;      ///< pinned, device visible: per-call filter count of a deferred device-pointer batch
  uint32_t *h_passed_dev = nullptr;
  hipEvent_t ev_passed = nullptr;
  /// Host-pointer batches smaller than this are collected and run as one device batch (0: every host batch is launched
  /// by the call that presents it).  On by default: the reference tools present 4096 rays per call.
  size_t coalesce_min_rays = size_t(1) << 16;

  // host mirror of the region table
  std::unordered_map<uint64_t, uint32_t> region_slots;
  std::vector<uint64_t> slot_keys_host;

  ohmhip_batch_stats stats = {};
  bool stats_pending = false;
  uint64_t cache_hits = 0, cache_misses = 0, cache_full = 0;  ///< ohmhip_map_cache_stats
  uint64_t memory_limit = 0;                                   ///< ohmhip_map_set_memory_limit
  /// Spill to host (ohmhip_map_set_spill_to_host): regions evicted from the pool when the memory limit is reached, by
  /// packed key.  A spilled region is still part of the map: it is listed, read and synced from here, and moves back
  /// into the pool when a batch (or an upload) touches it.
  struct SpilledRegion
  {
    /// One record of the pinned host store: the region's block of every enabled layer, in layer-id order, followed by
    /// its row of the NDT / TSDF replay mask (layerOffset / maskOffset below).  Pinned, so evictions and re-admissions
    /// are single asynchronous copies straight between the pool and the record -- no staging pass on either side.
    char *record = nullptr;
    uint32_t dirty = 0;
    uint32_t last_use = 0;  ///< stamp of the last batch that used the region before it left the pool
  };
  /// Pinned host store: slabs of fixed-size records, handed out from a free list.
  struct HostStore
  {
    size_t record_bytes = 0;
    size_t layer_offset[OHMHIP_LID_COUNT] = {};
    size_t mask_offset = 0;
    size_t mask_bytes = 0;
    std::vector<void *> slabs;
    std::vector<char *> free_records;
    size_t records_total = 0;
  } store;
  std::unordered_map<uint64_t, SpilledRegion> spilled;
  /// Background write-back (writeback_impl.h): resident regions whose content already sits in a store record, valid
  /// while the region's use stamp is the one the copy was taken at.
  struct Precleaned
  {
    char *record = nullptr;
    uint32_t last_use = 0;
  };
  std::unordered_map<uint64_t, Precleaned> precleaned;
  std::vector<char *> stale_records;  ///< records of discarded copies, recycled once the copy stream has passed them
  static constexpr uint32_t kWritebackRing = 4;
  struct WritebackRing
  {
    DevBuf jobs;
    hipEvent_t done = nullptr;
    bool used = false;
  } wb_ring[kWritebackRing];
  uint32_t wb_next = 0;
  uint32_t *h_use = nullptr;     ///< pinned: the resident regions' use stamps as of the latest plan (queueUseStamps)
  size_t h_use_capacity = 0;
  uint32_t h_use_slots = 0;      ///< slots the copy covers
  uint32_t evicted_per_call = 0; ///< regions the latest eviction moved out (sizes the write-back's lead)
  bool writeback_off = true;     ///< ohmhip_map_set_spill_writeback (off by default; OHMHIP_WRITEBACK=0 / 1 overrides)
  uint64_t writebacks = 0, writeback_hits = 0, writeback_stale = 0;
  bool spill_enabled = false;
  uint64_t evictions = 0, readmissions = 0;
  double spill_ms[6] = { 0, 0, 0, 0, 0, 0 };  ///< OHMHIP_DEBUG_FLAGS & 512: evict select / copy / compact, readmit copy, failed attempts, store growth
};

// Background write-back of the spill path (writeback_impl.h).
namespace
{
int queueUseStamps(ohmhip_map_t m, hipStream_t stream);
void scheduleWriteBack(ohmhip_map_t m, uint32_t now);
void dropPrecleaned(ohmhip_map_t m);
void dropPrecleanedKey(ohmhip_map_t m, uint64_t key);
}  // namespace

// Regions larger than one tile (tiling_impl.h): the entry points that name or list regions translate.
inline bool tiledBoundary(ohmhip_map_t m)
{
  return m && (m->mc.tile_split[1] > 1 || m->mc.tile_split[2] > 1) && m->tile_passthrough == 0;
}
namespace
{
void chooseTileDims(const int dims[3], int limit, int tile[3]);
int tiledListRegions(ohmhip_map_t m, bool dirty_only, int16_t *keys_xyz, size_t capacity, size_t *count);
int tiledReadRegions(ohmhip_map_t m, int layer_id, const int16_t *keys_xyz, size_t count, void *const *dsts);
int tiledWriteRegions(ohmhip_map_t m, int layer_id, const int16_t *keys_xyz, size_t count, const void *const *srcs);
int tiledRemoveRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, size_t *removed);
}  // namespace

// Defined further down (they use the region read / remove machinery of the C ABI section).
int removeResidentRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, size_t *removed);
int evictColdRegions(ohmhip_map_t m, uint32_t want_free, uint32_t max_evict = 0xffffffffu);
int makeRoomForNamedRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count);
int growPoolForNamedRegions(ohmhip_map_t m, uint32_t total, uint32_t keep);
int readmitSpilledSlots(ohmhip_map_t m, uint32_t first_slot, uint32_t end_slot);
int readmitSpilledKeys(ohmhip_map_t m, const int16_t *keys_xyz, size_t count);

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
  bs.sort_list = m->d_sort_list + h;
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
  if (m->copy_stream && (!m->precleaned.empty() || !m->stale_records.empty()))
  {
    OHMHIP_CHECK(hipStreamSynchronize(m->copy_stream));  // background write-back copies read the pool being replaced
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
    OHMHIP_CHECK(zalloc(reinterpret_cast<void **>(&n_sort_list), sizeof(uint32_t) * 2 * hash_cap));
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
                           m->d_touched_flag, m->d_touched,    m->d_sort_list };
  for (uint32_t *p : per_hash)
  {
    OHMHIP_CHECK(hipMemsetAsync(p, 0, sizeof(uint32_t) * 2 * hash_words, s));  // (both parities)
  }
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

size_t walkLdsBytes(const MapConst &mc, uint32_t chunk_segments)
{
  // [count tile, padded to 16 B][per-wave queues][staged sample keys][interval counters][cursor + pad]
  // [length histogram][segment order, u16 each]
  const size_t count_words = (size_t((mc.region_voxels + 1) / 2) + 31u) & ~size_t(31);  // whole 32-word rows (tileWord)
  return (count_words + size_t(2 * kWalkWaves * kQueueCap) + size_t(2 * kLdsHits) + size_t(kLdsHits / 2) +
          kWalkCursorWords + 64 + (kIndexBuckets + 2) / 2 +
          kLengthClasses + (chunk_segments + 1) / 2) *
         sizeof(uint32_t);
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
  // decided once per call
  uint32_t next_info_index = 0;
  bool info_clean = false;
  uint32_t ray_blocks = 0, bin_rays_per_block = 0, bin_threads = 0, bin_blocks = 0, bin_tab_mask = 0;
  uint32_t batch_chunk_segments = 0;
  int mode = 0;
  bool stop_mode = false, occupancy_mode = false, ndt_mode = false, tsdf_mode = false;
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

  int prepare()
  {
    // This batch's summary block: the next of the three, zeroed by the previous batch's k_plan if that ran.  (Three: the
    // set-up pass of this batch runs under the previous batch's apply kernels, which still read theirs, and zeroes the
    // following batch's.)
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
    while (batch_chunk_segments > m->min_chunk_segments &&
           expected_segments / batch_chunk_segments < 3ull * m->walk_workgroups)
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
    if (bin_tab_mask < kLtabSmall)
    {
      hipLaunchKernelGGL(k_ray_bin<kLtabSmall>, dim3(bin_blocks), dim3(bin_threads), 0, s, m->mc, regionTable(m),
                         batchScratch(m), static_cast<const RayWalk *>(batchWalks(m).ptr), n_rays,
                         static_cast<Segment *>(m->segments.ptr), seg_capacity, hit_keys, m->d_hit_mask, ray_shift,
                         bucket ? 1 : 0, bin_rays_per_block, bin_tab_mask);
    }
    else
    {
      hipLaunchKernelGGL(k_ray_bin<kLtabSize>, dim3(bin_blocks), dim3(bin_threads), 0, s, m->mc, regionTable(m),
                         batchScratch(m), static_cast<const RayWalk *>(batchWalks(m).ptr), n_rays,
                         static_cast<Segment *>(m->segments.ptr), seg_capacity, hit_keys, m->d_hit_mask, ray_shift,
                         bucket ? 1 : 0, bin_rays_per_block, bin_tab_mask);
    }
    (void)hipEventRecord(m->ev_bin_done, s);
    m->bin_done_recorded = true;
  }

  void launchRegionSort()
  {
    hipLaunchKernelGGL(k_sort_region_hits, dim3(4 * m->walk_workgroups), dim3(kSortThreads), 0, s, regionTable(m),
                       batchScratch(m), static_cast<const unsigned long long *>(m->hit_keys_a.ptr),
                       static_cast<unsigned long long *>(m->hit_keys_b.ptr), m->mc.region_voxels);
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
    if (m->batch_done_recorded[m->parity])
    {
      OHMHIP_CHECK(hipStreamWaitEvent(f, m->ev_batch_done[m->parity], 0));
    }
    if (m->bin_done_recorded)
    {
      OHMHIP_CHECK(hipStreamWaitEvent(f, m->ev_bin_done, 0));
    }
    if (attempt > 0 || !info_clean)
    {
      OHMHIP_CHECK(hipMemsetAsync(m->d_info + m->info_index, 0, sizeof(BatchInfo), f));
    }
    OHMHIP_CHECK(hipEventRecord(tev[0], f));
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
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, f, regionTable(m), batchScratch(m), batchChunks(m),
                       m->chunk_capacity, batch_chunk_segments, m->h_info_dev, m->d_info + next_info_index,
                       batchEventCount(m));
    m->info_clean = true;
    OHMHIP_CHECK(queueUseStamps(m, f));  // (spill to host: the regions' use stamps reach the host with the summary)
    OHMHIP_CHECK(hipEventRecord(tev[5], f));
    OHMHIP_CHECK(hipEventRecord(m->ev[7], f));
    OHMHIP_CHECK(hipStreamWaitEvent(s, m->ev[7], 0));
    OHMHIP_CHECK(hipEventRecord(tev[6], s));
    // The host needs the batch summary (segment count, sample distribution, pool state) before it can size and launch
    // the rest -- a round trip during which the device would idle.  In steady state (occupancy, previous batch sorted
    // its samples per region) the binning and the sample sort are launched right away with the buffers of the previous
    // batch; the summary then only confirms the guess, and a wrong guess costs a repeat of the two passes.
    spec_seg_cap = uint32_t(std::min<size_t>(m->segments.bytes / sizeof(Segment), 0xffffffffu));
    speculated = occupancy_mode && m->spec_bucket_ok && attempt == 0 && spec_seg_cap > 0;
    if (speculated)
    {
      launchBin(true, spec_seg_cap, static_cast<unsigned long long *>(m->hit_keys_a.ptr));
      OHMHIP_CHECK(hipEventRecord(tev[1], s));
      launchRegionSort();
      OHMHIP_CHECK(hipEventRecord(tev[2], s));
    }
    OHMHIP_CHECK(hipEventSynchronize(m->ev[7]));
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
    if (m->debug_flags & 4096u)
    {
      std::fprintf(stderr, "[ohmhip dbg] batch: %u rays, %u segments, %u chunks, %u regions touched, %u with samples, "
                   "densest %u samples; binned speculatively: %d\n", n_rays, info.n_segments, info.n_chunks,
                   info.n_touched, info.n_hit_regions, info.max_region_hits, int(speculated));
    }
    sorted = keys_b;
    if (!speculated)
    {
      launchBin(bucket_hits, seg_cap, keys_a);
      if (tsdf_mode)
      {
        hipLaunchKernelGGL(k_tsdf_flag, dim3(ray_blocks), dim3(256), 0, s, m->mc, regionTable(m),
                           static_cast<const RayWalk *>(batchWalks(m).ptr), d_rays, n_rays, m->d_hit_mask);
      }
      OHMHIP_CHECK(hipEventRecord(tev[1], s));
      if (bucket_hits)
      {
        if (info.n_hit_regions)
        {
          launchRegionSort();
        }
      }
      else if (occupancy_mode)
      {
        size_t temp_bytes = m->sort_temp.bytes;
        // Sample keys are emitted in ray order and the radix sort is stable: sorting on the (slot, voxel) bits alone
        // leaves each voxel's samples in ray order.
        OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(m->sort_temp.ptr, temp_bytes, keys_a, keys_b, size_t(n_rays),
                                                          kHitRayBits, sortEndBit(info.n_slots), s));
        hipLaunchKernelGGL(k_hit_bounds, dim3(ray_blocks), dim3(256), 0, s, sorted, batchScratch(m),
                           m->mc.region_voxels);
      }
      OHMHIP_CHECK(hipEventRecord(tev[2], s));
    }
    return OHMHIP_OK;
  }

  int walk()
  {
    // Single-chunk regions are applied by the walk kernel straight from LDS (plain log-odds misses only).
    direct_occ = (occupancy_mode || mode == OHMHIP_MODE_NDT_OM) ? static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]) :
                                                                  nullptr;
    direct_segments = (direct_occ || tsdf_mode) ? batch_chunk_segments : 0u;
    n_events = 0;
    if (info.n_chunks)
    {
      for (int walk_attempt = 0; walk_attempt < 4; ++walk_attempt)
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
        wa.inline_hits = (occupancy_mode && !m->layers[OHMHIP_LID_MEAN] && !sec.traversal && !sec.touch_time &&
                          !sec.incident && !m->layers[OHMHIP_LID_INTENSITY] && !m->layers[OHMHIP_LID_HIT_MISS]) ?
                           1 :
                           0;
        // Traversal layer: its own fp64 pass over the chunk list after the count walk (traversal_kernels.h).
        const bool traversal_pass = sec.traversal != nullptr && walk_attempt == 0;
        // The lean instantiation applies unless ray origins are excluded (a first voxel that is not visited).  (An end
        // voxel that is walked -- kRfEndPointAsFree, clipped rays, TSDF -- is simply one more voxel of the ray's last
        // segment.)
        const bool special = (ray_flags & OHMHIP_RF_EXCLUDE_ORIGIN) != 0;
        const dim3 wgrid(std::min<uint32_t>(info.n_chunks, m->walk_workgroups)), wblock(kWalkThreads);
        const size_t wlds = walkLdsBytes(m->mc, m->chunk_segments);
        if (special)
        {
          hipLaunchKernelGGL((k_region_walk<true, false>), wgrid, wblock, wlds, s, wa);
        }
        else if (trace)
        {
          hipLaunchKernelGGL((k_region_walk<false, true>), wgrid, wblock, wlds, s, wa);
        }
        else
        {
          hipLaunchKernelGGL((k_region_walk<false, false>), wgrid, wblock, wlds, s, wa);
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
          hipLaunchKernelGGL(k_region_traversal, dim3(info.n_chunks), dim3(kWalkThreads), traversalLdsBytes(m->mc), s,
                             ta);
        }
        OHMHIP_CHECK(hipEventRecord(tev[3], s));
        if (occupancy_mode)
        {
          hipLaunchKernelGGL(k_flagged_events, dim3(4096), dim3(256), 0, s, batchScratch(m), events, event_capacity,
                             batchEventCount(m), sorted, m->d_miss_counts,
                             static_cast<uint32_t *>(m->interval_counts.ptr), m->mc.region_voxels,
                             reinterpret_cast<uint32_t *>(m->h_info_dev + 1));
          break;
        }
        // NDT / TSDF: the host needs the event count to size the sort; an overflowing list is re-walked.
        OHMHIP_CHECK(hipMemcpyAsync(&m->h_info[1], batchEventCount(m), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        OHMHIP_CHECK(hipStreamSynchronize(s));
        n_events = *reinterpret_cast<const uint32_t *>(&m->h_info[1]);
        m->event_demand = n_events;
        if (n_events <= event_capacity)
        {
          break;
        }
        // Overflow: undo the count flush, grow the key buffers (sample keys must be regenerated) and walk again.
        hipLaunchKernelGGL(k_clear_counts, dim3(info.n_touched), dim3(256), 0, s, m->mc, regionTable(m),
                           batchScratch(m), m->d_miss_counts);
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
        if (walk_attempt == 3)
        {
          return OHMHIP_ERR_INTERNAL;
        }
      }
    }
    else
    {
      OHMHIP_CHECK(hipEventRecord(tev[3], s));
    }
    OHMHIP_CHECK(hipEventRecord(m->ev[3], s));
    return OHMHIP_OK;
  }

  int applyOccupancy()
  {
    // (One launch for both halves -- k_apply_occupancy -- measured slower than the two below: 0.167 vs 0.147 ms for
    // sort + apply in C1; the sample replay wants small workgroups and few registers.)
    hipLaunchKernelGGL(k_apply_hits, dim3(ray_blocks), dim3(256), 0, s, m->mc, regionTable(m), batchScratch(m),
                       ray_flags, sorted, static_cast<uint32_t *>(m->interval_counts.ptr), m->d_miss_counts, d_rays,
                       static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]),
                       static_cast<uint32_t *>(m->layers[OHMHIP_LID_MEAN]), sec,
                       static_cast<const RayWalk *>(batchWalks(m).ptr));
    if (info.n_touched)
    {
      hipLaunchKernelGGL(k_apply_counts, dim3(info.n_touched), dim3(1024), 0, s, m->mc, regionTable(m),
                         batchScratch(m), ray_flags, m->d_miss_counts, m->d_hit_mask,
                         static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]), 1,
                         static_cast<uint32_t *>(nullptr), direct_segments, 0, sec.traversal,
                         sec.traversal ? m->d_traversal_acc : nullptr);
    }
    return OHMHIP_OK;
  }

  int replayEvents()
  {
    const size_t total = size_t(n_rays) + size_t(n_events);
    size_t sort_bytes = 0;
    OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(nullptr, sort_bytes, keys_a, keys_b, total, 0, sortEndBit(m), s));
    OHMHIP_CHECK(m->sort_temp.ensure(sort_bytes, false, s));
    size_t temp_bytes = m->sort_temp.bytes;
    OHMHIP_CHECK(rocprim::radix_sort_keys<SortConfig>(m->sort_temp.ptr, temp_bytes, keys_a, keys_b, total, 0,
                                                      sortEndBit(info.n_slots), s));
    // NDT: one lane per voxel group -- compact the group heads, then replay grid-stride over them (a voxel's event
    // list is long there and the maths heavy; a lane per event with the non-heads exiting ran at a few live lanes
    // per wave).
    uint32_t *heads = nullptr;
    uint32_t *n_heads = batchEventCount(m) + 2;
    uint32_t replay_blocks = uint32_t((total + 127) / 128);
    if (ndt_mode)
    {
      OHMHIP_CHECK(m->group_heads.ensure(sizeof(uint32_t) * total, false, s));
      heads = static_cast<uint32_t *>(m->group_heads.ptr);
      OHMHIP_CHECK(hipMemsetAsync(n_heads, 0, sizeof(uint32_t), s));
      hipLaunchKernelGGL(k_group_heads, dim3(uint32_t((total + kHeadsPerBlock - 1) / kHeadsPerBlock)), dim3(256), 0, s,
                         sorted, uint32_t(total), heads, n_heads);
      replay_blocks = uint32_t(std::min<size_t>(replay_blocks, size_t(m->walk_workgroups) * 32u));
    }
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
      hipLaunchKernelGGL(k_replay_ndt, dim3(replay_blocks), dim3(128), 0, s, m->mc, regionTable(m), sorted,
                         uint32_t(total), d_rays, d_intensities,
                         static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]),
                         static_cast<uint32_t *>(m->layers[OHMHIP_LID_MEAN]),
                         static_cast<float *>(m->layers[OHMHIP_LID_COVARIANCE]),
                         tm ? static_cast<float *>(m->layers[OHMHIP_LID_INTENSITY]) : nullptr,
                         tm ? static_cast<uint32_t *>(m->layers[OHMHIP_LID_HIT_MISS]) : nullptr, sec,
                         static_cast<const RayWalk *>(batchWalks(m).ptr), heads, n_heads);
      if (info.n_touched)
      {
        hipLaunchKernelGGL(k_apply_counts, dim3(info.n_touched), dim3(1024), 0, s, m->mc, regionTable(m),
                           batchScratch(m), 0u, m->d_miss_counts, m->d_hit_mask,
                           static_cast<float *>(m->layers[OHMHIP_LID_OCCUPANCY]), 0,
                           tm ? static_cast<uint32_t *>(m->layers[OHMHIP_LID_HIT_MISS]) : nullptr, direct_segments, 1,
                           sec.traversal, sec.traversal ? m->d_traversal_acc : nullptr);
      }
    }
    else
    {
      hipLaunchKernelGGL(k_replay_tsdf, dim3(replay_blocks), dim3(128), 0, s, m->mc, regionTable(m), sorted,
                         uint32_t(total), d_rays, static_cast<float *>(m->layers[OHMHIP_LID_TSDF]), heads, n_heads);
      if (info.n_touched)
      {
        hipLaunchKernelGGL(k_apply_counts_tsdf, dim3(info.n_touched), dim3(256), 0, s, m->mc, regionTable(m),
                           batchScratch(m), m->d_miss_counts, m->d_hit_mask,
                           static_cast<float *>(m->layers[OHMHIP_LID_TSDF]), direct_segments);
      }
    }
    return OHMHIP_OK;
  }

  int finish()
  {
    OHMHIP_CHECK(hipEventRecord(tev[4], s));
    OHMHIP_CHECK(hipEventRecord(m->ev_batch_done[m->parity], s));
    m->batch_done_recorded[m->parity] = true;
    OHMHIP_CHECK(hipGetLastError());

    m->stats = {};
    m->stats.rays_in = n_rays;
    m->stats.rays_integrated = info.rays_ok;
    m->stats.voxel_visits = info.visits;
    m->stats.ray_region_segments = info.n_segments;
    m->segments_per_ray = std::max(1.0, double(info.n_segments) / double(std::max<uint32_t>(n_rays, 1u)));
    m->stats.regions_touched = info.n_touched;
    m->stats.regions_resident = info.n_slots;
    m->stats_pending = true;
    ++m->batch_seq;
    return OHMHIP_OK;
  }
};

int integrateBatch(ohmhip_map_t m, const double *d_rays, const float *d_intensities, const double *d_timestamps,
                   uint32_t n_rays, unsigned ray_flags)
{
  BatchRun run{ m, d_rays, d_intensities, d_timestamps, n_rays, ray_flags, m->stream, m->front_stream,
                m->tev[m->batch_seq % kTimingRing] };
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
    OHMHIP_CHECK(run.commitRegions());
    scheduleWriteBack(m, uint32_t(m->batch_seq + 1u));  // (spill to host: keep the next eviction's victims clean)
    OHMHIP_CHECK(run.sizeBuffers());
    OHMHIP_CHECK(run.binAndOrder());
    OHMHIP_CHECK(run.walk());
    OHMHIP_CHECK(run.occupancy_mode ? run.applyOccupancy() : run.replayEvents());
    return run.finish();
  }
  return OHMHIP_ERR_CAPACITY;
}

inline char *slotRays(ohmhip_map_s::RaySlot &sl) { return sl.h; }
inline char *slotTimes(ohmhip_map_s::RaySlot &sl) { return sl.h + sl.capacity * 48; }
inline char *slotIntens(ohmhip_map_s::RaySlot &sl) { return sl.h + sl.capacity * 56; }
inline char *slotFilterFlags(ohmhip_map_s::RaySlot &sl) { return sl.h + sl.capacity * 60; }

/// Make room for `rays` rays in the filling slot, keeping what is pending in it.
int growRaySlot(ohmhip_map_t m, ohmhip_map_s::RaySlot &sl, size_t rays)
{
  if (rays <= sl.capacity)
  {
    return OHMHIP_OK;
  }
  ohmhip_map_s::RaySlot grown;
  grown.capacity = std::max<size_t>(rays + rays / 4, 4096);
  void *block = nullptr;
  OHMHIP_CHECK(hipHostMalloc(&block, grown.capacity * 61, hipHostMallocDefault));
  grown.h = static_cast<char *>(block);
  if (m->pending_rays)
  {
    std::memcpy(slotRays(grown), slotRays(sl), m->pending_rays * 48);
    std::memcpy(slotTimes(grown), slotTimes(sl), m->pending_rays * 8);
    std::memcpy(slotIntens(grown), slotIntens(sl), m->pending_rays * 4);
    std::memcpy(slotFilterFlags(grown), slotFilterFlags(sl), m->pending_rays);
  }
  if (sl.h)
  {
    OHMHIP_CHECK(hipHostFree(sl.h));
  }
  sl.h = grown.h;
  sl.capacity = grown.capacity;
  return OHMHIP_OK;
}

int integrateRaysDevice(ohmhip_map_t m, const double *d_rays, size_t element_count, const float *d_intensities,
                        const double *d_timestamps, unsigned ray_flags, size_t *integrated,
                        const unsigned char *d_filter_flags = nullptr);
int validateBatchRequest(ohmhip_map_t m, unsigned ray_flags);

/// Wait for the launch thread to finish the batch handed to it (ohmhip_map_set_async_launch) and collect its status.
int settleLaunch(ohmhip_map_t m)
{
  if (!m->launch_busy)
  {
    return OHMHIP_OK;
  }
  m->launch_thread->wait();
  m->launch_busy = false;
  const int err = m->launch_result;
  m->launch_result = OHMHIP_OK;
  return err;
}

/// Launch what the filling slot holds: H2D on the copy stream, the batch on the compute stream behind it.
/// `may_hand_over`: the caller is the host-pointer integrate call itself and needs nothing from the batch -- with
/// ohmhip_map_set_async_launch the launch sequence then runs on the map's thread.  Everybody else (the observers,
/// OHMHIP_SETTLE) gets the batch fully launched before this returns.
int flushPendingRays(ohmhip_map_t m, size_t *integrated = nullptr, bool may_hand_over = false)
{
  // One batch at a time is being launched; its error surfaces here.  The rays waiting in the filling slot are NOT that
  // batch's: they stay queued and run with the next flush -- which must then send the whole block again, because calls
  // appended from now on only reach the pinned block (a stale "uploaded" flag made that flush skip both the
  // resize of the device copy and the transfer).
  {
    const int settle_err = settleLaunch(m);
    if (settle_err != OHMHIP_OK)
    {
      m->ray_slots[m->fill_slot].rays_uploaded = false;
      return settle_err;
    }
  }
  const size_t n = m->pending_rays;
  if (n == 0)
  {
    return OHMHIP_OK;
  }
  ohmhip_map_s::RaySlot &sl = m->ray_slots[m->fill_slot];
  m->pending_rays = 0;
  m->pending_calls = 0;
  const bool on_device = m->pending_on_device;
  m->pending_on_device = false;
  const double *d_ts = nullptr;
  const float *d_int = nullptr;
  const unsigned char *d_ff = nullptr;
  if (on_device)
  {
    // (device-pointer calls: the copy stream has the device-to-device copies queued already)
    d_ts = m->pending_times ? static_cast<const double *>(sl.d_times.ptr) : nullptr;
    d_int = m->pending_intens ? static_cast<const float *>(sl.d_intens.ptr) : nullptr;
  }
  else
  {
    if (!sl.rays_uploaded)
    {
      OHMHIP_CHECK(sl.d_rays.ensure(n * 48, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_rays.ptr, slotRays(sl), n * 48, hipMemcpyHostToDevice, m->copy_stream));
    }
    sl.rays_uploaded = false;
    if (m->pending_times)
    {
      OHMHIP_CHECK(sl.d_times.ensure(n * 8, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_times.ptr, slotTimes(sl), n * 8, hipMemcpyHostToDevice, m->copy_stream));
      d_ts = static_cast<const double *>(sl.d_times.ptr);
    }
    if (m->pending_intens)
    {
      OHMHIP_CHECK(sl.d_intens.ensure(n * 4, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_intens.ptr, slotIntens(sl), n * 4, hipMemcpyHostToDevice, m->copy_stream));
      d_int = static_cast<const float *>(sl.d_intens.ptr);
    }
    if (m->pending_fflags)
    {
      OHMHIP_CHECK(sl.d_fflags.ensure(n, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_fflags.ptr, slotFilterFlags(sl), n, hipMemcpyHostToDevice, m->copy_stream));
      d_ff = static_cast<const unsigned char *>(sl.d_fflags.ptr);
    }
  }
  OHMHIP_CHECK(hipEventRecord(sl.uploaded, m->copy_stream));
  OHMHIP_CHECK(hipStreamWaitEvent(m->stream, sl.uploaded, 0));
  OHMHIP_CHECK(hipStreamWaitEvent(m->front_stream, sl.uploaded, 0));  // (the set-up pass reads the rays first)
  if (may_hand_over && m->async_launch && !on_device && !integrated)
  {
    // The launch sequence blocks on the batch's plan summary in its middle: it runs on the launch thread, the caller
    // goes on (typically to stage its next block into the other slot, whose upload then runs beside this wait).
    if (!m->launch_thread)
    {
      m->launch_thread.reset(new StagePool(1));
    }
    ohmhip_map_s::RaySlot *slot = &sl;
    const double *d_r = static_cast<const double *>(sl.d_rays.ptr);
    const unsigned flags = m->pending_flags;
    sl.in_flight = true;
    m->fill_slot ^= 1;
    m->launch_busy = true;
    m->launch_thread->start(1, [m, slot, d_r, n, d_int, d_ts, flags, d_ff](unsigned) {
      int err = int(hipSetDevice(m->device));
      if (err == 0)
      {
        err = integrateRaysDevice(m, d_r, n * 2, d_int, d_ts, flags, nullptr, d_ff);
      }
      const int rec = int(hipEventRecord(slot->done, m->stream));
      m->launch_result = err ? err : rec;
    });
    return OHMHIP_OK;
  }
  const int err = integrateRaysDevice(m, static_cast<const double *>(sl.d_rays.ptr), n * 2, d_int, d_ts,
                                      m->pending_flags, integrated, d_ff);
  OHMHIP_CHECK(hipEventRecord(sl.done, m->stream));
  sl.in_flight = true;
  m->fill_slot ^= 1;
  return err;
}

#define OHMHIP_SETTLE(m)                          \
  if (m)                                          \
  {                                               \
    const int settle_err_ = flushPendingRays(m);  \
    if (settle_err_ != OHMHIP_OK)                 \
    {                                             \
      return settle_err_;                         \
    }                                             \
  }
}  // namespace

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
  if ((err = hipStreamCreateWithFlags(&m->front_stream, hipStreamNonBlocking)) != 0 ||
      (err = hipEventCreateWithFlags(&m->ev_batch_done[0], hipEventDisableTiming)) != 0 ||
      (err = hipEventCreateWithFlags(&m->ev_batch_done[1], hipEventDisableTiming)) != 0 ||
      (err = hipEventCreateWithFlags(&m->ev_bin_done, hipEventDisableTiming)) != 0)
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
  if (const char *env = std::getenv("OHMHIP_BIN_RAYS"))
  {
    m->bin_rays_per_block = uint32_t(std::max(128, std::min(int(kBinRaysPerBlock), std::atoi(env))));
  }
  if (const char *env = std::getenv("OHMHIP_MIN_CHUNK_SEGMENTS"))
  {
    m->min_chunk_segments = uint32_t(std::max(64, std::min(int(kMaxChunkSegments), std::atoi(env))));
  }
  // The walk kernel keeps a region's count tile, the staged samples and the chunk's segment order in LDS (about
  // 150 KiB of the CU's 160 KiB for 32^3 regions): the chunk size gives way if the region tile is large.
  while (walkLdsBytes(mc, m->chunk_segments) > size_t(160) * 1024 && m->chunk_segments > 64)
  {
    m->chunk_segments /= 2;
  }
  const size_t lds_bytes = walkLdsBytes(mc, m->chunk_segments);
  const void *walk_kernels[3] = { reinterpret_cast<const void *>(k_region_walk<false, false>),
                                  reinterpret_cast<const void *>(k_region_walk<true, false>),
                                  reinterpret_cast<const void *>(k_region_walk<false, true>) };
  for (const void *kernel : walk_kernels)
  {
    if ((err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes))) != 0)
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
  }
  if (const char *env = std::getenv("OHMHIP_REFILL_MIN_IDLE"))
  {
    m->refill_min_idle = std::max(1, std::min(64, std::atoi(env)));
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
  if (m->h_use)
  {
    (void)hipHostFree(m->h_use);
  }
  for (auto &ring : m->wb_ring)
  {
    ring.jobs.release();
    if (ring.done)
    {
      (void)hipEventDestroy(ring.done);
    }
  }
  m->partition.table_dev.release();
  m->partition.masks.release();
  m->partition.block_counts.release();
  m->partition.totals.release();
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
                 "store-growth %.1f\n",
                 (unsigned long long)m->evictions, (unsigned long long)m->readmissions, m->spill_ms[0], m->spill_ms[1],
                 m->spill_ms[2], m->spill_ms[3], m->spill_ms[5]);
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
  for (hipEvent_t e : { m->ev_batch_done[0], m->ev_batch_done[1], m->ev_bin_done })
  {
    if (e)
    {
      (void)hipEventDestroy(e);
    }
  }
  delete m;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

}  // extern "C"

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
          for (int k = 0; k < 25; ++k)
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
  if ((m->debug_flags & 256u) && m->batch_seq >= 4)
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
  hipEvent_t *tev = m->tev[(m->batch_seq - 1 - batches_back) % kTimingRing];
  OHMHIP_CHECK(hipEventSynchronize(tev[4]));
  float sort_ms = 0, apply_ms = 0, front_ms = 0, bin_ms = 0;
  OHMHIP_CHECK(hipEventElapsedTime(&ms[0], tev[0], tev[4]));
  // The set-up pass of a batch runs on its own stream under the previous batch's last kernels, so back-to-back batches
  // complete at intervals shorter than first start -> last end; that interval is the device time the batch cost.
  if (uint64_t(batches_back) + 1 < m->batch_seq && batches_back + 1 < kTimingRing)
  {
    hipEvent_t *prev = m->tev[(m->batch_seq - 2 - batches_back) % kTimingRing];
    float period = 0;
    if (hipEventElapsedTime(&period, prev[4], tev[4]) == hipSuccess && period > 0 && period < ms[0])
    {
      ms[0] = period;
    }
  }
  OHMHIP_CHECK(hipEventElapsedTime(&front_ms, tev[0], tev[5]));
  OHMHIP_CHECK(hipEventElapsedTime(&bin_ms, tev[6], tev[1]));
  ms[1] = front_ms + bin_ms;
  OHMHIP_CHECK(hipEventElapsedTime(&ms[2], tev[2], tev[3]));
  OHMHIP_CHECK(hipEventElapsedTime(&sort_ms, tev[1], tev[2]));
  OHMHIP_CHECK(hipEventElapsedTime(&apply_ms, tev[3], tev[4]));
  ms[3] = sort_ms + apply_ms;
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

/// Drop resident regions from the pool (ohmhip_map_remove_regions; also the second half of an eviction).
int removeResidentRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, size_t *removed)
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
    OHMHIP_CHECK(hipStreamSynchronize(m->copy_stream));
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

int evictColdRegions(ohmhip_map_t m, uint32_t want_free, uint32_t max_evict)
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
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rank[a] > rank[b]; });
  // The victims' content goes straight from the pool into pinned store records, all regions and layers by ONE kernel
  // that writes the mapped host memory itself (k_copy_jobs); the compute stream is idle here -- it was drained above.
  lap(0, t_mark);
  OHMHIP_CHECK(reserveStoreRecords(m, k));
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
    const int err = int(hipStreamSynchronize(m->copy_stream));
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
int growPoolForNamedRegions(ohmhip_map_t m, uint32_t total, uint32_t keep)
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
int makeRoomForNamedRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count)
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
int readmitSpilledSlots(ohmhip_map_t m, uint32_t first_slot, uint32_t end_slot)
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
int readmitSpilledKeys(ohmhip_map_t m, const int16_t *keys_xyz, size_t count)
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
