// map_state.h -- the resident map object of libohmhip.so: device buffers, the staging thread pool, `struct ohmhip_map_s`
// and the forward declarations the parts of the translation unit share.
//
// Part of ohmhip_map.hip's translation unit (included there, in order): not a stand-alone header.
#ifndef OHMHIP_MAP_STATE_H
#define OHMHIP_MAP_STATE_H

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
  hipStream_t wb_stream = nullptr;    ///< background write-back of the spill path (created on first use): its device-to-host
                                      ///< copies must not delay the re-admissions queued on copy_stream
  /// Stream of a batch's set-up pass (k_ray_setup, k_plan).  It reads the rays and the region table only, and writes
  /// per-batch scratch that exists twice (see `parity`), so the set-up of batch N+1 runs beside the sample sort of batch
  /// N and in the CUs its walk kernel vacates.  It is idle whenever no batch call is in progress: every call waits for
  /// its own plan summary.
  hipStream_t front_stream = nullptr;
  /// Cross-stream ordering uses the STOP EVENTS of the kernels themselves (hipExtLaunchKernelGGL binds an event to the
  /// kernel's own completion signal: free), never a hipEventRecord behind a kernel -- on gfx950 / ROCm 7.2 a record is a
  /// barrier packet that idles the queue for 3-7 us before the next kernel starts (scripts/probes/event_probe.hip,
  /// profiles/r05_event_probe.txt; round 4 paid eight of them per batch).  The events live in the timing ring below.
  hipEvent_t batch_done_event[2] = { nullptr, nullptr };  ///< per parity: stop event of the last kernel of the batch that last used this scratch copy
  hipEvent_t bin_done_event = nullptr;  ///< stop event of the latest k_ray_bin
  uint32_t parity = 0;  ///< which copy of the doubled per-batch scratch (RayWalk array, per-hash / per-slot counters,
                        ///< chunk list, event counters) the current batch uses
  hipEvent_t ev[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  /// Events of the last kTimingRing batches: [1] binned, [2] samples ordered (the kernel before the walk), [3] walked,
  /// [4] batch done, [5] plan done -- all stop events of kernels --, and with `phase_timing` [0] set-up pass starts, [6]
  /// binning starts (records: they cost the batch a few microseconds each).  tev_mask: which of them the batch recorded;
  /// tev_pre_walk: the event that marks the start of the batch's walk phase ([2], or [1] when nothing ran in between).
  hipEvent_t tev[kTimingRing][7] = {};
  uint8_t tev_mask[kTimingRing] = {};
  uint8_t tev_pre_walk[kTimingRing] = {};
  bool phase_timing = false;  ///< ohmhip_map_set_phase_timing / OHMHIP_PHASE_TIMING=1 / OHMHIP_DEBUG_FLAGS & 256
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
    hipStream_t route_stream = nullptr;  ///< routing runs beside the batches in flight: it reads no map state
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
  /// Regions / tiles of at most 4 096 voxels (16^3) are walked by the WalkHalf shape of k_region_walk: 512-thread workgroups with
  /// half of everything, two per CU (occupancy_kernels.h; OHMHIP_WALK_HALF=0 keeps the full shape for A/B runs).
  bool walk_half = false;
  uint32_t walkSlots() const { return walk_workgroups * (walk_half ? 2u : 1u); }  ///< persistent walk workgroups of a launch
  unsigned long long *d_dbg = nullptr;  ///< 8 debug counters (OHMHIP_DEBUG_FLAGS & 64)
  double first_ray_time = -1.0;  ///< OccupancyMap::firstRayTime() (ohm/OccupancyMap.cpp:343-347)
  uint32_t event_demand = 0;
  uint32_t event_limit = 0;  ///< OHMHIP_EVENT_LIMIT (tests): cap of the NDT / TSDF event list's first sizing
  bool spec_bucket_ok = false;  ///< the previous occupancy batch used the per-region sample sort: bin speculatively
  uint32_t spec_max_region_hits = 0;  ///< ... and its densest region held this many samples (which sort kernels to launch)
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
  uint64_t rays_beyond_tiles = 0;  ///< ohmhip_map_rays_beyond_tiles (tiled maps: rays cut for key-range reasons)
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
    void *jobs_host = nullptr;  ///< pinned, device visible: the kernel reads its job list straight from here (no copy call:
                                ///< a blocking hipMemcpy from pageable memory would stall the batch pipeline)
    void *jobs_dev = nullptr;
    size_t capacity = 0;        ///< bytes
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
  uint32_t writeback_workgroups = 32;  ///< grid of the background copy kernel (OHMHIP_WRITEBACK_WGS): the CUs it may hold
  bool spill_enabled = false;
  /// The last batch failed with OHMHIP_ERR_CAPACITY because it alone touches more regions than the residency limit
  /// holds (not: hash full, device memory, slot field) -- the cause integrateRaysDevice answers by splitting the batch.
  bool batch_exceeds_limit = false;
  uint64_t evictions = 0, readmissions = 0;
  double wb_host_ms = 0;  ///< OHMHIP_DEBUG_FLAGS & 512: host time spent scheduling write-backs
  double spill_ms[6] = { 0, 0, 0, 0, 0, 0 };  ///< OHMHIP_DEBUG_FLAGS & 512: evict select / copy / compact, readmit copy, failed attempts, store growth
};

// Background write-back of the spill path (writeback_impl.h).
namespace
{
int queueUseStamps(ohmhip_map_t m, hipStream_t stream);
void scheduleWriteBack(ohmhip_map_t m, uint32_t now);
void dropPrecleaned(ohmhip_map_t m);
int drainWriteBack(ohmhip_map_t m);
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

// Defined further down (they use the region read / remove machinery of the C ABI section).  Internal linkage: the
// shared library exports the C ABI of include/ohmhip.h and nothing else (tests/test_cabi.py).
static int removeResidentRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, size_t *removed);
static int evictColdRegions(ohmhip_map_t m, uint32_t want_free, uint32_t max_evict = 0xffffffffu);
static int makeRoomForNamedRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count);
static int growPoolForNamedRegions(ohmhip_map_t m, uint32_t total, uint32_t keep);
static int readmitSpilledSlots(ohmhip_map_t m, uint32_t first_slot, uint32_t end_slot);
static int readmitSpilledKeys(ohmhip_map_t m, const int16_t *keys_xyz, size_t count);


#endif  // OHMHIP_MAP_STATE_H
