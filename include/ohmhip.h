/*
 * ohmhip.h -- C ABI of libohmhip.so: the MI355X (gfx950) replacement for the part of ohm's `gputil` + `ohmgpu`
 * that sits under ohm::GpuMap::integrateRays / GpuNdtMap / GpuTsdfMap.
 *
 * Plain C: opaque handles, POD structs, pointers and sizes.  No C++/torch types cross this boundary and no
 * exception does either: every function returns an int status (0 == OHMHIP_OK, negative == ohmhip error,
 * positive == hipError_t value); ohmhip_error_string() renders it.
 *
 * Citations are file:line in the reference checkout (csiro-robotics/ohm).  Two groups:
 *   1. device plumbing  -- what ohmgpu's hot path uses of gputil::Device/Queue/Event/Buffer/PinnedBuffer;
 *   2. the typed hot path -- one resident voxel map per handle, ray batches in, region layers out.  This replaces
 *      the generic `gputil::Kernel` variadic launch of regionRayUpdateOccupancy, regionRayUpdateNdt, covarianceHitNdt and tsdfRayUpdate
 *      (ohmgpu/GpuMap.cpp:1130-1164, ohmgpu/GpuNdtMap.cpp:383-486, ohmgpu/GpuTsdfMap.cpp:262-263) and the
 *      GpuLayerCache upload/download protocol (ohmgpu/GpuLayerCache.cpp:172-182, 300-321, 429-633).
 */
#ifndef OHMHIP_H
#define OHMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CORE ABI.  The header has grown to 86 entry points over six rounds; a reference-side binding of ohm::GpuMap /
 * GpuNdtMap / GpuTsdfMap / GpuCache needs FIFTEEN of them.  The list below is exact: it is every ohmhip_* call made by
 * ohm_amd/host/ref_adaptor/private/HipBindingCore.cpp, the compiled and GPU-tested logic of the Level-2 adaptor
 * (INTEGRATION.md), and tests/test_cabi.py keeps the two in step.
 *   OHMHIP_CORE_ABI: ohmhip_error_string ohmhip_map_config_default ohmhip_map_create ohmhip_map_destroy
 *   OHMHIP_CORE_ABI: ohmhip_map_update_config ohmhip_map_integrate_rays ohmhip_map_integrate_rays_filtered ohmhip_map_sync
 *   OHMHIP_CORE_ABI: ohmhip_map_write_regions ohmhip_map_dirty_regions ohmhip_map_read_regions ohmhip_map_clear_dirty
 *   OHMHIP_CORE_ABI: ohmhip_map_clear ohmhip_map_remove_regions ohmhip_map_cache_stats
 * (the gputil::Device / Queue / Event / Buffer backend adds group 1 below; ohmhip_map_integrate_rays_device and
 * ohmhip_transform_samples serve GpuTransformSamples callers.)  Everything else is optional: residency limit and spill,
 * region listings and zero-copy views, LineKeysQueryGpu, multi-GPU (group 3).  Entry points that only tune or measure
 * -- knobs found useful while optimising, nothing a binding has to know -- carry OHMHIP_EXPERIMENTAL: they may change
 * or go without notice. */
#define OHMHIP_EXPERIMENTAL

#define OHMHIP_OK 0
#define OHMHIP_ERR_INVALID_ARG (-1)
#define OHMHIP_ERR_NO_DEVICE (-2)
#define OHMHIP_ERR_CAPACITY (-3)    /* region pool exhausted and could not grow */
#define OHMHIP_ERR_UNSUPPORTED (-4) /* flag / layout not supported by the HIP path */
#define OHMHIP_ERR_NOT_FOUND (-5)
#define OHMHIP_ERR_INTERNAL (-6)
#define OHMHIP_ERR_PEER (-7)        /* a collective call was abandoned because ANOTHER rank failed in it */

const char *ohmhip_error_string(int status);
/* Identifies the build: 16 hex digits over the library's sources (set by __graft_entry__.build()), "unversioned" for
 * other builds.  Recorded with profiles so a counter profile of another build is not quoted for this one. */
const char *ohmhip_build_id(void);

/* ------------------------------------------------------------------------------------------------------------------ */
/* 1. Device plumbing (replaces gputil)                                                                               */
/* ------------------------------------------------------------------------------------------------------------------ */
typedef struct ohmhip_stream_s *ohmhip_stream_t; /* gputil::Queue  (gputil/gpuQueue.h:39)  */
typedef struct ohmhip_event_s *ohmhip_event_t;   /* gputil::Event  (gputil/gpuEvent.h:23)  */
typedef struct ohmhip_buffer_s *ohmhip_buffer_t; /* gputil::Buffer (gputil/gpuBuffer.h:73) */

typedef struct ohmhip_device_info
{
  char name[256];
  char arch[64];
  uint64_t total_memory;      /* gputil::Device::deviceMemory()       gputil/gpuDevice.h:163-171 */
  uint64_t max_allocation;    /* gputil::Device::maxAllocationSize()  */
  int compute_units;
  int lds_bytes_per_block;
  int unified_memory;         /* gputil::Device::unifiedMemory()      */
} ohmhip_device_info;

int ohmhip_device_count(int *count);                            /* gputil/cuda/gpuDevice.cpp:108-115 */
int ohmhip_device_select(int device);                           /* gputil/cuda/gpuKernel.cpp:33 (cudaSetDevice) */
int ohmhip_device_get_info(int device, ohmhip_device_info *info);
int ohmhip_device_synchronize(void);                            /* cudaDeviceSynchronize, gputil/cuda/gpuBuffer.cpp (pin) */

int ohmhip_stream_create(ohmhip_stream_t *stream);              /* gputil/cuda/gpuDevice.cpp:156 */
int ohmhip_stream_destroy(ohmhip_stream_t stream);              /* gputil/cuda/gpuQueue.cpp:22  */
int ohmhip_stream_finish(ohmhip_stream_t stream);               /* Queue::finish   gputil/cuda/gpuQueue.cpp:114 */
int ohmhip_stream_wait_event(ohmhip_stream_t stream, ohmhip_event_t event); /* gputil/cuda/gpuKernel.cpp:91 */

int ohmhip_event_create(ohmhip_event_t *event);                 /* gputil/cuda/gpuEvent.cpp:150 (blocking sync) */
int ohmhip_event_destroy(ohmhip_event_t event);                 /* gputil/cuda/gpuEvent.cpp:22  */
int ohmhip_event_record(ohmhip_event_t event, ohmhip_stream_t stream); /* Queue::mark gputil/gpuQueue.h:68-96 */
int ohmhip_event_wait(ohmhip_event_t event);                    /* Event::wait      gputil/cuda/gpuEvent.cpp:95 */
int ohmhip_event_is_complete(ohmhip_event_t event, int *complete); /* Event::isComplete gputil/cuda/gpuEvent.cpp:76 */
int ohmhip_event_elapsed_ms(ohmhip_event_t start, ohmhip_event_t stop, float *ms);

/* Buffer flags: gputil/gpuBuffer.h:24-45 */
#define OHMHIP_BF_READ (1u << 0)
#define OHMHIP_BF_WRITE (1u << 1)
#define OHMHIP_BF_HOST_ACCESS (1u << 2) /* pinned host allocation (cudaHostAlloc, gputil/cuda/gpuBuffer.cpp:249) */

int ohmhip_buffer_create(ohmhip_buffer_t *buffer, size_t bytes, unsigned flags); /* gputil/cuda/gpuBuffer.cpp:240 */
int ohmhip_buffer_destroy(ohmhip_buffer_t buffer);                               /* :273-276 */
int ohmhip_buffer_resize(ohmhip_buffer_t buffer, size_t bytes, size_t *actual);  /* grow-only, gpuBuffer.h:161 */
int ohmhip_buffer_size(ohmhip_buffer_t buffer, size_t *bytes);
int ohmhip_buffer_ptr(ohmhip_buffer_t buffer, void **device_ptr);                /* Buffer::argPtr */
/* stream == NULL => synchronous (gputil/cuda/gpuBuffer.cpp:78); otherwise async + optional completion event (:58-67) */
int ohmhip_buffer_write(ohmhip_buffer_t buffer, const void *src, size_t bytes, size_t dst_offset,
                        ohmhip_stream_t stream, ohmhip_event_t block_on, ohmhip_event_t completion);
int ohmhip_buffer_read(ohmhip_buffer_t buffer, void *dst, size_t bytes, size_t src_offset, ohmhip_stream_t stream,
                       ohmhip_event_t block_on, ohmhip_event_t completion);
int ohmhip_buffer_fill(ohmhip_buffer_t buffer, int byte_value, size_t bytes, size_t offset,
                       ohmhip_stream_t stream);                                  /* gputil/cuda/gpuBuffer.cpp:206 */
/* Buffer::fill / fillPartial with an arbitrary pattern (gputil/gpuBuffer.h:199-236): `bytes` from `offset` are filled
 * with repetitions of the pattern, the last one cut short if it does not fit.  Asynchronous when `stream` is given. */
int ohmhip_buffer_fill_pattern(ohmhip_buffer_t buffer, const void *pattern, size_t pattern_size, size_t bytes,
                               size_t offset, ohmhip_stream_t stream, ohmhip_event_t block_on,
                               ohmhip_event_t completion);
/* gputil::copyBuffer (gputil/gpuBuffer.h:459-510): device-side copy between two buffers. */
int ohmhip_buffer_copy(ohmhip_buffer_t dst, size_t dst_offset, ohmhip_buffer_t src, size_t src_offset, size_t bytes,
                       ohmhip_stream_t stream, ohmhip_event_t block_on, ohmhip_event_t completion);
int ohmhip_buffer_flags(ohmhip_buffer_t buffer, unsigned *flags);                /* Buffer::flags */
/* Pinned host staging memory (gputil::PinnedBuffer, gputil/cuda/gpuPinnedBuffer.cpp:66-131). */
int ohmhip_host_alloc(void **ptr, size_t bytes);
int ohmhip_host_free(void *ptr);

/* ------------------------------------------------------------------------------------------------------------------ */
/* 2. The hot path: a device-resident voxel map                                                                       */
/* ------------------------------------------------------------------------------------------------------------------ */

/* Voxel layers (ohm/DefaultLayer.cpp:76-311).  Bit i <=> layer id i. */
enum ohmhip_layer_id
{
  OHMHIP_LID_OCCUPANCY = 0,  /* float, clear +inf                        */
  OHMHIP_LID_MEAN = 1,       /* VoxelMean {u32 coord, u32 count}         ohm/VoxelMeanCompute.h:29-33 */
  OHMHIP_LID_COVARIANCE = 2, /* CovarianceVoxel float[6]                 ohm/CovarianceVoxelCompute.h:56-66 */
  OHMHIP_LID_TRAVERSAL = 3,  /* float                                    */
  OHMHIP_LID_TOUCH_TIME = 4, /* u32 ms since first ray                   */
  OHMHIP_LID_INCIDENT = 5,   /* u32 packed normal                        */
  OHMHIP_LID_INTENSITY = 6,  /* IntensityMeanCov {float, float}          */
  OHMHIP_LID_HIT_MISS = 7,   /* HitMissCount {u32, u32}                  */
  OHMHIP_LID_TSDF = 8,       /* VoxelTsdf {float weight, float distance} ohm/VoxelTsdfCompute.h:20-24 */
  OHMHIP_LID_COUNT = 9
};
#define OHMHIP_LAYER_BIT(id) (1u << (id))

/* Which update rule integrateRays applies: GpuMap / GpuNdtMap / GpuTsdfMap. */
enum ohmhip_map_mode
{
  OHMHIP_MODE_OCCUPANCY = 0, /* ohmgpu/GpuMap.h:143     (CPU semantics: ohm/RayMapperOccupancy.cpp:68-339) */
  OHMHIP_MODE_NDT_OM = 1,    /* ohmgpu/GpuNdtMap.h:63   (ohm/RayMapperNdt.cpp:84-407, NdtMode::kOccupancy)  */
  OHMHIP_MODE_NDT_TM = 2,    /* NdtMode::kTraversability                                                   */
  OHMHIP_MODE_TSDF = 3       /* ohmgpu/GpuTsdfMap.h:37  (ohm/RayMapperTsdf.cpp:87-182)                      */
};

/* Ray flags, bit-compatible with ohm::RayFlag (ohm/RayFlag.h:16-60). */
#define OHMHIP_RF_DEFAULT 0u
#define OHMHIP_RF_END_POINT_AS_FREE (1u << 0)
#define OHMHIP_RF_STOP_ON_FIRST_OCCUPIED (1u << 1) /* exact, via per-ray stop iteration over the sorted visits of the
                                                      batch (slow path); also with a traversal layer */
#define OHMHIP_RF_EXCLUDE_ORIGIN (1u << 2)
#define OHMHIP_RF_EXCLUDE_SAMPLE (1u << 3)
#define OHMHIP_RF_EXCLUDE_RAY (1u << 4)
#define OHMHIP_RF_EXCLUDE_UNOBSERVED (1u << 5)
#define OHMHIP_RF_EXCLUDE_FREE (1u << 6)
#define OHMHIP_RF_EXCLUDE_OCCUPIED (1u << 7)
#define OHMHIP_RF_REVERSE_WALK (1u << 8) /* accepted and ignored: results follow the CPU forward walk */

/* Ray filter applied on device before integration (ohm/RayFilter.cpp:12-58; map default = good rays <= 1e10,
 * ohm/OccupancyMap.cpp:215-218). */
enum ohmhip_ray_filter
{
  OHMHIP_FILTER_NONE = 0,
  OHMHIP_FILTER_GOOD = 1,
  OHMHIP_FILTER_CLIP = 2
};

typedef struct ohmhip_map_config
{
  double resolution;          /* OccupancyMap::resolution()                              */
  int region_dim[3];          /* OccupancyMap::regionVoxelDimensions(); 0 => 32          */
  double origin[3];           /* OccupancyMap::origin()                                  */
  unsigned layers;            /* OHMHIP_LAYER_BIT mask                                   */
  int mode;                   /* ohmhip_map_mode                                         */
  float hit_value;            /* OccupancyMap::hitValue()   (log odds)                   */
  float miss_value;           /* OccupancyMap::missValue()                               */
  float threshold_value;      /* OccupancyMap::occupancyThresholdValue()                 */
  float min_value, max_value; /* OccupancyMap::min/maxVoxelValue()                       */
  int saturate_at_min, saturate_at_max;
  int ray_filter;             /* ohmhip_ray_filter                                       */
  double ray_filter_range;
  /* NDT (ohm/private/NdtMapDetail.h:20-45) */
  float ndt_sensor_noise;
  unsigned ndt_sample_threshold;
  float ndt_adaptation_rate;
  float ndt_reinit_threshold;
  unsigned ndt_reinit_count;
  float ndt_initial_intensity_cov;
  /* TSDF (ohm/VoxelTsdf.h:27-37) */
  float tsdf_max_weight, tsdf_trunc, tsdf_dropoff, tsdf_sparsity;
  /* Residency: the whole map lives in HBM.  region_capacity regions are preallocated per layer (0 => derived from
   * gpu_mem_size, itself defaulting to 4 GiB; cf. GpuCache default 1 GiB, ohmgpu/GpuCache.h:90). The pool grows by
   * doubling when exhausted. */
  uint64_t gpu_mem_size;
  uint32_t region_capacity;
} ohmhip_map_config;

typedef struct ohmhip_map_s *ohmhip_map_t;

/* Timing / accounting of the most recent integrate call (device side measured with hipEvents on the map's stream). */
typedef struct ohmhip_batch_stats
{
  uint64_t rays_in;          /* rays submitted                                                           */
  uint64_t rays_integrated;  /* rays that passed the filter                                              */
  uint64_t voxel_visits;     /* exact count of miss visits + sample updates (== CPU walk visit count)    */
  uint64_t ray_region_segments;
  uint32_t regions_touched;
  uint32_t regions_resident;
  float ms_total;            /* first kernel start -> last kernel end                                    */
  float ms_setup;            /* ray setup + region binning                                               */
  float ms_walk;             /* the region line-walk kernel (dominant)                                   */
  float ms_apply;            /* hit sort + ordered apply                                                 */
} ohmhip_batch_stats;

/* LARGE REGIONS.  region_dim takes what the reference takes: any 1..255 voxels per axis (ohm/OccupancyMap.h:287,
 * glm::u8vec3; default 32^3, :24-26).  A region of up to 32768 voxels is one LDS tile of the walk kernel.  A larger one
 * is cut, inside the library, into equal tiles that each are one contiguous piece of the region's MapChunk block -- z
 * slabs of whole x-y layers (64^3 -> 8 tiles of 64 x 64 x 8), or, when one layer alone exceeds a tile, y strips of single
 * layers (255 x 255 x n -> 255 x 85 x 1) -- and everything internal works per tile.  The ABI keeps speaking REGIONS:
 * keys, listings, dirty sets and the blocks of read_regions / write_regions are the caller's regions (a tile no ray has
 * reached reads as cleared), results are those of the CPU mappers run with the same region size.  What counts tiles
 * instead: ohmhip_batch_stats::regions_touched / regions_resident, ohmhip_cache_stats, the memory limit's granule.
 * Not available for such maps (OHMHIP_ERR_UNSUPPORTED): the slot-level plumbing ohmhip_map_region_slot /
 * _ensure_regions / the replica merge -- the partitioned map works.  One limit: tile coordinates (region coordinate x
 * tiles per region on that axis) share the packed key's 16-bit fields, so with t tiles per region along an axis only
 * region coordinates within +-32767 / t are addressable there (64^3 at 0.1 m: +-26 km in z); rays beyond are rejected by
 * the key test like rays beyond the reference's own int16 region range. */
void ohmhip_map_config_default(ohmhip_map_config *config); /* reference defaults, ohm/OccupancyMap.cpp:192-223 */
int ohmhip_map_create(ohmhip_map_t *map, const ohmhip_map_config *config); /* GpuMap ctor + gpumap::enableGpu,
                                                                              ohmgpu/GpuMap.cpp:272, 106-122 */
int ohmhip_map_destroy(ohmhip_map_t map);
/* The reference GpuMap reads the OccupancyMap's probabilities, clamps and NDT / TSDF parameters at every launch
 * (ohmgpu/GpuMap.cpp:1036-1191, GpuNdtMap.cpp:289-503), so setters called on the map after the GpuMap exists take
 * effect from the next batch.  Same here: pass the configuration again; resolution, region dimensions, origin, mode and
 * layer set must equal those of creation (OHMHIP_ERR_INVALID_ARG otherwise), everything else applies to batches
 * presented afterwards.  One exception: the TSDF truncation distance of a map that already holds regions cannot
 * change (OHMHIP_ERR_UNSUPPORTED; free-space TSDF updates are exact only against one truncation distance). */
int ohmhip_map_update_config(ohmhip_map_t map, const ohmhip_map_config *config);

/* GpuMap::integrateRays (ohmgpu/GpuMap.cpp:416, 540-875): rays = element_count dvec3 (origin, sample pairs).
 * Host-pointer form stages through one of two pinned blocks and uploads on a side stream, so batch N+1 is copied while
 * batch N runs; returns after enqueue (like the reference the call is asynchronous; ohmhip_map_sync / any read is
 * the fence).  *integrated = points accepted (2 per ray). */
int ohmhip_map_integrate_rays(ohmhip_map_t map, const double *rays, size_t element_count, const float *intensities,
                              const double *timestamps, unsigned ray_flags, size_t *integrated);
/* GpuMap::setRayFilter with an arbitrary RayFilterFunction (ohm/RayFilter.h:45, ohmgpu/GpuMap.cpp:348-369; applied
 * per ray on the host at GpuMap.cpp:736-746).  The filter is host code, so the host mirror runs it and hands over what
 * passed: `rays` holds the accepted (possibly moved) origin / sample pairs and filter_flags[ray] the RayFilterFlag bits
 * the filter set (ohm/RayFilter.h:21-29: 1 invalid, 2 clipped start, 4 clipped end).  The map's built-in filter is not applied to
 * such a batch; a clipped end makes the end voxel part of the ray and suppresses the sample, as in the CPU mappers
 * (ohm/RayMapperOccupancy.cpp:209-223, ohm/RayMapperNdt.cpp:251-269). */
int ohmhip_map_integrate_rays_filtered(ohmhip_map_t map, const double *rays, size_t element_count,
                                       const float *intensities, const double *timestamps, unsigned ray_flags,
                                       const unsigned char *filter_flags, size_t *integrated);
/* Small host batches (the reference tools present 4096 rays per call, ohmapp/OhmAppGpu.cpp:187-207) would cost a full
 * pipeline pass each.  Consecutive host-pointer batches with the same flags and the same optional arrays are therefore
 * collected in the pinned staging block and run as ONE device batch once min_rays have accumulated -- or as soon as
 * anything observes the map (sync, stats, region reads, a device-pointer batch ...).  The result is the one the
 * separate calls give: the CPU mappers integrate ray by ray, so call boundaries carry no meaning, except for the
 * traversal layer (its exit range is carried within a call): maps with that layer never merge batches.  Every call
 * still reports its own *integrated: the ray filter's verdict is evaluated on the host while the rays are staged, with
 * the arithmetic the device uses.  An error of a deferred batch (pool exhausted ...) surfaces at the call that launches
 * it.  Default min_rays: 65536; 0 launches every call's batch in that call. */
OHMHIP_EXPERIMENTAL int ohmhip_map_set_batch_coalescing(ohmhip_map_t map, size_t min_rays);
/* Large host batches, opt-in: with enable != 0 a host-pointer call that is a device batch on its own returns as soon
 * as its rays are staged and their upload is queued; the device launch sequence -- which waits for the batch's plan
 * summary in its middle -- runs on a thread the map owns, so the caller stages its next block, and the link carries it,
 * beside that wait (1 M-ray calls: 1.4 -> ~1.1 ms).  The reference's GpuMap::integrateRays returns with
 * its GPU work in flight in the same way (ohmgpu/GpuMap.cpp:874).  What changes for the caller: *integrated is the
 * host's evaluation of the ray filter (the same verdict), and an error of the batch (OHMHIP_ERR_CAPACITY ...) is
 * returned by the NEXT call on the map that settles it -- any entry point but the staging part of integrate_rays; the
 * rays of the call that reports it stay queued.  Everything that observes the map waits for the launch first, so
 * results do not depend on the setting.  Default: off. */
OHMHIP_EXPERIMENTAL int ohmhip_map_set_async_launch(ohmhip_map_t map, int enable);
/* Same with rays (and optional intensities/timestamps) already resident in device memory.  The arrays must be COMPLETE
 * when the call is made (not merely enqueued on some stream): the map reads them on streams of its own.  Calls below
 * the coalescing threshold are collected like small host batches (round 3): their arrays are copied device to device
 * behind the rays already waiting and run as one batch once min_rays have accumulated or anything observes the map.
 * With `integrated` non-NULL the call waits for that copy (it reports its own count from a filter pass over the copy),
 * so the caller's arrays are free when it returns; with NULL they must stay valid until the next ohmhip_map_sync.
 * A call at or above the threshold is a device batch of its own and returns with that batch IN FLIGHT (its kernels read
 * the arrays until it ends).  At most two batches are in flight: when any integrate call returns, every batch but the
 * two launched last has completed (the map double-buffers its per-batch scratch and the set-up pass of a batch waits for
 * the one before the previous).  So the arrays of a batch may be reused once two further batches have been LAUNCHED and
 * the call that launched the second has returned -- three buffers used in turn, advanced per launch
 * (ohmhip_map_batches_launched; a call that launches nothing must not advance the turn), never need a sync
 * (ohm_amd/distributed.py, PartitionedIntegrator) -- or after ohmhip_map_sync. */
int ohmhip_map_integrate_rays_device(ohmhip_map_t map, const double *d_rays, size_t element_count,
                                     const float *d_intensities, const double *d_timestamps, unsigned ray_flags,
                                     size_t *integrated);
/* Residency accounting, the counterpart of ohm::GpuCacheStats (ohmgpu/GpuCacheStats.h, GpuLayerCache::queryStats,
 * ohmgpu/GpuLayerCache.h:334-339).  The whole map is resident, so the figures read: hits = regions a batch touched that
 * were resident already, misses = regions a batch (or an upload) created, full = times the pool was exhausted and had to
 * be re-allocated at twice the size (the reference evicts its least recently used region instead,
 * ohmgpu/GpuLayerCache.cpp:530-584).  Cumulative since creation or the last reset. */
typedef struct ohmhip_cache_stats
{
  uint64_t hits;
  uint64_t misses;
  uint64_t full;
  uint32_t regions_resident;
  uint32_t region_capacity;   /* regions the pool holds without growing                                   */
  uint64_t bytes_per_region;  /* all enabled layers + per-region scratch                                    */
  uint64_t memory_limit;      /* see ohmhip_map_set_memory_limit (0 = device memory is the limit)           */
  uint64_t evictions;         /* regions moved to the host store (ohmhip_map_set_spill_to_host)             */
  uint64_t readmissions;      /* regions brought back from it                                               */
  uint32_t regions_spilled;   /* regions in the host store right now                                        */
  uint32_t spill_enabled;
  uint64_t writebacks;        /* regions copied to the store in the background, ahead of their eviction     */
  uint64_t writeback_hits;    /* evictions that found their region clean (no copy-out on the batch's path)   */
  uint64_t writeback_stale;   /* background copies discarded because a batch touched the region afterwards  */
} ohmhip_cache_stats;
int ohmhip_map_cache_stats(ohmhip_map_t map, ohmhip_cache_stats *stats, int reset);
/* RESIDENCY LIMIT.  Without spilling (below) a map that outgrows what it may allocate fails the batch that needs the
 * extra regions with OHMHIP_ERR_CAPACITY and stays exactly as it was before that batch (the batch's region inserts are
 * rolled back; this holds for pool / chunk-list EXHAUSTION, the failure a caller can act on -- a batch that dies later
 * of a device error or a failed buffer allocation may leave regions it touched flagged as modified, which costs a
 * redundant copy at the next syncVoxels(), never a wrong value), so the caller can cull regions (ohmhip_map_remove_regions after reading them back) and present the
 * batch again -- or turn on ohmhip_map_set_spill_to_host and let the library move cold regions to host memory.  The
 * limit is free device memory -- 288 GB of HBM3E hold about 1 million 32^3 occupancy-only regions (8.1 B per voxel
 * with scratch) -- or, when set, `bytes` for this map's region pool (the reference's gpu_mem_size,
 * ohmgpu/GpuCache.h:90, bounds its cache the same way).  0 removes the limit. */
int ohmhip_map_set_memory_limit(ohmhip_map_t map, uint64_t bytes);
/* SPILL TO HOST (off by default).  With it on, a batch that needs more regions than the memory limit (or the device)
 * allows no longer fails: resident regions -- a quarter of the pool at a time; those not expected back soon: least
 * recently used first, but a region that has been coming back every N batches is kept while its next use is near (a
 * sweep over a map larger than the pool would otherwise lose every region once per revolution); the counterpart of
 * the reference's LRU slot reuse, ohmgpu/GpuLayerCache.cpp:530-584 -- are copied to a host store inside the
 * library and dropped from the pool, and the batch is repeated.  A stored region stays part of the map: it is listed by
 * ohmhip_map_regions / _region_count / _dirty_regions, ohmhip_map_read_regions serves it from the store, and it returns
 * to the pool with its content when a later batch reaches it or an upload / ohmhip_map_ensure_regions names it.  Results
 * are those of an unbounded pool.  WRITE-BACK: once the pool is under pressure the regions the policy would evict next
 * are copied to the store in the background, on the copy stream while batches run, so that an eviction finds them
 * clean and only drops them (the reference overlaps the download of the cache slot it reuses with queued work the same
 * way, ohmgpu/GpuLayerCache.cpp:550-584); a copy is discarded if a batch touches its region afterwards
 * (ohmhip_cache_stats::writebacks / writeback_hits / writeback_stale).  What does not combine with it: replica merge (OHMHIP_ERR_UNSUPPORTED either way
 * round) and the zero-copy views (ohmhip_map_region_slot reports OHMHIP_ERR_NOT_FOUND for a stored region).  A batch
 * that alone touches more regions than the limit allows is integrated as two halves in ray order (and those again, down
 * to single rays: the reference finalises what it has enqueued when its cache fills in the middle of a batch and carries
 * on, ohmgpu/GpuMap.cpp:900-996; results are those of the whole batch -- except on maps with a traversal layer, whose
 * exit range is carried within a call: there the call fails with OHMHIP_ERR_CAPACITY and changes nothing.  When a LATER
 * half -- in the end a single ray -- still does not fit, the call returns OHMHIP_ERR_CAPACITY with the halves before it
 * applied: `*integrated` then counts the leading elements that were integrated and must not be presented again; the
 * host mirrors put that count into the exception they raise).  Only this cause splits a batch: a full hash, a refused
 * allocation or the end of the slot field fail at once and change nothing.  Turning spilling on sets the batch coalescing threshold to 0 (a collected
 * batch touches the regions of all its calls at once). */
int ohmhip_map_set_spill_to_host(ohmhip_map_t map, int enable);
/* The background write-back of the spill path (see WRITE-BACK above), opt-in: off by default. */
OHMHIP_EXPERIMENTAL int ohmhip_map_set_spill_writeback(ohmhip_map_t map, int enable);
/* Wait for all queued work (GpuMap::syncVoxels fence half, ohmgpu/GpuMap.cpp:308-324). */
int ohmhip_map_sync(ohmhip_map_t map);
int ohmhip_map_last_stats(ohmhip_map_t map, ohmhip_batch_stats *stats);

/* Device phase times of one of the last 32 batches (batches_back = 0: the latest): ms[0] the device time the batch cost
 * -- first kernel start -> last kernel end, or, for batches that overlapped (this batch's plan had ended before the
 * previous batch's last kernel did), the interval between the previous batch's last kernel and this one's (a batch's
 * set-up pass runs on a second stream under the previous batch's last kernels); a batch presented after the device went
 * idle keeps its own span, host idle time between batches is never counted --, ms[1] ray setup + binning (the set-up pass is timed from the moment it may start: queued behind a running
 * walk kernel it mostly waits for CUs), ms[2] the region walk kernel, ms[3] sample ordering + ordered apply.
 * hipEvents on the map's streams (the gputil::Event / Queue::mark() bookkeeping of ohmgpu/GpuMap.cpp:1036-1191 serves
 * the same purpose); waits for that batch only.  Lets a caller time a run of batches without synchronising after each. */
OHMHIP_EXPERIMENTAL int ohmhip_map_batch_timings(ohmhip_map_t map, uint32_t batches_back, float ms[4]);
/* What the phase times are read from (round 5).  The end of a batch's binning, sample ordering, walk and apply phases and
 * of its plan are the STOP EVENTS of the kernels themselves (hipExtLaunchKernelGGL: bound to the kernel's completion
 * signal, free), so ms[0] (as the interval between consecutive batches' ends; for a batch on its own: plan end -> batch
 * end), ms[2] (end of the kernel before the walk -> end of the walk kernel, on the stream it runs on) and ms[3] are
 * always available.  The START of the set-up pass and of the binning pass have no kernel in front of them to carry an
 * event: they are hipEventRecord markers, and a marker idles the queue for 3-7 us before the next kernel
 * (scripts/probes/event_probe.hip) -- round 4 paid eight of them per batch, 5 % of a C1 batch.  They are therefore
 * recorded only with phase timing on (this call, OHMHIP_PHASE_TIMING=1): ms[1], and ms[0] of a batch on its own as first
 * kernel start -> last kernel end, need it and read 0 / the shorter span otherwise.  Default: off. */
OHMHIP_EXPERIMENTAL int ohmhip_map_set_phase_timing(ohmhip_map_t map, int enable);
/* The per-batch device buffers (ray set-up records, ray-region segments, sample keys, sort scratch) are grown by the
 * batch that first needs them -- a few hipMalloc calls, each a device synchronisation, inside that call.  The reference's
 * GpuMap constructor sizes its key / ray buffers for `expected_element_count` up front (ohmgpu/GpuMap.cpp:429-470); this is
 * the counterpart: size everything a batch of `ray_count` rays needs now (the deferred-event list: 16 per ray, at most
 * 2^27 events).  Optional; batches of any size still work, and the host mirrors treat a failing reservation as "grow on
 * demand", not as an error. */
OHMHIP_EXPERIMENTAL int ohmhip_map_reserve_rays(ohmhip_map_t map, size_t ray_count);
/* OccupancyMap::firstRayTime / setFirstRayTime (ohm/OccupancyMap.h:342-351): the time base the touch-time layer is
 * encoded against (milliseconds since it, ohm/VoxelTouchTimeCompute.h:24-37).  Like the reference the map takes it from
 * the first time stamp it is ever given; set it explicitly where that is not the map's to decide -- the ranks of a
 * partitioned map must share ONE base (the first stamp of the whole job), not each the first stamp that happens to be
 * routed to it.  A negative value means "not set yet". */
int ohmhip_map_set_first_ray_time(ohmhip_map_t map, double time);
int ohmhip_map_first_ray_time(ohmhip_map_t map, double *time);
/* Device batches the map has launched since it was created.  An integrate call that only collects its rays (batch
 * coalescing) or is rejected launches none: callers that recycle device ray buffers by the "two batches in flight" rule
 * of ohmhip_map_integrate_rays_device count LAUNCHES with this, not calls -- a buffer handed to the call that made the
 * count L is free once the count has reached L + 2 and a later integrate call has returned (or after ohmhip_map_sync).
 * Does not flush collected rays. */
OHMHIP_EXPERIMENTAL int ohmhip_map_batches_launched(ohmhip_map_t map, uint64_t *count);
/* Maps whose regions are cut into tiles (LARGE REGIONS above): rays, among those that passed the filter, with an end in
 * a region the reference still addresses (|region| <= 32767) but whose tile coordinates leave the key's 16-bit fields.
 * Such a ray is not walked and, when it is the sample that lies out there, its sample is dropped -- the one place results
 * differ from the CPU mappers run with the same region size; this count (since the map was created, rejected batches
 * excluded) lets a caller see that it happened.  Always 0 for regions of up to 32768 voxels.  Flushes collected rays. */
OHMHIP_EXPERIMENTAL int ohmhip_map_rays_beyond_tiles(ohmhip_map_t map, uint64_t *count);

/* Region table (replaces GpuLayerCache::lookup, ohmgpu/GpuLayerCache.cpp:104-119). keys = int16 xyz triples. */
int ohmhip_map_region_count(ohmhip_map_t map, size_t *count);
int ohmhip_map_regions(ohmhip_map_t map, int16_t *keys_xyz, size_t capacity, size_t *count);
/* Regions modified on device since the last ohmhip_map_clear_dirty (dirty-region tracking for syncVoxels). */
int ohmhip_map_dirty_regions(ohmhip_map_t map, int16_t *keys_xyz, size_t capacity, size_t *count);
int ohmhip_map_clear_dirty(ohmhip_map_t map);
size_t ohmhip_layer_voxel_bytes(int layer_id);

/* GpuLayerCache::syncToMainMemory (ohmgpu/GpuLayerCache.cpp:300-321, 670-696): copy `count` regions' layer blocks
 * into dsts[i] (each region_voxels * voxel_bytes, MapChunk layout x + y*dx + z*dx*dy, ohm/MapChunk.h:33-50).
 * Pinned staging + hipMemcpyAsync on the copy stream. */
int ohmhip_map_read_regions(ohmhip_map_t map, int layer_id, const int16_t *keys_xyz, size_t count, void *const *dsts);
/* GpuLayerCache::upload (ohmgpu/GpuLayerCache.cpp:172-182): make CPU-side voxel blocks resident (creates regions). */
int ohmhip_map_write_regions(ohmhip_map_t map, int layer_id, const int16_t *keys_xyz, size_t count,
                             const void *const *srcs);
/* GpuCache::clear (ohmgpu/GpuCache.cpp, ohm/MapRegionCache.h): drop all regions. */
int ohmhip_map_clear(ohmhip_map_t map);
/* MapRegionCache::remove as the core map calls it when it drops regions (OccupancyMap::cullRegions ->
 * gpu_cache->remove, ohm/OccupancyMap.cpp:1202-1234; GpuLayerCache::remove, ohmgpu/GpuLayerCache.cpp): the listed
 * regions leave the device map (their voxels are discarded, modified or not -- sync first to keep them); unknown keys
 * are ignored.  *removed = regions actually dropped. */
int ohmhip_map_remove_regions(ohmhip_map_t map, const int16_t *keys_xyz, size_t count, size_t *removed);

/* LineKeysQueryGpu / `calculateLines` (ohmgpu/LineKeysQueryGpu.cpp, ohmgpu/gpu/LineKeys.cl:66-100): the voxel keys on
 * `line_count` query lines (6 doubles each: start, end), in walk order, with the CPU walk's fp64 semantics.  keys_out is
 * line_count * max_keys_per_line records in the reference's GpuKey layout (short region[3]; uchar voxel[4], 10 bytes,
 * ohmgpu/GpuKey.h:37-46); counts_out[i] is the line's total voxel count.  Host pointers; synchronous. */
int ohmhip_map_line_keys(ohmhip_map_t map, const double *lines, size_t line_count, uint32_t max_keys_per_line,
                         void *keys_out, uint32_t *counts_out);

/* GpuTransformSamples::transform (ohmgpu/GpuTransformSamples.h:75-79, .cpp:97-210; kernel transformTimestampedPoints,
 * ohmgpu/gpu/TransformSamples.cl:94-228): sensor-frame samples with time stamps + a timestamped trajectory (translations
 * xyz, rotations as quaternions x,y,z,w) -> world-frame ray pairs (sensor origin, sample), 6 doubles per valid sample,
 * compacted in input order into `output` (resized), ready for ohmhip_map_integrate_rays_device().  Same rules as the
 * reference: samples with a NaN component or dot(s, s) > max_range are skipped; pose = lerp of the bracketing
 * translations and rot[from] * slerp(rot[from], rot[to], f); fp64 throughout (the reference kernel is fp32).  Host
 * pointers in; *ray_elements = 2 x valid samples (the reference's return value).  Synchronous on `stream` (NULL: the
 * default stream). */
int ohmhip_transform_samples(const double *transform_times, const double *transform_translations,
                             const double *transform_rotations_xyzw, uint32_t transform_count,
                             const double *sample_times, const double *local_samples, uint32_t point_count,
                             double max_range, ohmhip_stream_t stream, ohmhip_buffer_t output, uint32_t *ray_elements);

/* Multi-GPU merge support (no reference equivalent -- ohm is single device).  The resident layer of a
 * map is one allocation of region_stride_bytes per slot: ohm_amd/distributed.py wraps it as a device tensor and runs
 * the RCCL all-reduce of touched-region occupancy deltas on it.  ensure_regions makes regions resident (cleared)
 * without touching their data and returns their slots; mark_dirty flags slots for the next syncVoxels(). */
int ohmhip_map_device_layer_ptr(ohmhip_map_t map, int layer_id, void **device_ptr, size_t *region_stride_bytes);
int ohmhip_map_region_slot(ohmhip_map_t map, const int16_t key_xyz[3], uint32_t *slot);
int ohmhip_map_ensure_regions(ohmhip_map_t map, const int16_t *keys_xyz, size_t count, uint32_t *slots);
int ohmhip_map_mark_dirty(ohmhip_map_t map, const uint32_t *slots, size_t count);

/* Exact multi-GPU integration, "owner computes" (SURVEY 8e mode 2: map partitioned by region; no reference equivalent).
 * With world_size > 1 the map integrates only what falls in regions ohmhip_region_owner() assigns to `rank`: the ray
 * segments crossing those regions and the samples landing in them.  Every voxel's update sequence depends only on the
 * rays that reach it, in order, so when each of world_size maps is given the SAME ray stream the union of their
 * regions is bit-identical to one map integrating that stream (tests/test_gpu_owner_computes.py); what the ranks
 * exchange is rays (an all-gather), never voxels.  Regions are dealt to ranks by a hash of their block of
 * 2^block_shift regions per axis.  Set before the first integrate call; world_size <= 1 turns the filter off.
 * ohmhip_batch_stats::visits keeps counting the voxels of the presented rays, owned or not. */
int ohmhip_map_set_region_ownership(ohmhip_map_t map, uint32_t world_size, uint32_t rank, int block_shift);
int ohmhip_region_owner(const int16_t *keys_xyz, size_t count, int block_shift, uint32_t world_size,
                        uint32_t *owners);

/* ------------------------------------------------------------------------------------------------------------------
 * Partitioned map (north_star: "ray batches shard by sensor-origin region across GPUs"; no reference
 * equivalent -- ohm is single device).  The exact multi-GPU mode, and what `bench.py --gpus N` runs: every rank owns a
 * TERRITORY of region blocks and holds only those regions; rays travel to the owners of the regions they cross (48 B
 * per routed ray over RCCL -- voxels never cross a link), and every rank integrates the rays addressed to it in
 * (source rank, ray) order.  The union of the ranks' regions is bit-identical to one map integrating rank 0's batch,
 * then rank 1's, ... -- for every map type, clamps included (tests/test_gpu_partitioned.py).  One RayFlag is refused on
 * a partitioned map (OHMHIP_ERR_UNSUPPORTED): OHMHIP_RF_STOP_ON_FIRST_OCCUPIED, whose effect on a voxel depends on voxels
 * other ranks own.
 *
 * ohmhip_map_set_region_partition: like ohmhip_map_set_region_ownership, but the blocks of 2^block_shift regions per
 * axis are dealt by a TABLE (x fastest) over [grid_origin, grid_origin + grid_dims) in block coordinates (region
 * coordinate >> block_shift); blocks outside the grid belong to the owner of the nearest cell, so territories that reach
 * the table's edge extend outwards without limit.  grid_dims = {0,0,0} / owners = NULL: the block hash of
 * ohmhip_region_owner.  The table is copied.  Only on an empty map.  world_size <= 64.
 * ohmhip_map_region_owners: owners under the map's current partition (host evaluation).
 * ohmhip_map_route_rays: for `ray_count` rays in DEVICE memory (6 doubles each) find, per ray, the ranks owning a
 * region its walk touches or its sample lands in -- exactly, with the functions the integration itself runs (rays the
 * map's ray filter rejects go nowhere) -- and compact the rays per destination, in ray order, into d_routed
 * (destination d's block starts at counts[0] + .. + counts[d-1]); d_routed_index (may be NULL) receives each routed
 * ray's index in the input, for callers that route side arrays (time stamps, intensities) themselves.  counts (host,
 * world_size entries) = rays per destination; *visits (may be NULL) = voxel visits of the input rays.  Returns
 * OHMHIP_ERR_CAPACITY -- with counts valid -- when d_routed holds fewer than sum(counts) rays: grow and repeat.
 * Synchronous, on a stream of its own: it reads nothing a batch writes, so it runs beside the batches in flight (the
 * next batch is routed and exchanged while the previous one integrates).
 * ohmhip_comm_exchange_counts / _rays: the all-to-all of the routed rays over RCCL (both collective).  First the counts
 * (recv_counts[s] = rays rank s addressed to this rank), then, with d_recv holding sum(recv_counts) rays, the blocks:
 * on return (stream order) d_recv holds the rays addressed to this rank in source-rank order, ready for
 * ohmhip_map_integrate_rays_device.  A torch.distributed all_to_all_single does the same job for Python hosts
 * (ohm_amd/distributed.py: PartitionedIntegrator). */
typedef struct ohmhip_partition
{
  uint32_t world_size;
  uint32_t rank;
  int32_t block_shift;
  int32_t grid_origin[3];
  uint32_t grid_dims[3];
  const uint8_t *owners;
} ohmhip_partition;
int ohmhip_map_set_region_partition(ohmhip_map_t map, const ohmhip_partition *partition);
int ohmhip_map_region_owners(ohmhip_map_t map, const int16_t *keys_xyz, size_t count, uint32_t *owners);
/* The same rule for a partition description alone (no map, no device): what hosts use to plan and check a table. */
int ohmhip_partition_owners(const ohmhip_partition *partition, const int16_t *keys_xyz, size_t count, uint32_t *owners);
int ohmhip_map_route_rays(ohmhip_map_t map, const double *d_rays, size_t ray_count, unsigned ray_flags, double *d_routed,
                          uint32_t *d_routed_index, size_t capacity, uint32_t *counts, uint64_t *visits);

/* ------------------------------------------------------------------------------------------------------------------
 * Replica merge (SURVEY 8e mode 1: every GPU integrates the rays of its own sensor origins into its own resident map;
 * regions touched by more than one GPU are reconciled on demand).  No reference equivalent -- ohm is single device.
 *
 * Rule, per voxel of the occupancy layer, relative to the state `base` all replicas shared after the previous merge:
 *     merged = clamp(base + sum over ranks of (value_r - base), min, max)
 * (+inf == unobserved counts as 0 and stays unobserved only if no rank observed the voxel).  That equals integrating the
 * ranks' rays one rank after the other wherever no min / max clamp engaged in between -- log-odds updates commute until
 * they saturate -- and is the usual order-free map-merge value where one did.  Exact multi-GPU integration is the
 * region-ownership mode above.  Other layers are not additive and stay per replica.
 *
 * ohmhip_map_enable_merge() gives the map a base copy of its occupancy layer (current content == base) and starts
 * tracking the regions modified since.  INVARIANT: `base` of a region is the same on every rank (a rank that does not
 * hold the region == base unobserved).  It changes only when the region is EXCHANGED, and then on every rank (a rank
 * that never touched the region creates it and receives the merged tile).  A region that only one rank modified is
 * not exchanged in OHMHIP_MERGE_SHARED_ONLY mode (the default): it stays PENDING on that rank -- listed by
 * merge_keys in every later round, its base untouched -- until a second rank lists it too; then the owner's whole
 * pending delta travels.  (Round 2 rebased such regions locally, which made `base` rank-private and later exchanges
 * wrong on the peers.)  So after a merge: an exchanged region is bit-identical on all ranks; a pending region is
 * exact on its one rank and absent / at its last exchanged value elsewhere.  OHMHIP_MERGE_FULL_UNION exchanges every
 * pending region of any rank: all replicas are then bit-identical maps after every merge, at the price of moving
 * the tiles only one rank needed.
 *
 * The merge itself is either one call over RCCL (ohmhip_map_merge_replicas: region key lists are all-gathered, the
 * delta tiles of the agreed region list are all-reduced -- 5 bytes per voxel: float delta summed + observer flag by
 * max --, everything on the map's stream; the ranks agree on the outcome of their local steps with a one-word
 * all-reduce BEFORE the tile all-reduce, so a rank-local failure makes every rank return together: the failing rank its
 * own error, the others OHMHIP_ERR_PEER, nothing applied, all regions still pending) or, for any other transport, the
 * steps it is made of: ohmhip_map_merge_keys -> (exchange keys, agree on the ordered region list) ->
 * ohmhip_map_merge_pack -> (sum the deltas / max the observer flags across ranks) -> ohmhip_map_merge_apply. */
#define OHMHIP_MERGE_SHARED_ONLY 0
#define OHMHIP_MERGE_FULL_UNION 1
typedef struct ohmhip_comm_s *ohmhip_comm_t;
#define OHMHIP_COMM_ID_BYTES 128
int ohmhip_comm_unique_id(unsigned char id[OHMHIP_COMM_ID_BYTES]);  /* ncclGetUniqueId; hand it to every rank */
/* ncclCommInitRank on the calling thread's current device (collective: every rank calls it with the same id). */
int ohmhip_comm_init_rank(ohmhip_comm_t *comm, const unsigned char id[OHMHIP_COMM_ID_BYTES], int world_size, int rank);
int ohmhip_comm_destroy(ohmhip_comm_t comm);
/* Partitioned map: the all-to-all of routed rays (see "Partitioned map" above). */
/* FAILURE AGREEMENT: a rank that cannot take part in the step (its routing failed, it is out of memory ...) still calls
 * ohmhip_comm_exchange_counts, with OHMHIP_COUNT_FAILED in every send count: the call then returns OHMHIP_ERR_PEER on
 * EVERY rank (recv_counts zeroed) and no rank enters the payload exchange -- all ranks leave the step together.  A failure
 * after the exchange (the integration of what arrived) is local: the failing rank's caller has to end the group's run. */
#define OHMHIP_COUNT_FAILED 0xffffffffu
int ohmhip_comm_exchange_counts(ohmhip_comm_t comm, const uint32_t *send_counts, uint32_t *recv_counts,
                                ohmhip_stream_t stream);
int ohmhip_comm_exchange_rays(ohmhip_comm_t comm, const double *d_send, const uint32_t *send_counts, double *d_recv,
                              const uint32_t *recv_counts, ohmhip_stream_t stream);
/* The routed rays' SIDE ARRAYS (round 5) -- the time stamps and intensities the reference passes with every batch
 * (ohmgpu/GpuMap.cpp:416; GpuNdtMap.cpp:433-486: NDT-TM needs the intensities, a touch-time layer the stamps) travel the
 * same way: ohmhip_gather_rows puts an array into routed order with the index list ohmhip_map_route_rays returns
 * (d_dst[i] = d_src[d_index[i]], rows of bytes_per_row bytes, device pointers, stream order), and
 * ohmhip_comm_exchange_side is the all-to-all of ohmhip_comm_exchange_rays for bytes_per_ray bytes per ray with the
 * same counts.  Both exchange calls validate their arguments before the collective starts: an invalid call returns
 * OHMHIP_ERR_INVALID_ARG without having taken part in it. */
int ohmhip_comm_exchange_side(ohmhip_comm_t comm, const void *d_send, const uint32_t *send_counts, void *d_recv,
                              const uint32_t *recv_counts, uint32_t bytes_per_ray, ohmhip_stream_t stream);
int ohmhip_gather_rows(const void *d_src, const uint32_t *d_index, size_t count, uint32_t bytes_per_row, void *d_dst,
                       ohmhip_stream_t stream);

typedef struct ohmhip_merge_stats
{
  uint32_t regions_local;   /* regions pending on this rank (modified since they were last exchanged)  */
  uint32_t regions_union;   /* ... any rank did                                                       */
  uint32_t regions_shared;  /* the ones whose tiles travel (pending on > 1 rank; all in FULL_UNION)    */
  uint64_t payload_bytes;   /* bytes this rank contributed to the tile all-reduce (5 per shared voxel) */
  uint64_t key_bytes;       /* bytes it contributed to the key all-gather                              */
  float ms_total;           /* host wall time of the call                                              */
} ohmhip_merge_stats;

int ohmhip_map_enable_merge(ohmhip_map_t map);
int ohmhip_map_set_merge_mode(ohmhip_map_t map, int mode);
/* Collective over `comm`.  On return every rank holds the merged values of the exchanged regions, which become the new
 * base on every rank. */
int ohmhip_map_merge_replicas(ohmhip_map_t map, ohmhip_comm_t comm, ohmhip_merge_stats *stats);
/* The steps, for other transports.  merge_keys: keys (int16 x 3 each) of the PENDING regions -- modified since they were
 * last exchanged --, at most `capacity` written, *count = how many there are.  merge_pack: for `count` regions in the
 * given order (made resident if they are not) write count x region_voxels float deltas and uint8 observer flags to the
 * DEVICE buffers.  merge_apply: the same regions with the deltas summed / flags max-ed (or summed) over all ranks; EVERY
 * rank applies every exchanged region.  merge_finish: kept for ABI compatibility, nothing left to do. */
int ohmhip_map_merge_keys(ohmhip_map_t map, int16_t *keys_xyz, size_t capacity, size_t *count);
int ohmhip_map_merge_pack(ohmhip_map_t map, const int16_t *keys_xyz, size_t count, float *d_delta,
                          unsigned char *d_observers);
int ohmhip_map_merge_apply(ohmhip_map_t map, const int16_t *keys_xyz, size_t count, const float *d_delta_sum,
                           const unsigned char *d_observer_sum);
int ohmhip_map_merge_finish(ohmhip_map_t map);

#ifdef __cplusplus
}
#endif

#endif /* OHMHIP_H */
