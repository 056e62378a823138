// ohmhip_internal.h -- shared declarations for libohmhip.so (gfx950 only).
#ifndef OHMHIP_INTERNAL_H
#define OHMHIP_INTERNAL_H

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <new>

#include "../../include/ohmhip.h"

#define OHMHIP_CHECK(expr)                \
  do                                      \
  {                                       \
    const int err__ = static_cast<int>(expr); \
    if (err__ != 0)                       \
    {                                     \
      return err__;                       \
    }                                     \
  } while (0)

/// Closes the function-try-block of every entry point: no C++ exception crosses the C ABI (include/ohmhip.h).
#define OHMHIP_ABI_CATCH                \
  catch (const std::bad_alloc &)        \
  {                                     \
    return OHMHIP_ERR_CAPACITY;         \
  }                                     \
  catch (...)                           \
  {                                     \
    return OHMHIP_ERR_INTERNAL;         \
  }

struct ohmhip_stream_s
{
  hipStream_t stream;
};

struct ohmhip_event_s
{
  hipEvent_t event;
  bool recorded;
};

struct ohmhip_buffer_s
{
  void *ptr;
  size_t bytes;
  unsigned flags;
};

namespace ohmhip
{
/// Constant per-map parameters handed to every kernel by value.
struct MapConst
{
  double resolution;
  double region_dim[3];  ///< Spatial size of a region per axis (ohm/OccupancyMap.cpp:200-202).
  double origin[3];
  /// Voxels per TILE per axis: the unit the whole pipeline works in (hash table, pool slots, LDS count tile, segment and
  /// sample keys).  A region of up to 32768 voxels is one tile; a larger one (ohm/OccupancyMap.h:287 allows 255 per axis)
  /// is cut into equal tiles of full x rows -- z slabs, and y strips of single z layers when one layer alone is too
  /// large -- so that a tile's voxels are contiguous in the region's MapChunk block.  Tile coordinates are global voxel
  /// coordinates divided by dim[]; they play the part of region coordinates everywhere but in the key maths.
  int dim[3];
  int region_voxels;     ///< dim[0] * dim[1] * dim[2] (voxels per tile)
  int kdim[3];           ///< Voxels per REGION per axis as the caller configured them: key maths only (voxelKey, voxel centres).
  int tile_split[3];     ///< kdim / dim: tiles per region per axis ({1, 1, 1}: a region is a tile)
  float hit_value;
  float miss_value;
  float threshold_value;
  float min_value;
  float max_value;
  float sat_min;         ///< saturation_min as the CPU mapper derives it (lowest() when disabled).
  float sat_max;
  int filter_mode;
  double filter_range;
  // NDT
  float sensor_noise;
  unsigned sample_threshold;
  float adaptation_rate;
  float reinit_threshold;
  unsigned reinit_count;
  float initial_intensity_cov;
  // TSDF
  float tsdf_max_weight, tsdf_trunc, tsdf_dropoff, tsdf_sparsity;
  // Region ownership (multi-GPU "owner computes"): with owner_world > 1 the map only integrates the
  // ray segments and samples which fall in regions regionOwner() assigns to owner_rank.
  unsigned owner_world;
  unsigned owner_rank;
  int owner_shift;
  /// Region partition table (ohmhip_map_set_region_partition; null: blocks are dealt by regionOwner()'s hash): owner
  /// rank per block of 2^owner_shift regions per axis over the grid [owner_grid_origin, owner_grid_origin +
  /// owner_grid_dims) in block coordinates, x fastest; blocks outside the grid belong to the nearest cell's owner.
  const unsigned char *owner_table;
  int owner_grid_origin[3];
  int owner_grid_dims[3];
  /// Per-ray RayFilterFlag bits of a batch the CALLER filtered (ohmhip_map_integrate_rays_filtered; null otherwise):
  /// the device then applies no filter of its own and takes "end point was clipped" from kRffClippedEnd (bit 2).
  const unsigned char *batch_filter_flags;
  /// Fixed-point walk predictor (see Segment): units per metre of ray parameter, chosen so that a region's diagonal
  /// maps to 2^30 / 1.01, and the lead (in units) the smallest candidate needs over the second smallest to be trusted.
  double fix_scale;
  uint32_t fix_margin;
};

/// Per-ray line-walk parameters: everything ohm/LineWalkCompute.h:260-280 derives once per ray, in fp64, plus the
/// integer extent of the walk.  The walk state at ANY step is a pure function of the per-axis step counts because
/// the reference recomputes `time_next = initial + delta * |stepped|` rather than accumulating it
/// (ohm/LineWalkCompute.h:299-301).  That is what lets a region workgroup resume a ray mid-walk bit-exactly.
struct RayWalk
{
  double init[3];   ///< initial_delta[]: exit time of the start voxel per axis.
  double delta[3];  ///< step_delta[]
  double length;    ///< WalkSteps::length (0 for rays shorter than the 1e-3 epsilon)
  int g0[3];        ///< start voxel in global voxel coordinates (region * dim + local)
  int total[3];     ///< |steps_remaining| at the start = Manhattan extent per axis
  unsigned flags;   ///< kRw* bits
  unsigned pad;
};

enum : unsigned
{
  kRwValid = 1u << 0,
  kRwSign0 = 1u << 1,  ///< sign[a] (1 => negative step direction) in bits 1..3
  kRwIncludeEnd = 1u << 4,    ///< end voxel is visited as part of the ray (clipped end / kRfEndPointAsFree)
  kRwApplySample = 1u << 5,   ///< sample voxel receives the hit update
  kRwExcludeStart = 1u << 6,  ///< kRfExcludeOrigin
  kRwWalk = 1u << 7,          ///< ray part is walked (not kRfExcludeRay)
  kRwPassed = 1u << 8,        ///< the ray passed the ray filter (counted as integrated, like the reference's upload count)
  /// Tiled maps only: an end of the ray lies in a region the reference addresses but whose TILE coordinates leave the
  /// packed key's 16-bit fields (include/ohmhip.h, "LARGE REGIONS"); counted in BatchInfo::n_beyond_tiles.
  kRwBeyondTiles = 1u << 9
};

/// One (ray, region) unit of line-walk work: "visit `count` voxels of ray `ray` starting at voxel `vi` of the region".
/// Computed densely by k_ray_bin so the walk kernel's lane refill is two 16-byte loads and a few unpacking
/// instructions.
///
/// The record carries a FIXED-POINT PREDICTOR of the walk, not the walk state itself: f[a] is the time of the next step
/// along axis a relative to the time the ray entered the region (the segment's first step for the ray's first segment),
/// e[a] the step delta, both in units of 1 / MapConst::fix_scale metres of ray parameter, truncated.  The walk kernel
/// advances the predictor with integer adds and takes its choice of axis whenever the smallest candidate leads the
/// second smallest by more than MapConst::fix_margin units (the accumulated truncation error is at most 1 + steps per
/// candidate); otherwise -- ties included -- the lane recomputes the exact fp64 time_next values from the ray's RayWalk
/// record and the step counts implied by its position, and decides exactly as ohm/LineWalkCompute.h:282-301 does.  The
/// voxel sequence is therefore bit-identical to the CPU walk and the common step needs no fp64 arithmetic.
///   * values are clamped LOW only: f >= 2^31 is stored as kFixFar, e >= 2^30 as kFixMaxDelta; the kernel never trusts
///     a candidate at or beyond kFixMaxDelta (every step a segment needs happens within the region's diagonal, which
///     fix_scale maps to 2^30 / 1.01)
///   * an axis with no steps left in the whole ray has f = kFixFar, e = 0 (time_next = inf in the reference)
///   * an axis that runs out of steps INSIDE the segment keeps counting (its phantom steps lie beyond the ray's last
///     real step; rays for which that cannot be shown in fp64 are poisoned)
///   * a poisoned record (f = e = 0: negative / non-finite times, degenerate rays) is never certain: every step of it
///     is decided exactly
/// Packing: vox = first voxel index (15 bits) | voxel count << 15 (16 bits); e[a] bit 31 = the ray steps towards
/// negative coordinates on axis a; ray = ray index (29 bits, as in the sample sort key) | kSegSkipFirst | kSegEnd.
struct Segment
{
  uint32_t f[3];
  uint32_t vox;
  uint32_t e[3];
  uint32_t ray;
};
static_assert(sizeof(Segment) == 32, "segment records are loaded as two 16-byte words");
constexpr uint32_t kSegRayMask = (1u << 29) - 1u;
constexpr uint32_t kSegSkipFirst = 1u << 29;  ///< the segment's first voxel is the ray's origin voxel and kRfExcludeOrigin is set
constexpr uint32_t kSegEnd = 1u << 30;        ///< the segment's last voxel is the ray's end voxel, visited as part of the ray
constexpr uint32_t kSegVoxelBits = 15;
constexpr uint32_t kSegNegative = 0x80000000u;
constexpr uint32_t kFixFar = 0x80000000u;
constexpr uint32_t kFixMaxDelta = 0x40000000u;

struct Chunk
{
  uint32_t slot;
  uint32_t seg_begin;
  uint32_t seg_end;
  uint32_t hash_index;
};

/// Device-side batch summary read back by the host after the binning pass.
struct BatchInfo
{
  unsigned long long visits;
  unsigned long long rays_ok;
  uint32_t n_touched;
  uint32_t n_chunks;
  uint32_t n_segments;
  uint32_t n_slots;
  uint32_t error;
  uint32_t n_hits;
  uint32_t max_region_hits;  ///< most samples any one region receives in the batch
  uint32_t n_hit_regions;    ///< regions receiving samples (length of the sort list)
  /// Occupancy maps whose walk kernel applies single-chunk regions itself (counts and samples): how many regions are
  /// left for the apply kernels -- regions cut into several chunks (their counts), and regions with samples that are cut
  /// into several chunks, have no chunk at all, or hold more samples than the walk stages (their samples).
  uint32_t n_apply_counts;
  uint32_t n_apply_hits;
  uint32_t n_beyond_tiles;  ///< rays with kRwBeyondTiles (tiled maps: dropped or cut short for key-range reasons)
};

#ifdef OHMHIP_MAX_CHUNK_SEGMENTS
constexpr uint32_t kChunkSegments = OHMHIP_MAX_CHUNK_SEGMENTS;
#else
constexpr uint32_t kChunkSegments = 8192;
#endif
constexpr uint64_t kKeyOccupied = 1ull << 63;
constexpr uint32_t kSlotUnassigned = 0xffffffffu;

// Hit sort key: [slot:20][voxel:15][ray:29]
constexpr int kHitRayBits = 29;
constexpr int kHitVoxelBits = 15;
constexpr int kHitSlotShift = kHitRayBits + kHitVoxelBits;
constexpr uint64_t kHitInvalid = ~0ull;

__host__ __device__ inline uint64_t packRegionKey(int rx, int ry, int rz)
{
  return kKeyOccupied | uint64_t(uint16_t(rx)) | (uint64_t(uint16_t(ry)) << 16) | (uint64_t(uint16_t(rz)) << 32);
}

__host__ __device__ inline void unpackRegionKey(uint64_t key, int16_t out[3])
{
  out[0] = int16_t(uint16_t(key & 0xffffu));
  out[1] = int16_t(uint16_t((key >> 16) & 0xffffu));
  out[2] = int16_t(uint16_t((key >> 32) & 0xffffu));
}

/// Owner of region (rx, ry, rz) among `world` region-partitioned replicas: a hash of the region's block of
/// 2^shift regions per axis (blocks keep a rank's regions spatially clustered), the same on host and device.
__host__ __device__ inline uint32_t regionOwner(int rx, int ry, int rz, int shift, uint32_t world)
{
  uint32_t h = uint32_t(rx >> shift) * 0x9E3779B1u;
  h = (h ^ uint32_t(ry >> shift)) * 0x85EBCA77u;
  h = (h ^ uint32_t(rz >> shift)) * 0xC2B2AE3Du;
  h ^= h >> 15;
  h *= 0x27D4EB2Fu;
  h ^= h >> 13;
  return h % world;
}

/// Cell of a partition table that region coordinate `r` falls in along one axis (clamped: the table's outer cells
/// extend outwards without limit, so a territory that reaches the table's edge owns everything beyond it).
__host__ __device__ inline int partitionCell(int r, int shift, int grid_origin, int grid_dim)
{
  const int c = (r >> shift) - grid_origin;
  return (c < 0) ? 0 : ((c >= grid_dim) ? grid_dim - 1 : c);
}

/// Owner of region (rx, ry, rz) under a partition: the table when there is one (`table` must be addressable by the
/// caller: the device copy in kernels, the host copy on the host), the block hash otherwise.
__host__ __device__ inline uint32_t partitionOwner(const unsigned char *table, const int grid_origin[3],
                                                   const int grid_dims[3], int shift, uint32_t world, int rx, int ry,
                                                   int rz)
{
  if (table)
  {
    const int cx = partitionCell(rx, shift, grid_origin[0], grid_dims[0]);
    const int cy = partitionCell(ry, shift, grid_origin[1], grid_dims[1]);
    const int cz = partitionCell(rz, shift, grid_origin[2], grid_dims[2]);
    return table[(size_t(cz) * size_t(grid_dims[1]) + size_t(cy)) * size_t(grid_dims[0]) + size_t(cx)];
  }
  return regionOwner(rx, ry, rz, shift, world);
}

/// floor(a / b) for b > 0
__host__ __device__ inline int floorDiv(int a, int b)
{
  const int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

/// Owner of the region a TILE belongs to under the map's partition (device code: MapConst::owner_table is a device
/// pointer).  Ownership is a property of the caller's regions: all tiles of a region share their owner.
__device__ inline uint32_t regionOwnerOf(const MapConst &mc, uint64_t key)
{
  int16_t r[3];
  unpackRegionKey(key, r);
  return partitionOwner(mc.owner_table, mc.owner_grid_origin, mc.owner_grid_dims, mc.owner_shift, mc.owner_world,
                        floorDiv(r[0], mc.tile_split[0]), floorDiv(r[1], mc.tile_split[1]),
                        floorDiv(r[2], mc.tile_split[2]));
}

__device__ inline bool ownsRegion(const MapConst &mc, uint64_t key)
{
  if (mc.owner_world <= 1u)
  {
    return true;
  }
  return regionOwnerOf(mc, key) == mc.owner_rank;
}

__host__ __device__ inline uint32_t hashRegionKey(uint64_t key, uint32_t mask)
{
  uint64_t h = key * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return uint32_t(h) & mask;
}
}  // namespace ohmhip

#endif  // OHMHIP_INTERNAL_H
