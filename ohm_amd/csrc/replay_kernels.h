// replay_kernels.h -- ordered per-voxel event replay for the NDT and TSDF mappers (gfx950).
//
// Voxels whose update depends on more than a miss COUNT (voxels which hold / receive samples for NDT; voxels near a
// surface for TSDF) get every event of the batch as a 64-bit key [slot:20][voxel:15][ray:28][is_sample:1].  The keys
// (sample keys from k_ray_bin + deferred visit keys from k_region_walk) are radix sorted, which groups them per voxel
// in ray order -- exactly the order in which the single-threaded CPU mapper applies them -- and one lane per voxel
// group replays the group sequentially with the CPU mapper's arithmetic.  Everything else in those layers is updated
// from integer visit counts (k_apply_counts*), which are order independent.
#ifndef OHMHIP_REPLAY_KERNELS_H
#define OHMHIP_REPLAY_KERNELS_H

#include "ndt_tsdf_device.h"
#include "occupancy_kernels.h"

namespace ohmhip
{
constexpr int kEvRayShift = 1;
constexpr unsigned long long kEvRayMask = (1ull << kHitRayBits) - 1ull;

__device__ inline void voxelCentreOf(const MapConst &mc, const RegionTable &rt, uint32_t slot, uint32_t vi,
                                     double centre[3])
{
  int16_t rk[3];
  unpackRegionKey(rt.slot_keys[slot], rk);
  const int lx = int(vi % uint32_t(mc.dim[0]));
  const int ly = int((vi / uint32_t(mc.dim[0])) % uint32_t(mc.dim[1]));
  const int lz = int(vi / uint32_t(mc.dim[0] * mc.dim[1]));
  centre[0] = globalVoxelCentreAxis(mc, 0, int(rk[0]) * mc.dim[0] + lx);
  centre[1] = globalVoxelCentreAxis(mc, 1, int(rk[1]) * mc.dim[1] + ly);
  centre[2] = globalVoxelCentreAxis(mc, 2, int(rk[2]) * mc.dim[2] + lz);
}

/// Compact the indices of the first event of every voxel group: the replay kernels then run one lane per VOXEL with full
/// waves (launching one lane per event and letting the non-heads exit leaves a wave with one or two live lanes that
/// each loop over a whole group).  Wave-aggregated append; the order of the list does not matter.
constexpr uint32_t kHeadsPerBlock = 2048;  ///< events scanned by one k_group_heads workgroup (256 threads x 8)

/// Round 6: the NDT / TSDF event sort is launched on the PREVIOUS batch's event count (plus head room) instead of waiting
/// for this batch's to reach the host -- 26 us of idle device per C2 batch.  This kernel, behind the walk, pads the event
/// list from the true count up to the speculated one with kHitInvalid keys (they sort to the end and every consumer skips
/// them) and sends the true count to pinned host memory; the kernels that consume the sorted list take the pair
/// (event_count, event_limit) and do NOTHING when the true count exceeded the speculation -- the host, which has read the
/// count by then, repeats the sort and the replay with the exact size (batch_run.h: settleSpeculatedEvents).
__global__ void __launch_bounds__(256)
  k_pad_events(unsigned long long *__restrict__ events, const uint32_t *__restrict__ event_count, uint32_t spec_events,
               uint32_t *__restrict__ host_event_count)
{
  const uint32_t n = *event_count;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    *host_event_count = n;
    __threadfence_system();
  }
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = min(n, spec_events) + blockIdx.x * blockDim.x + threadIdx.x; i < spec_events; i += stride)
  {
    events[i] = kHitInvalid;
  }
}

/// True when a speculatively sized event list turned out too short: the caller (a kernel) leaves without side effects.
__device__ inline bool speculationFailed(const uint32_t *__restrict__ event_count, uint32_t event_limit)
{
  return event_count != nullptr && *event_count > event_limit;
}

__global__ void __launch_bounds__(256)
  k_group_heads(const unsigned long long *__restrict__ sorted, uint32_t n_events, uint32_t *__restrict__ heads,
                uint32_t *__restrict__ n_heads, const uint32_t *__restrict__ event_count, uint32_t event_limit)
{
  if (speculationFailed(event_count, event_limit))
  {
    return;
  }
  // Heads are collected in LDS and the workgroup reserves its output range with ONE global atomic (the counter is a
  // single address: one atomic per wave serialises the whole launch on it).
  __shared__ uint32_t s_list[kHeadsPerBlock];
  __shared__ uint32_t s_count;
  __shared__ uint32_t s_base;
  if (threadIdx.x == 0)
  {
    s_count = 0;
  }
  __syncthreads();
  const unsigned lane = __lane_id();
  const uint32_t block_first = blockIdx.x * kHeadsPerBlock;
#pragma unroll
  for (uint32_t j = 0; j < kHeadsPerBlock / 256; ++j)
  {
    const uint32_t i = block_first + j * 256 + threadIdx.x;
    bool head = false;
    if (i < n_events)
    {
      const unsigned long long key = sorted[i];
      head = key != kHitInvalid && (i == 0 || (sorted[i - 1] >> kHitRayBits) != (key >> kHitRayBits));
    }
    const unsigned long long mask = __ballot(head);
    if (mask)
    {
      const int leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (int(lane) == leader)
      {
        base = atomicAdd(&s_count, uint32_t(__popcll(mask)));
      }
      base = __shfl(base, leader);
      if (head)
      {
        s_list[base + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)))] = i;
      }
    }
  }
  __syncthreads();
  const uint32_t count = s_count;
  if (threadIdx.x == 0 && count)
  {
    s_base = atomicAdd(n_heads, count);
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < count; k += 256)
  {
    heads[s_base + k] = s_list[k];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// NDT: RayMapperNdt::integrateRays per-voxel semantics (ohm/RayMapperNdt.cpp:135-230 misses, :262-402 sample).
// ---------------------------------------------------------------------------------------------------------------------
/// The layers an NDT replay reads and writes.
struct NdtLayers
{
  float *occupancy;
  uint32_t *mean;
  float *covariance;
  float *intensity;    ///< NDT-TM only (null otherwise, together with hit_miss)
  uint32_t *hit_miss;
};

/// Replay ONE voxel group -- the events sorted[i .. ) that share sorted[i]'s (slot, voxel), in ray order -- on the
/// voxel's state.
__device__ inline void replayNdtGroup(const MapConst &mc, const RegionTable &rt, const unsigned long long *sorted,
                                      uint32_t i, uint32_t n_events, const double *__restrict__ rays,
                                      const float *__restrict__ intensities, const NdtLayers &ly,
                                      const SecondaryLayers &sec, const RayWalk *__restrict__ walks)
{
  float *occupancy = ly.occupancy;
  uint32_t *mean_layer = ly.mean;
  float *cov_layer = ly.covariance;
  float *intensity_layer = ly.intensity;
  uint32_t *hit_miss_layer = ly.hit_miss;
  const unsigned long long key = sorted[i];
  const unsigned long long group = key >> kHitRayBits;
  const uint32_t slot = uint32_t(key >> kHitSlotShift);
  const uint32_t vi = uint32_t(key >> kHitRayBits) & ((1u << kHitVoxelBits) - 1u);
  const size_t gi = size_t(slot) * size_t(mc.region_voxels) + vi;

  double centre_a[3];
  voxelCentreOf(mc, rt, slot, vi, centre_a);
  const D3 centre = d3(centre_a[0], centre_a[1], centre_a[2]);

  float occ = occupancy[gi];
  uint32_t mcoord = mean_layer[2 * gi];
  uint32_t mcount = mean_layer[2 * gi + 1];
  Cov6 cov;
  cov.c0 = cov_layer[6 * gi + 0];
  cov.c1 = cov_layer[6 * gi + 1];
  cov.c2 = cov_layer[6 * gi + 2];
  cov.c3 = cov_layer[6 * gi + 3];
  cov.c4 = cov_layer[6 * gi + 4];
  cov.c5 = cov_layer[6 * gi + 5];
  float int_mean = 0, int_cov = 0;
  uint32_t hm_hit = 0, hm_miss = 0;
  if (hit_miss_layer)
  {
    int_mean = intensity_layer[2 * gi];
    int_cov = intensity_layer[2 * gi + 1];
    hm_hit = hit_miss_layer[2 * gi];
    hm_miss = hit_miss_layer[2 * gi + 1];
  }

  uint32_t packed_normal = sec.incident ? sec.incident[gi] : 0u;
  float traversal = sec.traversal ? sec.traversal[gi] : 0.0f;
  uint32_t last_sample_ray = 0xffffffffu;
  for (uint32_t j = i; j < n_events; ++j)
  {
    const unsigned long long kj = sorted[j];
    if ((kj >> kHitRayBits) != group)
    {
      break;
    }
    const uint32_t ray = uint32_t((kj & kEvRayMask) >> kEvRayShift);
    const bool is_sample = (kj & 1ull) != 0;
    double start[3], end[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      start[a] = rays[size_t(ray) * 6 + a];
      end[a] = rays[size_t(ray) * 6 + 3 + a];
    }
    bool clipped = false;
    filterRay(mc, start, end, clipped, ray);  // clip filter may move the end point used by the miss maths
    const D3 sensor = d3(start[0], start[1], start[2]);
    const D3 sample = d3(end[0], end[1], end[2]);
    const D3 mean = subVoxelToLocal(mcoord, mc.resolution) + centre;
    const float initial_value = occ;
    if (!is_sample)
    {
      bool is_miss = false;
      const float adjusted = calculateMissNdt(mc, cov, initial_value, is_miss, sensor, sample, mean, mcount);
      hm_miss += is_miss ? 1u : 0u;
      occ = occupancyAdjustDown(mc, initial_value, adjusted);
    }
    else
    {
      float adjusted = initial_value;
      if (hit_miss_layer)
      {
        calculateHitMissUpdateOnHit(mc, cov, adjusted, hm_hit, hm_miss, sensor, sample, mean, mcount);
        calculateIntensityUpdateOnHit(mc, int_mean, int_cov, adjusted, intensities ? intensities[ray] : 0.0f, mcount);
      }
      const bool reset_mean = calculateHitWithCovariance(mc, cov, adjusted, sample, mean, mcount);
      occ = occupancyAdjustUp(mc, initial_value, adjusted);
      mcount = (!reset_mean) ? mcount : 0;
      mcoord = subVoxelUpdateD3(mcoord, mcount, sample - centre, mc.resolution);
      ++mcount;
      last_sample_ray = ray;
      if (sec.traversal)
      {
        // ohm/RayMapperNdt.cpp:367-373
        traversal += float(sqrt(dot(sample - sensor, sample - sensor)) - lastExitRange(walks, ray));
      }
      if (sec.incident)
      {
        // ohm/RayMapperNdt.cpp:381-388: weight = point count before this sample (count already incremented)
        const float dir[3] = { float(sensor.x - sample.x), float(sensor.y - sample.y), float(sensor.z - sample.z) };
        packed_normal = updateIncidentNormal(packed_normal, dir, mcount - 1);
      }
    }
  }
  if (sec.incident)
  {
    sec.incident[gi] = packed_normal;
  }
  if (sec.traversal)
  {
    sec.traversal[gi] = traversal;
  }
  if (sec.touch_time && sec.timestamps && last_sample_ray != 0xffffffffu)
  {
    sec.touch_time[gi] = encodeVoxelTouchTime(sec.time_base, sec.timestamps[last_sample_ray]);
  }

  occupancy[gi] = occ;
  mean_layer[2 * gi] = mcoord;
  mean_layer[2 * gi + 1] = mcount;
  cov_layer[6 * gi + 0] = cov.c0;
  cov_layer[6 * gi + 1] = cov.c1;
  cov_layer[6 * gi + 2] = cov.c2;
  cov_layer[6 * gi + 3] = cov.c3;
  cov_layer[6 * gi + 4] = cov.c4;
  cov_layer[6 * gi + 5] = cov.c5;
  if (hit_miss_layer)
  {
    intensity_layer[2 * gi] = int_mean;
    intensity_layer[2 * gi + 1] = int_cov;
    hit_miss_layer[2 * gi] = hm_hit;
    hit_miss_layer[2 * gi + 1] = hm_miss;
  }
}

__global__ void __launch_bounds__(128)
  k_replay_ndt(MapConst mc, RegionTable rt, const unsigned long long *__restrict__ sorted, uint32_t n_events,
               const double *__restrict__ rays, const float *__restrict__ intensities, float *__restrict__ occupancy,
               uint32_t *__restrict__ mean_layer, float *__restrict__ cov_layer, float *__restrict__ intensity_layer,
               uint32_t *__restrict__ hit_miss_layer, SecondaryLayers sec, const RayWalk *__restrict__ walks,
               const uint32_t *__restrict__ heads, const uint32_t *__restrict__ n_heads,
               const uint32_t *__restrict__ event_count, uint32_t event_limit)
{
  if (speculationFailed(event_count, event_limit))
  {
    return;
  }
  // One lane per voxel group (k_group_heads), grid-stride.
  const NdtLayers ly{ occupancy, mean_layer, cov_layer, intensity_layer, hit_miss_layer };
  const uint32_t head_count = *n_heads;
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < head_count; h += gridDim.x * blockDim.x)
  {
    replayNdtGroup(mc, rt, sorted, heads[h], n_events, rays, intensities, ly, sec, walks);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TSDF: RayMapperTsdf::integrateRays per-voxel semantics (ohm/RayMapperTsdf.cpp:105-160).
// ---------------------------------------------------------------------------------------------------------------------
/// Replay ONE voxel group of TSDF events (see replayNdtGroup).
__device__ inline void replayTsdfGroup(const MapConst &mc, const RegionTable &rt, const unsigned long long *sorted,
                                       uint32_t i, uint32_t n_events, const double *__restrict__ rays,
                                       float *__restrict__ tsdf_layer)
{
  const unsigned long long key = sorted[i];
  const unsigned long long group = key >> kHitRayBits;
  const uint32_t slot = uint32_t(key >> kHitSlotShift);
  const uint32_t vi = uint32_t(key >> kHitRayBits) & ((1u << kHitVoxelBits) - 1u);
  const size_t gi = size_t(slot) * size_t(mc.region_voxels) + vi;
  double centre_a[3];
  voxelCentreOf(mc, rt, slot, vi, centre_a);
  const D3 centre = d3(centre_a[0], centre_a[1], centre_a[2]);
  float weight = tsdf_layer[2 * gi];
  float distance = tsdf_layer[2 * gi + 1];
  for (uint32_t j = i; j < n_events; ++j)
  {
    const unsigned long long kj = sorted[j];
    if ((kj >> kHitRayBits) != group)
    {
      break;
    }
    const uint32_t ray = uint32_t((kj & kEvRayMask) >> kEvRayShift);
    // calculateTsdf takes the ORIGINAL (unfiltered) sensor / sample (ohm/RayMapperTsdf.cpp:163-164).
    const D3 sensor = d3(rays[size_t(ray) * 6 + 0], rays[size_t(ray) * 6 + 1], rays[size_t(ray) * 6 + 2]);
    const D3 sample = d3(rays[size_t(ray) * 6 + 3], rays[size_t(ray) * 6 + 4], rays[size_t(ray) * 6 + 5]);
    const float sdf = tsdfComputeDistance(sensor, sample, centre);
    tsdfUpdate(mc, sdf, weight, distance);
  }
  tsdf_layer[2 * gi] = weight;
  tsdf_layer[2 * gi + 1] = distance;
}

__global__ void __launch_bounds__(128)
  k_replay_tsdf(MapConst mc, RegionTable rt, const unsigned long long *__restrict__ sorted, uint32_t n_events,
                const double *__restrict__ rays, float *__restrict__ tsdf_layer, const uint32_t *__restrict__ heads,
                const uint32_t *__restrict__ n_heads, const uint32_t *__restrict__ event_count, uint32_t event_limit)
{
  if (speculationFailed(event_count, event_limit))
  {
    return;
  }
  // Grid-stride over the compacted voxel-group heads (k_group_heads) or, with heads == nullptr, over all events with
  // the non-heads skipped (TSDF groups are short: the compaction pass costs more than it saves there).
  const uint32_t head_count = heads ? *n_heads : n_events;
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < head_count; h += gridDim.x * blockDim.x)
  {
    const uint32_t i = heads ? heads[h] : h;
    const unsigned long long key = sorted[i];
    if (!heads && (key == kHitInvalid || (i > 0 && (sorted[i - 1] >> kHitRayBits) == (key >> kHitRayBits))))
    {
      continue;
    }
    replayTsdfGroup(mc, rt, sorted, i, n_events, rays, tsdf_layer);
  }
}

/// TSDF: voxels which only saw free-space visits this batch (count n, none flagged): weight = min(weight + n,
/// max_weight) (n unit increments, exact for integer-valued floats), distance = truncation distance -- the fixed point of
/// calculateTsdf for sdf >= truncation distance.  kTsdfApplyParts workgroups share a region (whole mask words each, four
/// voxels per lane and step): one 256-thread workgroup per region ran 128 dependent load -> store rounds (515 us of a
/// 10.3 ms C3 batch); the regions' per-batch scratch is put back by k_batch_reset behind this launch (every part reads
/// seg_count when it starts, so none of them may clear it).
constexpr uint32_t kTsdfApplyParts = 8;

__global__ void __launch_bounds__(256)
  k_apply_counts_tsdf(MapConst mc, RegionTable rt, BatchScratch bs, uint32_t *__restrict__ miss_counts,
                      const uint32_t *__restrict__ hit_mask, float *__restrict__ tsdf_layer,
                      uint32_t direct_chunk_segments)
{
  const uint32_t h = bs.touched[blockIdx.x / kTsdfApplyParts];
  const uint32_t part = blockIdx.x % kTsdfApplyParts;
  const uint32_t slot = rt.vals[h];
  const size_t base = size_t(slot) * size_t(mc.region_voxels);
  const uint32_t words = uint32_t(mc.region_voxels + 31) >> 5;
  const uint32_t *mask = hit_mask + size_t(slot) * words;
  // Regions with a single chunk were applied by the walk kernel itself (direct_chunk_segments != 0).
  const uint32_t region_segments = bs.seg_count[h];
  if (direct_chunk_segments && region_segments > 0 && region_segments <= direct_chunk_segments)
  {
    return;
  }
  const uint32_t v_lo = uint32_t(uint64_t(words) * part / kTsdfApplyParts) * 32u;
  const uint32_t v_hi = min(uint32_t(uint64_t(words) * (part + 1u) / kTsdfApplyParts) * 32u, uint32_t(mc.region_voxels));
  auto apply = [&](uint32_t vi, uint32_t n) {
    const float w = tsdf_layer[2 * (base + vi)];
    const float wn = w + float(n);
    tsdf_layer[2 * (base + vi)] = (mc.tsdf_max_weight < wn) ? mc.tsdf_max_weight : wn;
    tsdf_layer[2 * (base + vi) + 1] = mc.tsdf_trunc;
  };
  if (mc.region_voxels % 4 == 0)
  {
    uint4 *counts4 = reinterpret_cast<uint4 *>(miss_counts + base);
    for (uint32_t q = v_lo / 4u + threadIdx.x; q < v_hi / 4u; q += blockDim.x)
    {
      const uint4 n = counts4[q];
      if (n.x | n.y | n.z | n.w)
      {
        // near-surface (flagged) voxels are updated by the ordered replay only: their counts are dropped
        const uint32_t bits = (mask[(4u * q) >> 5] >> ((4u * q) & 31u)) & 15u;
        if (n.x && !(bits & 1u)) { apply(4u * q + 0u, n.x); }
        if (n.y && !(bits & 2u)) { apply(4u * q + 1u, n.y); }
        if (n.z && !(bits & 4u)) { apply(4u * q + 2u, n.z); }
        if (n.w && !(bits & 8u)) { apply(4u * q + 3u, n.w); }
        counts4[q] = make_uint4(0, 0, 0, 0);
      }
    }
    return;
  }
  for (uint32_t vi = v_lo + threadIdx.x; vi < v_hi; vi += blockDim.x)
  {
    const uint32_t n = miss_counts[base + vi];
    if (n)
    {
      if (!((mask[vi >> 5] >> (vi & 31u)) & 1u))
      {
        apply(vi, n);
      }
      miss_counts[base + vi] = 0;
    }
  }
}

/// Per-batch scratch of the touched regions back to its idle state (what a kernel with one workgroup per region does
/// itself at its end).
__global__ void __launch_bounds__(256) k_batch_reset(BatchScratch bs, uint32_t n_touched)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_touched)
  {
    const uint32_t h = bs.touched[i];
    bs.seg_count[h] = 0;
    bs.seg_cursor[h] = 0;
    bs.touched_flag[h] = 0;
  }
}

/// Safety margin on the truncation distance used to classify a visit as "free space" (count only).  A visit with
/// sdf >= kTsdfFreeMargin * trunc leaves a (weight, trunc) voxel at (min(weight + 1, max), trunc) exactly, including
/// float rounding of (sdf + trunc * w) / (w + 1) for w <= 1e4 (needs > 2e-3 relative margin; 1 % is used).
constexpr float kTsdfFreeMargin = 1.01f;

// ---------------------------------------------------------------------------------------------------------------------
// kRfStopOnFirstOccupied (ohm/RayFlag.h:28; ohm/RayMapperOccupancy.cpp:105-193, 222-239).
//
// CPU semantics: a ray is walked from its origin; every voxel gets its miss update until a voxel is met whose value
// BEFORE this ray's update is occupied (observed and >= the threshold): that voxel still gets its miss, every later
// voxel of the ray a null update, and the ray's sample is not applied.  Whether a voxel is occupied when ray r
// reaches it depends on what the rays before r did to it -- which depends on where THOSE rays stopped.  There is no
// per-voxel counting shortcut, so such a batch takes the fully general route: every visit of the batch is an event
// (WalkArgs::flag_all), events and samples are sorted per voxel in ray order, and the per-ray stop positions are found
// by iteration:
//   scan (k_stop_replay<false>): replay every voxel with the current stop estimates deciding which visits are live;
//     for every visit -- live or not -- of ray r to a voxel that is occupied at that moment, r's stop candidate is
//     lowered to the visit's position along the ray (the Manhattan distance of the voxel from the ray's start voxel,
//     which is the visit's index in the walk).
//   A ray's stop depends only on the stops of rays before it, so after k scans the first k rays are final and a scan
//   that changes nothing has produced the sequential result; real batches settle in a handful of scans.
//   commit (k_stop_replay<true>): the same replay with the final stops, writing the layers.
// Slow next to the counting path (every visit is sorted), bit-identical to the CPU mapper.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kNoStop = 0xffffffffu;

template <bool kCommit>
__global__ void __launch_bounds__(128)
  k_stop_replay(MapConst mc, RegionTable rt, const unsigned long long *__restrict__ sorted, uint32_t n_events,
                unsigned ray_flags, const RayWalk *__restrict__ walks, const uint32_t *__restrict__ stop,
                uint32_t *__restrict__ stop_next, const double *__restrict__ rays, float *__restrict__ occupancy,
                uint32_t *__restrict__ mean, SecondaryLayers sec)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_events; i += gridDim.x * blockDim.x)
  {
    const unsigned long long key = sorted[i];
    const unsigned long long group = key >> kHitRayBits;
    if (key == kHitInvalid || (i > 0 && (sorted[i - 1] >> kHitRayBits) == group))
    {
      continue;  // not the head of its voxel group
    }
    const uint32_t slot = uint32_t(key >> kHitSlotShift);
    const uint32_t vi = uint32_t(key >> kHitRayBits) & ((1u << kHitVoxelBits) - 1u);
    const size_t gi = size_t(slot) * size_t(mc.region_voxels) + vi;
    int16_t rk[3];
    unpackRegionKey(rt.slot_keys[slot], rk);
    const int lx = int(vi % uint32_t(mc.dim[0]));
    const int ly = int((vi / uint32_t(mc.dim[0])) % uint32_t(mc.dim[1]));
    const int lz = int(vi / uint32_t(mc.dim[0] * mc.dim[1]));
    const int gx = int(rk[0]) * mc.dim[0] + lx;
    const int gy = int(rk[1]) * mc.dim[1] + ly;
    const int gz = int(rk[2]) * mc.dim[2] + lz;
    const float inf = __int_as_float(0x7f800000);
    float x = occupancy[gi];
    uint32_t mcoord = 0, mcount = 0;
    double centre[3] = { 0, 0, 0 };
    uint32_t packed_normal = 0;
    uint32_t last_sample_ray = kNoStop;
    float traversal = 0.0f;
    if (kCommit)
    {
      traversal = sec.traversal ? sec.traversal[gi] : 0.0f;
      if (mean)
      {
        mcoord = mean[2 * gi];
        mcount = mean[2 * gi + 1];
        centre[0] = globalVoxelCentreAxis(mc, 0, gx);
        centre[1] = globalVoxelCentreAxis(mc, 1, gy);
        centre[2] = globalVoxelCentreAxis(mc, 2, gz);
      }
      packed_normal = sec.incident ? sec.incident[gi] : 0u;
    }
    for (uint32_t j = i; j < n_events; ++j)
    {
      const unsigned long long kj = sorted[j];
      if ((kj >> kHitRayBits) != group)
      {
        break;
      }
      const uint32_t ray = uint32_t((kj & kEvRayMask) >> kEvRayShift);
      const bool is_sample = (kj & 1ull) != 0;
      const uint32_t ray_stop = stop[ray];
      if (is_sample)
      {
        if (ray_stop != kNoStop)
        {
          continue;  // the ray stopped on its way: no sample update (ohm/RayMapperOccupancy.cpp:234)
        }
        x = occHit(mc, ray_flags, x);
        if (kCommit)
        {
          last_sample_ray = ray;
          if (sec.traversal)
          {
            // ohm/RayMapperOccupancy.cpp:299-305: remaining ray length inside the sample voxel (only for a ray that
            // applies its sample, i.e. one that was not stopped).
            const double dx = rays[size_t(ray) * 6 + 3] - rays[size_t(ray) * 6 + 0];
            const double dy = rays[size_t(ray) * 6 + 4] - rays[size_t(ray) * 6 + 1];
            const double dz = rays[size_t(ray) * 6 + 5] - rays[size_t(ray) * 6 + 2];
            traversal += float(sqrt((dx * dx + dy * dy) + dz * dz) - lastExitRange(walks, ray));
          }
          if (sec.incident)
          {
            const float dir[3] = { float(rays[size_t(ray) * 6 + 0] - rays[size_t(ray) * 6 + 3]),
                                   float(rays[size_t(ray) * 6 + 1] - rays[size_t(ray) * 6 + 4]),
                                   float(rays[size_t(ray) * 6 + 2] - rays[size_t(ray) * 6 + 5]) };
            packed_normal = updateIncidentNormal(packed_normal, dir, mean ? mcount : 0u);
          }
          if (mean)
          {
            const double local[3] = { rays[size_t(ray) * 6 + 3] - centre[0], rays[size_t(ray) * 6 + 4] - centre[1],
                                      rays[size_t(ray) * 6 + 5] - centre[2] };
            mcoord = subVoxelUpdate(mcoord, mcount, local, mc.resolution);
            ++mcount;
          }
        }
        continue;
      }
      // a visit of the ray part: its index in the ray's walk
      const int *g0 = walks[ray].g0;
      const uint32_t position = uint32_t(abs(gx - g0[0]) + abs(gy - g0[1]) + abs(gz - g0[2]));
      const bool occupied = x != inf && x >= mc.threshold_value;
      if (!kCommit && occupied)
      {
        atomicMin(&stop_next[ray], position);
      }
      if (position <= ray_stop)
      {
        x = occMiss(mc, ray_flags, x);  // live: at or before the ray's stopping voxel
      }
    }
    if (kCommit)
    {
      occupancy[gi] = x;
      if (mean)
      {
        mean[2 * gi] = mcoord;
        mean[2 * gi + 1] = mcount;
      }
      if (sec.incident)
      {
        sec.incident[gi] = packed_normal;
      }
      if (sec.traversal && last_sample_ray != kNoStop)
      {
        sec.traversal[gi] = traversal;
      }
      if (sec.touch_time && sec.timestamps && last_sample_ray != kNoStop)
      {
        sec.touch_time[gi] = encodeVoxelTouchTime(sec.time_base, sec.timestamps[last_sample_ray]);
      }
    }
  }
}

/// stop <- stop_next, stop_next <- "no stop"; *changed is set when any ray's stop moved.
__global__ void __launch_bounds__(256)
  k_stop_advance(uint32_t *__restrict__ stop, uint32_t *__restrict__ stop_next, uint32_t n_rays,
                 uint32_t *__restrict__ changed)
{
  const uint32_t ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray < n_rays)
  {
    const uint32_t next = stop_next[ray];
    if (next != stop[ray])
    {
      stop[ray] = next;
      *changed = 1u;
    }
    stop_next[ray] = kNoStop;
  }
}

/// TSDF pre-pass, one lane per ray: flag every voxel near the ray's end whose sdf is below the free-space margin.
/// Those voxels (and, persistently, every voxel ever flagged) take the ordered replay path.
__global__ void __launch_bounds__(256)
  k_tsdf_flag(MapConst mc, RegionTable rt, const RayWalk *__restrict__ walks, const double *__restrict__ rays,
              uint32_t n_rays, uint32_t *__restrict__ hit_mask)
{
  const uint32_t ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays)
  {
    return;
  }
  const RayWalk rw = walks[ray];
  if (!(rw.flags & kRwValid))
  {
    return;
  }
  double start[3], end[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    start[a] = rays[size_t(ray) * 6 + a];
    end[a] = rays[size_t(ray) * 6 + 3 + a];
  }
  const D3 sensor = d3(start[0], start[1], start[2]);
  const D3 sample = d3(end[0], end[1], end[2]);
  bool clipped = false;
  filterRay(mc, start, end, clipped, ray);
  const double dx = end[0] - start[0], dy = end[1] - start[1], dz = end[2] - start[2];
  const double length = sqrt((dx * dx + dy * dy) + dz * dz);
  // Any voxel left before ray parameter t_star has its centre's projection at least margin * trunc short of the
  // sample (centre projection is within one voxel edge... of the voxel's ray interval; 1 voxel of slack is used).
  const double t_star = length - (double(kTsdfFreeMargin) * double(mc.tsdf_trunc) + 2.0 * mc.resolution);
  int s0 = 0, s1 = 0, s2 = 0;
  if (t_star > 0)
  {
    // Steps strictly before t_star form a valid prefix of the walk (axis == 3 disables the tie rule).
    s0 = stepsBefore(rw.init[0], rw.delta[0], 1.0 / rw.delta[0], rw.total[0], 0, 3, t_star);
    s1 = stepsBefore(rw.init[1], rw.delta[1], 1.0 / rw.delta[1], rw.total[1], 1, 3, t_star);
    s2 = stepsBefore(rw.init[2], rw.delta[2], 1.0 / rw.delta[2], rw.total[2], 2, 3, t_star);
  }
  const int d0 = rwDir(rw, 0), d1 = rwDir(rw, 1), d2 = rwDir(rw, 2);
  int g0 = rw.g0[0] + d0 * s0, g1 = rw.g0[1] + d1 * s1, g2 = rw.g0[2] + d2 * s2;
  int rem0 = rw.total[0] - s0, rem1 = rw.total[1] - s1, rem2 = rw.total[2] - s2;
  const double inf = dInf();
  double k0 = double(s0), k1 = double(s1), k2 = double(s2);
  double t0 = rem0 ? ((s0 == 0) ? rw.init[0] : rw.init[0] + rw.delta[0] * k0) : inf;
  double t1 = rem1 ? ((s1 == 0) ? rw.init[1] : rw.init[1] + rw.delta[1] * k1) : inf;
  double t2 = rem2 ? ((s2 == 0) ? rw.init[2] : rw.init[2] + rw.delta[2] * k2) : inf;
  uint64_t cached_key = 0;
  uint32_t cached_slot = kSlotUnassigned;
  const uint32_t mask_words = uint32_t(mc.region_voxels + 31) >> 5;
  while (true)
  {
    int r0, r1, r2, l0, l1, l2;
    splitGlobal(g0, mc.dim[0], r0, l0);
    splitGlobal(g1, mc.dim[1], r1, l1);
    splitGlobal(g2, mc.dim[2], r2, l2);
    const D3 centre = d3(globalVoxelCentreAxis(mc, 0, g0), globalVoxelCentreAxis(mc, 1, g1), globalVoxelCentreAxis(mc, 2, g2));
    const float sdf = tsdfComputeDistance(sensor, sample, centre);
    if (sdf < kTsdfFreeMargin * mc.tsdf_trunc)
    {
      const uint64_t rkey = packRegionKey(r0, r1, r2);
      if (rkey != cached_key)
      {
        const uint32_t h = regionFind(rt, rkey);
        cached_slot = (h != 0xffffffffu) ? rt.vals[h] : kSlotUnassigned;
        cached_key = rkey;
      }
      if (cached_slot < rt.slot_capacity)
      {
        const uint32_t vi = uint32_t(l0 + l1 * mc.dim[0] + l2 * mc.dim[0] * mc.dim[1]);
        // (the TSDF mask is persistent: a surface that has been seen before has its voxels flagged already -- a load
        // instead of an atomic for nearly every visit of a steady-state batch; a stale 0 only costs the atomic)
        uint32_t *word = &hit_mask[size_t(cached_slot) * mask_words + (vi >> 5)];
        if (!(*word & (1u << (vi & 31))))
        {
          atomicOr(word, 1u << (vi & 31));
        }
      }
    }
    if ((rem0 | rem1 | rem2) == 0)
    {
      break;
    }
    const bool c01 = t0 < t1;
    const double t01 = c01 ? t0 : t1;
    const bool c2 = t01 < t2;
    if (!c2)
    {
      g2 += d2;
      --rem2;
      k2 += 1.0;
      t2 = rem2 ? rw.init[2] + rw.delta[2] * k2 : inf;
    }
    else if (c01)
    {
      g0 += d0;
      --rem0;
      k0 += 1.0;
      t0 = rem0 ? rw.init[0] + rw.delta[0] * k0 : inf;
    }
    else
    {
      g1 += d1;
      --rem1;
      k1 += 1.0;
      t1 = rem1 ? rw.init[1] + rw.delta[1] * k1 : inf;
    }
  }
}

/// Rebuild the persistent "ordered replay" voxel mask of one region from its stored layers (after a CPU upload):
/// NDT: voxels holding samples (mean.count > 0); TSDF: observed voxels whose distance is not the free-space value.
__global__ void __launch_bounds__(256)
  k_rebuild_mask(MapConst mc, uint32_t slot, const uint32_t *__restrict__ mean_layer,
                 const float *__restrict__ tsdf_layer, uint32_t *__restrict__ hit_mask)
{
  const uint32_t mask_words = uint32_t(mc.region_voxels + 31) >> 5;
  const size_t base = size_t(slot) * size_t(mc.region_voxels);
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < mask_words; w += gridDim.x * blockDim.x)
  {
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32; ++b)
    {
      const uint32_t vi = w * 32 + b;
      if (vi < uint32_t(mc.region_voxels))
      {
        bool flag = false;
        if (mean_layer)
        {
          flag = mean_layer[2 * (base + vi) + 1] > 0;
        }
        if (tsdf_layer)
        {
          const float wgt = tsdf_layer[2 * (base + vi)];
          const float dist = tsdf_layer[2 * (base + vi) + 1];
          flag = flag || (wgt != 0.0f && dist != mc.tsdf_trunc) || (wgt == 0.0f && dist != 0.0f);
        }
        bits |= flag ? (1u << b) : 0u;
      }
    }
    hit_mask[size_t(slot) * mask_words + w] = bits;
  }
}
/// GpuKey layout of the reference (ohmgpu/GpuKey.h:37-46): short region[3]; uchar voxel[4].
struct GpuKeyOut
{
  int16_t region[3];
  uint8_t voxel[4];
};
// The reference's record, from its own header compiled in place (tests/golden/ref_vectors.npz: gpukey_layout).
static_assert(sizeof(GpuKeyOut) == 10 && alignof(GpuKeyOut) == 2 && offsetof(GpuKeyOut, region) == 0 &&
                offsetof(GpuKeyOut, voxel) == 6,
              "GpuKeyOut must keep the layout of ohm::GpuKey (ohmgpu/GpuKey.h:37-46)");

/// LineKeysQueryGpu / `calculateLines` (ohmgpu/gpu/LineKeys.cl:66-100) with the CPU walk's semantics
/// (ohm/LineWalk.h:112-129 walkSegmentKeys, flags 0: start and end voxel included): one lane per query line writes the
/// keys of every voxel on the line, in walk order.  counts[i] is the full number of voxels even when it exceeds
/// max_keys_per_line (only the first max_keys_per_line keys are stored).
__global__ void __launch_bounds__(256)
  k_line_keys(MapConst mc, const double *__restrict__ lines, uint32_t n_lines, uint32_t max_keys_per_line,
              GpuKeyOut *__restrict__ keys_out, uint32_t *__restrict__ counts)
{
  const uint32_t line = blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= n_lines)
  {
    return;
  }
  double start[3], end[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    start[a] = lines[size_t(line) * 6 + a];
    end[a] = lines[size_t(line) * 6 + 3 + a];
  }
  MapConst nofilter = mc;
  nofilter.filter_mode = OHMHIP_FILTER_NONE;
  nofilter.batch_filter_flags = nullptr;
  RayWalk rw;
  setupRay(nofilter, start, end, OHMHIP_RF_END_POINT_AS_FREE, rw, line);
  if (!(rw.flags & kRwValid))
  {
    counts[line] = 0;
    return;
  }
  const int d0 = rwDir(rw, 0), d1 = rwDir(rw, 1), d2 = rwDir(rw, 2);
  int g0 = rw.g0[0], g1 = rw.g0[1], g2 = rw.g0[2];
  int rem0 = rw.total[0], rem1 = rw.total[1], rem2 = rw.total[2];
  const double inf = dInf();
  double k0 = 0, k1 = 0, k2 = 0;
  double t0 = rem0 ? rw.init[0] : inf;
  double t1 = rem1 ? rw.init[1] : inf;
  double t2 = rem2 ? rw.init[2] : inf;
  uint32_t n = 0;
  GpuKeyOut *out = keys_out + size_t(line) * max_keys_per_line;
  while (true)
  {
    if (n < max_keys_per_line)
    {
      int r0, r1, r2, l0, l1, l2;  // the caller's region key: region edge, not tile edge
      splitGlobal(g0, mc.kdim[0], r0, l0);
      splitGlobal(g1, mc.kdim[1], r1, l1);
      splitGlobal(g2, mc.kdim[2], r2, l2);
      GpuKeyOut k;
      k.region[0] = int16_t(r0);
      k.region[1] = int16_t(r1);
      k.region[2] = int16_t(r2);
      k.voxel[0] = uint8_t(l0);
      k.voxel[1] = uint8_t(l1);
      k.voxel[2] = uint8_t(l2);
      k.voxel[3] = 0;
      out[n] = k;
    }
    ++n;
    if ((rem0 | rem1 | rem2) == 0)
    {
      break;
    }
    const bool c01 = t0 < t1;
    const double t01 = c01 ? t0 : t1;
    const bool c2 = t01 < t2;
    if (!c2)
    {
      g2 += d2;
      --rem2;
      k2 += 1.0;
      t2 = rem2 ? rw.init[2] + rw.delta[2] * k2 : inf;
    }
    else if (c01)
    {
      g0 += d0;
      --rem0;
      k0 += 1.0;
      t0 = rem0 ? rw.init[0] + rw.delta[0] * k0 : inf;
    }
    else
    {
      g1 += d1;
      --rem1;
      k1 += 1.0;
      t1 = rem1 ? rw.init[1] + rw.delta[1] * k1 : inf;
    }
  }
  counts[line] = n;
}
}  // namespace ohmhip

#endif  // OHMHIP_REPLAY_KERNELS_H
