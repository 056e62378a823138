// occupancy_kernels.h -- the gfx950 kernels of the occupancy ray-integration path.
//
// Pipeline per ray batch (all on one HIP stream; see DESIGN.md for the full picture):
//   k_ray_setup         1 lane / ray       filter, keys, fp64 line-walk set-up; enumerate the regions the ray crosses;
//                                          segment and sample counts per region in a workgroup-level LDS table
//   k_plan              1 workgroup        scans over the touched regions: segment / sample offsets, equal-size chunks
//                                          (largest first), sample-sort order
//   k_ray_bin           1 lane / ray       scatter segment records (resume state + voxel count) and sample keys into
//                                          the per-region ranges; set the sample bitmask
//   k_sort_region_hits  1 workgroup/region order a region's sample keys by (voxel, ray) in LDS
//                                          (fallback for very dense regions: device-wide radix sort + k_hit_bounds)
//   k_region_walk       persistent, 1 workgroup / CU   THE hot kernel: chunks from a device-wide cursor; region
//                                          miss-count tile in LDS, 1 lane / segment resumes the fp64 walk inside the
//                                          region; LDS atomics; single-chunk regions applied straight from LDS
//   k_apply_hits        1 lane / sample    ordered replay (misses-before-hit counts, hit, mean) for voxels with samples
//   k_apply_counts      1 workgroup/region remaining miss counts of multi-chunk regions, clear scratch
//
// Ordering argument (why integer counting reproduces the sequential CPU result exactly): every miss applies the
// same function m(x) and every hit the same h(x) to a voxel's value.  The CPU result for a voxel is the composition
// of its events in ray order; that composition is fully determined by the NUMBER of misses between consecutive
// hits.  Integer atomics are order-independent, so counting is deterministic, and the float updates are then
// replayed one voxel per lane in exactly the CPU order -> bit-identical log-odds.
#ifndef OHMHIP_OCCUPANCY_KERNELS_H
#define OHMHIP_OCCUPANCY_KERNELS_H

#include "secondary_device.h"
#include "walk_device.h"

namespace ohmhip
{
struct RegionTable
{
  unsigned long long *keys;  ///< [hash_capacity] packed region key or 0
  uint32_t *vals;            ///< [hash_capacity] slot index
  uint64_t *slot_keys;       ///< [slot_capacity] packed key per slot
  uint32_t *n_slots;         ///< number of slots handed out
  uint32_t hash_mask;
  uint32_t slot_capacity;
};

/// Per-batch scratch indexed by hash index / slot.
struct BatchScratch
{
  uint32_t *seg_count;     ///< [hash_capacity]
  uint32_t *seg_cursor;    ///< [hash_capacity]
  uint32_t *seg_offset;    ///< [hash_capacity]
  uint32_t *touched_flag;  ///< [hash_capacity]
  uint32_t *touched;       ///< [hash_capacity] list of touched hash indices
  uint32_t *hit_count;     ///< [hash_capacity] samples per region (k_ray_setup -> k_plan, which zeroes it again)
  uint32_t *sort_list;     ///< [hash_capacity] regions receiving samples, most samples first (k_plan -> sort)
  uint32_t *apply_counts_list;  ///< [hash_capacity] regions cut into several chunks: k_apply_counts_list applies their counts
  uint32_t *apply_hits_list;    ///< [hash_capacity] regions whose samples the walk does not replay itself: k_apply_hits_list
  uint32_t *hit_begin;     ///< [slot_capacity] first sample of the region in the sorted list
  uint32_t *hit_end;       ///< [slot_capacity]
  uint32_t *dirty;         ///< [slot_capacity]
  uint32_t *last_use;      ///< [2 x slot_capacity] use history per slot (touchRegionUse; spill to host)
  uint32_t stamp;          ///< this batch's stamp
  uint32_t *voxel_first_hit;  ///< [slot_capacity * region_voxels] index of a voxel's first sample in the sorted list
  BatchInfo *info;
  struct WgRegion *wg_regions;  ///< [workgroups * kLtabSize] regions each binning workgroup feeds (k_ray_setup -> bin)
  uint32_t *wg_region_count;    ///< [workgroups]
};

/// One region a binning workgroup feeds: written by k_ray_setup, consumed by k_ray_bin (same workgroup -> rays mapping),
/// which therefore does not have to enumerate the rays' regions a second time just to count.
struct WgRegion
{
  unsigned long long key;
  uint32_t count;  ///< segments of the workgroup's rays in the region
  uint32_t entry;  ///< position in the workgroup's LDS region table
  uint32_t hash;   ///< index in the global region table
  uint32_t hits;   ///< samples of the workgroup's rays in the region
};

enum : uint32_t
{
  kErrHashFull = 1u << 0,
  kErrSlotsFull = 1u << 1,
  kErrSegments = 1u << 2
};

// ---------------------------------------------------------------------------------------------------------------------
// Region hash table
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline uint32_t regionInsert(const RegionTable &rt, uint64_t key, uint32_t *error)
{
  uint32_t idx = hashRegionKey(key, rt.hash_mask);
  for (uint32_t probe = 0; probe <= rt.hash_mask; ++probe)
  {
    unsigned long long prev = rt.keys[idx];
    if (prev == 0)
    {
      prev = atomicCAS(&rt.keys[idx], 0ull, (unsigned long long)key);
      if (prev == 0)
      {
        const uint32_t slot = atomicAdd(rt.n_slots, 1u);
        if (slot < rt.slot_capacity)
        {
          rt.slot_keys[slot] = key;
        }
        else
        {
          atomicOr(error, kErrSlotsFull);
        }
        // Published for later kernels; nothing in this kernel reads vals[].
        rt.vals[idx] = slot;
        return idx;
      }
    }
    if (prev == key)
    {
      return idx;
    }
    idx = (idx + 1) & rt.hash_mask;
  }
  atomicOr(error, kErrHashFull);
  return 0;
}

__device__ inline uint32_t regionFind(const RegionTable &rt, uint64_t key)
{
  uint32_t idx = hashRegionKey(key, rt.hash_mask);
  for (uint32_t probe = 0; probe <= rt.hash_mask; ++probe)
  {
    const unsigned long long k = rt.keys[idx];
    if (k == key)
    {
      return idx;
    }
    if (k == 0)
    {
      break;
    }
    idx = (idx + 1) & rt.hash_mask;
  }
  return 0xffffffffu;
}

// ---------------------------------------------------------------------------------------------------------------------
// Region enumeration for one ray: a 3-way merge of the per-axis region-crossing steps in walk order.
// ---------------------------------------------------------------------------------------------------------------------
struct RegionCursor
{
  int region[3];  ///< current region
  int next_j[3];  ///< step index (1-based) of the next region crossing per axis; > total => none left
  int dir[3];
};

__device__ inline void regionCursorInit(const MapConst &mc, const RayWalk &rw, RegionCursor &rc)
{
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    int local;
    splitGlobal(rw.g0[a], mc.dim[a], rc.region[a], local);
    rc.dir[a] = rwDir(rw, a);
    // Steps needed to leave the start region along this axis.
    rc.next_j[a] = (rc.dir[a] > 0) ? (mc.dim[a] - local) : (local + 1);
  }
}

/// Advance to the next region crossing.  Returns false when the ray crosses no further region boundary.
/// On success `axis`/`j` identify the step which enters the new region and rc.region is updated.
__device__ inline bool regionCursorNext(const MapConst &mc, const RayWalk &rw, RegionCursor &rc, int &axis, int &j)
{
  int best = -1;
  double best_t = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    if (rc.next_j[a] <= rw.total[a])
    {
      const double t = stepTime(rw.init[a], rw.delta[a], rc.next_j[a]);
      if (best < 0 || stepPrecedes(t, a, best_t, best))
      {
        best = a;
        best_t = t;
      }
    }
  }
  if (best < 0)
  {
    return false;
  }
  axis = best;
  j = sel3(best, rc.next_j[0], rc.next_j[1], rc.next_j[2]);
  // Per-axis updates spelled out: a run-time index here would push the cursor into scratch memory.
  if (best == 0)
  {
    rc.region[0] += rc.dir[0];
    rc.next_j[0] += mc.dim[0];
  }
  else if (best == 1)
  {
    rc.region[1] += rc.dir[1];
    rc.next_j[1] += mc.dim[1];
  }
  else
  {
    rc.region[2] += rc.dir[2];
    rc.next_j[2] += mc.dim[2];
  }
  return true;
}

/// Is the voxel reached by step (axis, j) the ray's end voxel?  (All three axes exhausted.)  True only when j is the
/// last step on its axis and every step of the other axes precedes it.
__device__ inline bool stepReachesEnd(const RayWalk &rw, int axis, int j)
{
  if (j != sel3(axis, rw.total[0], rw.total[1], rw.total[2]))
  {
    return false;
  }
  const double ta =
    stepTime(sel3(axis, rw.init[0], rw.init[1], rw.init[2]), sel3(axis, rw.delta[0], rw.delta[1], rw.delta[2]), j);
  bool all_before = true;
#pragma unroll
  for (int b = 0; b < 3; ++b)
  {
    if (rw.total[b] > 0)
    {
      const double tb = stepTime(rw.init[b], rw.delta[b], rw.total[b]);
      all_before = all_before && (b == axis || stepPrecedes(tb, b, ta, axis));
    }
  }
  return all_before;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-level aggregation: lanes with equal 64-bit keys elect a leader which performs one operation for the group.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline unsigned laneId()
{
  return __lane_id();
}

__device__ inline uint64_t shfl64(uint64_t v, int src)
{
  const uint32_t lo = __shfl(uint32_t(v), src);
  const uint32_t hi = __shfl(uint32_t(v >> 32), src);
  return (uint64_t(hi) << 32) | lo;
}

/// Wave-level match: for every lane with `has`, find the lowest lane holding the same 32-bit value and the mask of all
/// lanes holding it.  Compute only (no memory traffic), one loop trip per distinct value in the wave.
__device__ inline void waveMatch(bool has, uint32_t value, unsigned lane, int &leader, unsigned long long &group)
{
  leader = -1;
  group = 0;
  unsigned long long todo = __ballot(has);
  while (todo)
  {
    const int l = __ffsll((long long)todo) - 1;
    const uint32_t lv = __shfl(value, l);
    const bool mine = has && value == lv;
    const unsigned long long same = __ballot(mine);
    if (mine)
    {
      leader = l;
      group = same;
    }
    todo &= ~same;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Block-level region table in LDS.
//
// Rays of one workgroup cross the same few dozen regions.  Counting (k_ray_setup) and bucket reservation (k_ray_bin)
// therefore aggregate per workgroup in an LDS hash table and touch each global per-region counter ONCE per workgroup:
// the per-region counters of the regions around a sensor are otherwise hit by every wave of the launch, and atomics on
// one address serialise at the memory side (that, not arithmetic, dominated the first version of these kernels).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBinThreads = 512;        ///< workgroup size of the binning kernels for large batches (launch bound)
constexpr int kBinRaysPerBlock = 1024;  ///< rays per binning workgroup for large batches; small batches use fewer so the
                                        ///< launch still spreads over the CUs (the host picks both per batch)
constexpr uint32_t kLtabSize = 2048;  ///< entries (power of two)

constexpr uint32_t kLtabSmall = 256;  ///< entries of the small-batch instantiations (128-ray workgroups)

/// kTab entries (kLtabSize, or kLtabSmall for the small-batch instantiations of k_ray_setup / k_ray_bin: with the full
/// table's 40 KiB of static LDS only three of their two-wave workgroups fit a CU and the kernels are latency bound).
template <uint32_t kTab>
struct LdsRegionTableT
{
  unsigned long long keys[kTab];
  uint32_t count[kTab];   ///< k_ray_setup: segments of this workgroup in the region; k_ray_bin: sample cursor
  uint32_t cursor[kTab];  ///< k_ray_setup: samples of this workgroup in the region; k_ray_bin: segment cursor
                          ///< (next free global position of the workgroup's reserved range)
};

/// Hash of a packed region key for the workgroups' LDS tables.  The global table's hashRegionKey multiplies 64-bit
/// values -- four quarter-rate 32-bit multiplies on gfx950, ~60 cycles of issue per look-up, and the binning kernels look
/// a region up for every ray-region segment.  Three full-rate 24-bit multiplies of the 16-bit coordinates do here: a
/// workgroup's regions are a compact neighbourhood, odd multipliers spread neighbours over the table.
__device__ inline uint32_t ltabHash(uint64_t key, uint32_t mask)
{
  const uint32_t lo = uint32_t(key);
  const uint32_t h = __umul24(lo & 0xffffu, 0x9E3Bu) ^ __umul24(lo >> 16, 0x85EBu) ^
                     __umul24(uint32_t(key >> 32) & 0xffffu, 0xC2B3u);
  return (h ^ (h >> 11)) & mask;
}

/// Find or insert `key`; returns the entry index or kLtabSize when the table is full (caller falls back to global).
/// `mask` = entries in use - 1 (a power of two <= kLtabSize: small workgroups use a small table so clearing and scanning
/// it does not dominate their run time).
template <uint32_t kTab>
__device__ inline uint32_t ltabFindOrInsert(LdsRegionTableT<kTab> &tab, uint64_t key, uint32_t mask)
{
  uint32_t idx = ltabHash(key, mask);
  for (uint32_t probe = 0; probe < 64; ++probe)
  {
    unsigned long long prev = tab.keys[idx];
    if (prev == 0)
    {
      prev = atomicCAS(&tab.keys[idx], 0ull, (unsigned long long)key);
    }
    if (prev == 0 || prev == key)
    {
      return idx;
    }
    idx = (idx + 1) & mask;
  }
  return kLtabSize;
}

template <uint32_t kTab>
__device__ inline uint32_t ltabFind(const LdsRegionTableT<kTab> &tab, uint64_t key, uint32_t mask)
{
  uint32_t idx = ltabHash(key, mask);
  for (uint32_t probe = 0; probe < 64; ++probe)
  {
    const unsigned long long k = tab.keys[idx];
    if (k == key)
    {
      return idx;
    }
    if (k == 0)
    {
      break;
    }
    idx = (idx + 1) & mask;
  }
  return kLtabSize;
}

/// Where a ray enters a region and how much of it lies inside (forEachSegment with resume state).
struct SegmentEntry
{
  uint32_t k0 = 0, k1 = 0, k2 = 0;  ///< steps taken per axis at the moment the region is entered
  uint32_t count = 0;               ///< voxels the ray visits inside the region
  double t_base = 0;                ///< time of the step that enters the region (0 for the ray's first segment)
  bool first = false;               ///< the segment starts at the ray's origin voxel
  bool end = false;                 ///< the segment's last voxel is the ray's end voxel (visited as part of the ray)
};

/// Walk the regions a ray crosses and call emit(region key, entry) for every ray-region segment which produces at
/// least one voxel visit.  With `with_resume_state` the entry is filled in: the per-axis step counts at the moment the
/// region is entered, the time of the entering step, the first / end flags and the number of voxels visited in the
/// region; otherwise it is empty (the counting passes only need the keys).  Segments in regions another replica owns
/// (MapConst::owner_world > 1) are skipped: every segment carries its own resume state, so dropping some of a ray's
/// segments does not change what the others do.
template <typename F>
__device__ inline void forEachSegment(const MapConst &mc, const RayWalk &rw, bool with_resume_state, F emit)
{
  auto f = [&](uint64_t key, const SegmentEntry &entry) {
    if (ownsRegion(mc, key))
    {
      emit(key, entry);
    }
  };
  if (!(rw.flags & kRwValid) || !(rw.flags & kRwWalk))
  {
    return;
  }
  const int manhattan = rw.total[0] + rw.total[1] + rw.total[2];
  const bool include_end = (rw.flags & kRwIncludeEnd) != 0;
  RegionCursor rc;
  regionCursorInit(mc, rw, rc);
  // A segment's voxel count is known once the NEXT region entry is: emission trails the enumeration by one.
  bool have = false;
  uint64_t p_key = 0;
  SegmentEntry p;
  int p_sum = 0;
  if (manhattan > 0 || include_end)
  {
    p_key = packRegionKey(rc.region[0], rc.region[1], rc.region[2]);
    p.first = true;
    have = true;
    if (!with_resume_state)
    {
      f(p_key, p);
    }
  }
  // Reciprocals of the step deltas for the step-count estimates (one division per axis per ray, not per crossing).
  double r0 = 0, r1 = 0, r2 = 0;
  if (with_resume_state)
  {
    r0 = 1.0 / rw.delta[0];
    r1 = 1.0 / rw.delta[1];
    r2 = 1.0 / rw.delta[2];
  }
  int axis, j;
  while (regionCursorNext(mc, rw, rc, axis, j))
  {
    // Entering a region at the ray's end voxel only produces work when the end voxel is part of the ray.
    if (stepReachesEnd(rw, axis, j) && !include_end)
    {
      continue;
    }
    const uint64_t key = packRegionKey(rc.region[0], rc.region[1], rc.region[2]);
    if (!with_resume_state)
    {
      f(key, p);
      continue;
    }
    const double ta = stepTime(sel3(axis, rw.init[0], rw.init[1], rw.init[2]),
                               sel3(axis, rw.delta[0], rw.delta[1], rw.delta[2]), j);
    const uint32_t rs0 = uint32_t((axis == 0) ? j : stepsBefore(rw.init[0], rw.delta[0], r0, rw.total[0], 0, axis, ta));
    const uint32_t rs1 = uint32_t((axis == 1) ? j : stepsBefore(rw.init[1], rw.delta[1], r1, rw.total[1], 1, axis, ta));
    const uint32_t rs2 = uint32_t((axis == 2) ? j : stepsBefore(rw.init[2], rw.delta[2], r2, rw.total[2], 2, axis, ta));
    const int sum = int(rs0 + rs1 + rs2);
    if (have)
    {
      p.count = uint32_t(sum - p_sum);
      f(p_key, p);
    }
    p_key = key;
    p.k0 = rs0;
    p.k1 = rs1;
    p.k2 = rs2;
    p.t_base = ta;
    p.first = false;
    p_sum = sum;
    have = true;
  }
  if (have && with_resume_state)
  {
    p.count = uint32_t(manhattan - p_sum) + (include_end ? 1u : 0u);
    p.end = include_end;
    f(p_key, p);
  }
}

/// Per-ray part of the fixed-point walk predictor (see Segment): the step deltas in fixed point, with the step direction
/// in the top bit, and whether the ray has to be walked exactly throughout.
struct RayFix
{
  uint32_t e0, e1, e2;
  bool poison;
};

__device__ inline RayFix rayFix(const MapConst &mc, const RayWalk &rw)
{
  RayFix rf;
  rf.poison = false;
  double t_last = 0;       // time of the ray's last real step
  double t_phantom = dInf();  // earliest step the predictor would take beyond an axis' last real one
  uint32_t e[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    e[a] = 0;
    if (rw.total[a] > 0)
    {
      const double ex = rw.delta[a] * mc.fix_scale;
      // (negated comparisons: NaN fails them and poisons the ray)
      rf.poison = rf.poison || !(ex >= 1.0);
      e[a] = !(ex >= 1.0) ? 0u : ((ex >= double(kFixMaxDelta)) ? kFixMaxDelta : uint32_t(ex));
      const double tl = stepTime(rw.init[a], rw.delta[a], rw.total[a]);
      const double tp = rw.init[a] + rw.delta[a] * double(rw.total[a]);
      // (a start point ON a voxel face -- every ray of a sensor standing on the voxel lattice -- can give a first
      // crossing time that rounds to -1e-18 instead of 0: anything above minus one predictor unit quantises to 0 within
      // the predictor's one-unit error per value, so only times below that poison the ray)
      rf.poison = rf.poison || !(tl * mc.fix_scale > -1.0) || !(tp >= tl);
      t_last = (tl > t_last) ? tl : t_last;
      t_phantom = (tp < t_phantom) ? tp : t_phantom;
    }
    e[a] |= rwSign(rw, a) ? kSegNegative : 0u;
  }
  // An axis which has taken its last step keeps stepping in the predictor.  That is harmless when every such phantom
  // step lies at or beyond the last real step of the ray (it then never leads a step the walk still needs by the trust
  // margin); where the end point's key and the geometry disagree (end points within the 1e-6 quantisation slack of a
  // region face) that can fail, and the ray is walked exactly.
  rf.poison = rf.poison || !(t_phantom >= t_last);
  rf.e0 = e[0];
  rf.e1 = e[1];
  rf.e2 = e[2];
  return rf;
}

/// Fixed-point time of the next step along one axis at a region entry (see Segment).  Returns false when the value
/// cannot be represented safely (negative or non-finite): the segment is then poisoned.
__device__ inline bool fixTime(const MapConst &mc, double init, double delta, int total, uint32_t k, double t_base,
                               uint32_t &f)
{
  if (int(k) >= total)
  {
    f = kFixFar;  // no steps left on this axis: time_next = inf (ohm/LineWalkCompute.h:299-301)
    return true;
  }
  const double t = (k == 0) ? init : init + delta * double(k);
  const double x = (t - t_base) * mc.fix_scale;
  const bool ok = x > -1.0;  // (false for NaN; (-1, 0) quantises to 0, still within one unit of the true time)
  f = !(x >= 1.0) ? 0u : ((x >= double(kFixFar)) ? kFixFar : uint32_t(x));
  return ok;
}

/// Build the walk kernel's record of one ray-region segment.
__device__ inline Segment makeSegment(const MapConst &mc, const RayWalk &rw, const RayFix &rf, uint32_t ray,
                                      const SegmentEntry &en)
{
  Segment sg;
  bool ok = !rf.poison;
  ok = fixTime(mc, rw.init[0], rw.delta[0], rw.total[0], en.k0, en.t_base, sg.f[0]) && ok;
  ok = fixTime(mc, rw.init[1], rw.delta[1], rw.total[1], en.k1, en.t_base, sg.f[1]) && ok;
  ok = fixTime(mc, rw.init[2], rw.delta[2], rw.total[2], en.k2, en.t_base, sg.f[2]) && ok;
  // An axis without steps left never moves: its delta is irrelevant, zero keeps the candidate at kFixFar.
  sg.e[0] = (int(en.k0) < rw.total[0]) ? rf.e0 : (rf.e0 & kSegNegative);
  sg.e[1] = (int(en.k1) < rw.total[1]) ? rf.e1 : (rf.e1 & kSegNegative);
  sg.e[2] = (int(en.k2) < rw.total[2]) ? rf.e2 : (rf.e2 & kSegNegative);
  if (!ok)
  {
    sg.f[0] = sg.f[1] = sg.f[2] = 0;
    sg.e[0] &= kSegNegative;
    sg.e[1] &= kSegNegative;
    sg.e[2] &= kSegNegative;
  }
  const int l0 = localCoord(rw.g0[0] + rwDir(rw, 0) * int(en.k0), mc.dim[0]);
  const int l1 = localCoord(rw.g0[1] + rwDir(rw, 1) * int(en.k1), mc.dim[1]);
  const int l2 = localCoord(rw.g0[2] + rwDir(rw, 2) * int(en.k2), mc.dim[2]);
  const uint32_t vi = uint32_t(l0 + l1 * mc.dim[0] + l2 * mc.dim[0] * mc.dim[1]);
  sg.vox = vi | (en.count << kSegVoxelBits);
  sg.ray = ray | ((en.first && (rw.flags & kRwExcludeStart)) ? kSegSkipFirst : 0u) | (en.end ? kSegEnd : 0u);
  return sg;
}

/// Region / voxel of the ray's sample (end) voxel.
__device__ inline void sampleVoxel(const MapConst &mc, const RayWalk &rw, uint64_t &region_key, uint32_t &vi)
{
  int r1[3], l1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    splitGlobal(rw.g0[a] + rwDir(rw, a) * rw.total[a], mc.dim[a], r1[a], l1[a]);
  }
  region_key = packRegionKey(r1[0], r1[1], r1[2]);
  vi = uint32_t(l1[0] + l1[1] * mc.dim[0] + l1[2] * mc.dim[0] * mc.dim[1]);
}

__device__ inline void markTouched(const BatchScratch &bs, uint32_t h)
{
  if (bs.touched_flag[h] == 0 && atomicExch(&bs.touched_flag[h], 1u) == 0)
  {
    const uint32_t t = atomicAdd(&bs.info->n_touched, 1u);
    bs.touched[t] = h;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ray order inside a binning workgroup.  A lane enumerates its ray's region crossings one after the other, so a wave
// runs as long as its longest ray: with rays in arrival order (a lidar's beams have unrelated ranges) a third of the
// lane slots idle.  The workgroup therefore visits its rays by descending extent (counting sort on the ray's Manhattan
// extent in regions, 64 bins, positions in LDS): the rays of one wave then cross about the same number of regions.
// The order is irrelevant to the result -- segment records and sample keys carry the ray index.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kRayOrderBins = 64;

struct RayOrder
{
  uint32_t bins[kRayOrderBins];
  uint16_t perm[kBinRaysPerBlock];
};

__device__ inline uint32_t rayExtentBin(const MapConst &mc, unsigned flags, const int total[3])
{
  if (!(flags & kRwValid) || !(flags & kRwWalk))
  {
    return 0;
  }
  const int max_dim = max(mc.dim[0], max(mc.dim[1], mc.dim[2]));
  const uint32_t manhattan = uint32_t(total[0] + total[1] + total[2]);
  return min(manhattan >> (31 - __clz(max_dim)), kRayOrderBins - 1u);
}

/// Build the order of the workgroup's `n_local` rays from their extent bins (bin_of(k) = bin of the thread's k-th ray,
/// local index threadIdx.x + k * blockDim.x; at most 8 rays per thread).  Ends with a barrier.
template <typename BinOf>
__device__ inline void buildRayOrder(RayOrder &order, uint32_t n_local, BinOf bin_of)
{
  if (threadIdx.x < kRayOrderBins)
  {
    order.bins[threadIdx.x] = 0;
  }
  __syncthreads();
  uint32_t my_bins[8];
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k)
  {
    const uint32_t idx = threadIdx.x + k * blockDim.x;
    my_bins[k] = 0;
    if (idx < n_local)
    {
      my_bins[k] = kRayOrderBins - 1u - bin_of(k, idx);  // descending extent
      atomicAdd(&order.bins[my_bins[k]], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < kRayOrderBins)
  {
    const uint32_t count = order.bins[threadIdx.x];
    uint32_t incl = count;
#pragma unroll
    for (int d = 1; d < int(kRayOrderBins); d <<= 1)
    {
      const uint32_t up = __shfl_up(incl, d);
      incl += (int(threadIdx.x) >= d) ? up : 0u;
    }
    order.bins[threadIdx.x] = incl - count;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k)
  {
    const uint32_t idx = threadIdx.x + k * blockDim.x;
    if (idx < n_local)
    {
      order.perm[atomicAdd(&order.bins[my_bins[k]], 1u)] = uint16_t(idx);
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// k_ray_setup: per-ray line-walk set-up + per-region segment counts.
// ---------------------------------------------------------------------------------------------------------------------
// (6 waves per SIMD: the kernel sits at 80-odd VGPRs, right at an allocation step -- 80 registers give a wave per SIMD
// more than 88 do, and the kernel is latency bound)
template <uint32_t kTab>
__global__ void __launch_bounds__(kBinThreads) __attribute__((amdgpu_waves_per_eu(6, 8)))
  k_ray_setup(MapConst mc, RegionTable rt, BatchScratch bs, const double *__restrict__ rays, uint32_t n_rays,
              unsigned ray_flags, RayWalk *__restrict__ walks, uint32_t rays_per_block, uint32_t tab_mask)
{
  __shared__ LdsRegionTableT<kTab> tab;
  __shared__ unsigned long long s_visits;
  __shared__ uint32_t s_rays_ok;
  __shared__ uint32_t s_list_n;
  for (uint32_t i = threadIdx.x; i <= tab_mask; i += blockDim.x)
  {
    tab.keys[i] = 0;
    tab.count[i] = 0;
    tab.cursor[i] = 0;
  }
  if (threadIdx.x == 0)
  {
    s_visits = 0;
    s_rays_ok = 0;
    s_list_n = 0;
  }
  __syncthreads();

  const uint32_t first = blockIdx.x * rays_per_block;
  const uint32_t last = min(first + rays_per_block, n_rays);
  unsigned long long my_visits = 0;
  uint32_t my_ok = 0;
  for (uint32_t ray = first + threadIdx.x; ray < last; ray += blockDim.x)
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
    if (mc.owner_world > 1u && (rw.flags & kRwApplySample))
    {
      // A sample in a region another replica owns is that replica's to apply.
      uint64_t sample_key;
      uint32_t sample_vi;
      sampleVoxel(mc, rw, sample_key, sample_vi);
      rw.flags &= ownsRegion(mc, sample_key) ? ~0u : ~unsigned(kRwApplySample);
    }
    walks[ray] = rw;
    my_ok += (rw.flags & kRwPassed) ? 1u : 0u;
    if (rw.flags & kRwBeyondTiles)
    {
      atomicAdd(&bs.info->n_beyond_tiles, 1u);  // (rare by construction: no LDS stage, no register kept for it)
    }
    if (!(rw.flags & kRwValid))
    {
      continue;
    }
    // Visit accounting: Manhattan extent == number of voxels the walk reports before the end voxel.
    const int manhattan = rw.total[0] + rw.total[1] + rw.total[2];
    if (rw.flags & kRwWalk)
    {
      my_visits += (unsigned long long)manhattan;
      my_visits -= ((rw.flags & kRwExcludeStart) && manhattan > 0) ? 1u : 0u;
      my_visits += (rw.flags & kRwIncludeEnd) ? 1u : 0u;
    }
    my_visits += (rw.flags & kRwApplySample) ? 1u : 0u;

    forEachSegment(mc, rw, false, [&](uint64_t key, const SegmentEntry &) {
      const uint32_t e = ltabFindOrInsert(tab, key, tab_mask);
      if (e < kLtabSize)
      {
        atomicAdd(&tab.count[e], 1u);
      }
      else
      {
        // LDS table full (very long rays): count straight in the global table.
        const uint32_t h = regionInsert(rt, key, &bs.info->error);
        atomicAdd(&bs.seg_count[h], 1u);
        markTouched(bs, h);
      }
    });
    if (rw.flags & kRwApplySample)
    {
      // The sample's region must exist (and be listed as touched) even when no segment enters it.
      uint64_t key;
      uint32_t vi;
      sampleVoxel(mc, rw, key, vi);
      const uint32_t e = ltabFindOrInsert(tab, key, tab_mask);
      if (e < kLtabSize)
      {
        atomicAdd(&tab.cursor[e], 1u);
      }
      else
      {
        const uint32_t h = regionInsert(rt, key, &bs.info->error);
        atomicAdd(&bs.hit_count[h], 1u);
        markTouched(bs, h);
      }
    }
  }
  atomicAdd(&s_visits, my_visits);
  atomicAdd(&s_rays_ok, my_ok);
  __syncthreads();
  if (threadIdx.x == 0)
  {
    atomicAdd(&bs.info->visits, s_visits);
    atomicAdd(&bs.info->rays_ok, (unsigned long long)s_rays_ok);
  }
  // One global insert + one counter atomic per (workgroup, region).
  for (uint32_t e = threadIdx.x; e <= tab_mask; e += blockDim.x)
  {
    const unsigned long long key = tab.keys[e];
    if (key)
    {
      const uint32_t h = regionInsert(rt, key, &bs.info->error);
      const uint32_t c = tab.count[e];
      if (c)
      {
        atomicAdd(&bs.seg_count[h], c);
      }
      markTouched(bs, h);
      const uint32_t hits = tab.cursor[e];
      if (hits)
      {
        atomicAdd(&bs.hit_count[h], hits);
      }
      WgRegion wr;
      wr.key = key;
      wr.count = c;
      wr.entry = e;
      wr.hash = h;
      wr.hits = hits;
      bs.wg_regions[size_t(blockIdx.x) * kLtabSize + atomicAdd(&s_list_n, 1u)] = wr;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    bs.wg_region_count[blockIdx.x] = s_list_n;
  }
}

/// A region's use history, two words per pool slot: [0] the stamp of the batch that used it last, [1] the stamp of its
/// last use BEFORE the current run of consecutive batches (0: none).  last - before = the period a region comes back with
/// (a sweep that revisits it every N batches), which is what the spill policy predicts its next use from.
__device__ inline void touchRegionUse(uint32_t *use, uint32_t slot, uint32_t stamp)
{
  const uint32_t last = use[2 * size_t(slot)];
  if (last != stamp)
  {
    if (last != 0 && last + 1u != stamp)
    {
      use[2 * size_t(slot) + 1] = last;  // back after a gap
    }
    use[2 * size_t(slot)] = stamp;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_plan: one block.  Exclusive scan of per-region segment counts over the touched list; build chunk list.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
  k_plan(RegionTable rt, BatchScratch bs, Chunk *__restrict__ chunks, uint32_t chunk_capacity,
         uint32_t chunk_segments, BatchInfo *__restrict__ host_info, BatchInfo *__restrict__ next_info,
         uint32_t *__restrict__ event_count, uint32_t inline_max_hits)
{
  constexpr uint32_t kSizeClasses = 32;  // chunk size classes for the largest-first order (class = 32 * size / max)
  constexpr int kRounds = 4;             // touched regions held in registers between the two passes: 4 x 1024
  constexpr uint32_t kBigQueue = 256;
  __shared__ uint32_t s_seg[16];
  __shared__ uint32_t s_chk[16];
  __shared__ uint32_t s_hit[16];
  __shared__ uint32_t s_seg_base;
  __shared__ uint32_t s_chk_base;
  __shared__ uint32_t s_hit_base;
  __shared__ uint32_t s_hit_max;
  __shared__ uint32_t s_class[kSizeClasses + 1];
  __shared__ uint32_t s_hclass[33];  // regions per sample-count class (class = bits of count - 1)
  __shared__ uint32_t s_big_n;
  __shared__ uint32_t s_big[kBigQueue][4];  // hash index, slot, segment offset, segment count
  __shared__ uint32_t s_n_apply_counts, s_n_apply_hits;
  const uint32_t n = bs.info->n_touched;
  const uint32_t tid = threadIdx.x;
  if (tid == 0)
  {
    s_seg_base = 0;
    s_chk_base = 0;
    s_hit_base = 0;
    s_hit_max = 0;
    s_big_n = 0;
    s_n_apply_counts = 0;
    s_n_apply_hits = 0;
  }
  if (tid <= kSizeClasses)
  {
    s_class[tid] = 0;
    s_hclass[tid] = 0;
  }
  __syncthreads();
  auto hitClass = [](uint32_t hits) { return uint32_t(32 - __clz(int(hits - 1u))) & 31u; };
  auto sizeClass = [&](uint32_t size) { return min(size * kSizeClasses / chunk_segments, kSizeClasses); };
  auto emit = [&](uint32_t h, uint32_t slot, uint32_t seg_excl, uint32_t cnt, uint32_t nchk, uint32_t per, uint32_t c) {
    const uint32_t begin = c * per;
    const uint32_t end = min(cnt, (c + 1) * per);
    const uint32_t pos = atomicAdd(&s_class[sizeClass(end - begin)], 1u);
    if (pos < chunk_capacity)
    {
      Chunk ch;
      ch.slot = slot;
      ch.seg_begin = seg_excl + begin;
      ch.seg_end = seg_excl + end;
      ch.hash_index = h | ((nchk == 1) ? 0x80000000u : 0u);
      chunks[pos] = ch;
    }
  };

  // Pass 1: segment offsets per region (prefix sum in touched order) and the chunk size histogram.  A region with
  // more than chunk_segments segments is split into equal chunks: nchk - 1 of `per` segments and a last one with the
  // rest.  The first kRounds x 1024 regions stay in registers for pass 2.
  uint32_t r_h[kRounds], r_cnt[kRounds], r_slot[kRounds], r_off[kRounds], r_hits[kRounds];
  for (uint32_t base = 0, round = 0; base < n; base += 1024, ++round)
  {
    const uint32_t i = base + tid;
    uint32_t h = 0, cnt = 0, nchk = 0, slot = 0, hits = 0;
    if (i < n)
    {
      h = bs.touched[i];
      cnt = bs.seg_count[h];
      hits = bs.hit_count[h];
      slot = rt.vals[h];
      nchk = (cnt + chunk_segments - 1) / chunk_segments;
      if (hits)
      {
        atomicMax(&s_hit_max, hits);
        atomicAdd(&s_hclass[hitClass(hits)], 1u);
      }
      // What the walk kernel leaves to the apply kernels (inline_max_hits: the most samples the walk replays for a
      // region it holds in one chunk; 0: it replays none): the counts of regions cut into several chunks, and the samples
      // of regions that are not held by exactly one chunk or are too dense.
      if (nchk > 1)
      {
        bs.apply_counts_list[atomicAdd(&s_n_apply_counts, 1u)] = h;
      }
      if (hits && (nchk != 1 || hits > inline_max_hits))
      {
        bs.apply_hits_list[atomicAdd(&s_n_apply_hits, 1u)] = h;
      }
    }
    // Inclusive scan of (segments, chunks, samples) over the 1024 threads: shuffles inside a wave, wave totals
    // through LDS.
    uint32_t inc_seg = cnt, inc_chk = nchk, inc_hit = hits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t a = __shfl_up(inc_seg, d);
      const uint32_t b = __shfl_up(inc_chk, d);
      const uint32_t c = __shfl_up(inc_hit, d);
      if (int(tid & 63u) >= d)
      {
        inc_seg += a;
        inc_chk += b;
        inc_hit += c;
      }
    }
    if ((tid & 63u) == 63u)
    {
      s_seg[tid >> 6] = inc_seg;
      s_chk[tid >> 6] = inc_chk;
      s_hit[tid >> 6] = inc_hit;
    }
    __syncthreads();
    uint32_t wave_seg = 0, wave_hit = 0, all_seg = 0, all_chk = 0, all_hit = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w)
    {
      const uint32_t a = s_seg[w];
      const uint32_t c = s_hit[w];
      wave_seg += (w < (tid >> 6)) ? a : 0u;
      wave_hit += (w < (tid >> 6)) ? c : 0u;
      all_seg += a;
      all_chk += s_chk[w];
      all_hit += c;
    }
    const uint32_t seg_excl = s_seg_base + wave_seg + inc_seg - cnt;
    const uint32_t hit_excl = s_hit_base + wave_hit + inc_hit - hits;
    if (i < n)
    {
      bs.seg_offset[h] = seg_excl;
      bs.seg_cursor[h] = 0;
      if (slot < rt.slot_capacity)
      {
        // The region's range in the sample list; hit_end doubles as the scatter cursor of k_ray_bin and ends up at
        // hit_begin + samples.
        bs.hit_begin[slot] = hit_excl;
        bs.hit_end[slot] = hit_excl;
        // use history (spill to host): also stamped by an attempt the host goes on to reject -- that is what makes
        // the eviction which follows spare the regions the repeated batch needs
        touchRegionUse(bs.last_use, slot, bs.stamp);
      }
      if (nchk)
      {
        const uint32_t per = (cnt + nchk - 1) / nchk;
        if (nchk > 1)
        {
          atomicAdd(&s_class[sizeClass(per)], nchk - 1);
        }
        atomicAdd(&s_class[sizeClass(cnt - (nchk - 1) * per)], 1u);
      }
    }
#pragma unroll
    for (int k = 0; k < kRounds; ++k)
    {
      if (uint32_t(k) == round)
      {
        r_h[k] = h;
        r_cnt[k] = cnt;
        r_slot[k] = slot;
        r_off[k] = seg_excl;
        r_hits[k] = hits;
      }
    }
    __syncthreads();
    if (tid == 1023)
    {
      s_seg_base += all_seg;
      s_chk_base += all_chk;
      s_hit_base += all_hit;
    }
    __syncthreads();
  }
  // Chunks are emitted largest class first: the persistent walk workgroups take them in this order, so the launch
  // ends on its smallest chunks.
  if (tid == 0)
  {
    uint32_t run = 0;
    for (int c = int(kSizeClasses); c >= 0; --c)
    {
      const uint32_t count = s_class[c];
      s_class[c] = run;
      run += count;
    }
  }
  if (tid == 64)
  {
    uint32_t run = 0;
    for (int c = 31; c >= 0; --c)
    {
      const uint32_t count = s_hclass[c];
      s_hclass[c] = run;
      run += count;
    }
    s_hclass[32] = run;
  }
  __syncthreads();
  // Pass 2: emit the chunk records.  Regions with a few chunks are written by their own thread; the big ones (the
  // regions around the sensor split into hundreds of chunks) are queued and written by the whole workgroup.
  for (uint32_t base = 0, round = 0; base < n; base += 1024, ++round)
  {
    const uint32_t i = base + tid;
    uint32_t h = 0, cnt = 0, slot = 0, seg_excl = 0, hits = 0;
#pragma unroll
    for (int k = 0; k < kRounds; ++k)
    {
      if (uint32_t(k) == round)
      {
        h = r_h[k];
        cnt = r_cnt[k];
        slot = r_slot[k];
        seg_excl = r_off[k];
        hits = r_hits[k];
      }
    }
    if (round >= uint32_t(kRounds) && i < n)
    {
      h = bs.touched[i];
      cnt = bs.seg_count[h];
      slot = rt.vals[h];
      seg_excl = bs.seg_offset[h];
      hits = bs.hit_count[h];
    }
    if (i < n && hits)
    {
      bs.hit_count[h] = 0;
      bs.sort_list[atomicAdd(&s_hclass[hitClass(hits)], 1u)] = h;
    }
    const uint32_t nchk = (i < n) ? (cnt + chunk_segments - 1) / chunk_segments : 0u;
    if (nchk)
    {
      const uint32_t per = (cnt + nchk - 1) / nchk;
      uint32_t q = kBigQueue;
      if (nchk > 4)
      {
        q = atomicAdd(&s_big_n, 1u);
      }
      if (q < kBigQueue)
      {
        s_big[q][0] = h;
        s_big[q][1] = slot;
        s_big[q][2] = seg_excl;
        s_big[q][3] = cnt;
      }
      else
      {
        for (uint32_t c = 0; c < nchk; ++c)
        {
          emit(h, slot, seg_excl, cnt, nchk, per, c);
        }
      }
    }
  }
  __syncthreads();
  const uint32_t n_big = min(s_big_n, kBigQueue);
  // One wave per queued region, one lane per chunk.
  for (uint32_t q = tid >> 6; q < n_big; q += 16)
  {
    const uint32_t cnt = s_big[q][3];
    const uint32_t nchk = (cnt + chunk_segments - 1) / chunk_segments;
    const uint32_t per = (cnt + nchk - 1) / nchk;
    for (uint32_t c = tid & 63u; c < nchk; c += 64)
    {
      emit(s_big[q][0], s_big[q][1], s_big[q][2], cnt, nchk, per, c);
    }
  }
  // "Modified" flags (bit 0: since the last syncVoxels(), bit 1: since the last replica merge): only by a batch the
  // host is going to run -- one it rejects (pool / chunk list exhausted) is rolled back and must leave no trace here
  // (the header promises a failing batch leaves the map as it was).
  if (!(bs.info->error & (kErrSlotsFull | kErrHashFull)) && *rt.n_slots <= rt.slot_capacity && s_chk_base <= chunk_capacity)
  {
    for (uint32_t i = tid; i < n; i += 1024)
    {
      const uint32_t slot = rt.vals[bs.touched[i]];
      if (slot < rt.slot_capacity)
      {
        bs.dirty[slot] |= 3u;
      }
    }
  }
  if (tid == 0)
  {
    BatchInfo out = *bs.info;  // visits, rays, touched regions, errors: accumulated by k_ray_setup
    out.n_segments = s_seg_base;
    out.n_chunks = s_chk_base;
    out.n_slots = *rt.n_slots;
    out.n_hits = s_hit_base;
    out.max_region_hits = s_hit_max;
    out.n_hit_regions = s_hclass[32];
    out.n_apply_counts = s_n_apply_counts;
    out.n_apply_hits = s_n_apply_hits;
    *bs.info = out;
    // Housekeeping that would otherwise be separate copy / fill launches on the batch's critical path: the host's copy
    // of the summary goes straight to pinned memory, the next batch's summary and the walk's counters start at zero.
    *host_info = out;
    *next_info = BatchInfo{};
    event_count[0] = 0;  // deferred events
    event_count[1] = 0;  // walk chunk cursor
    event_count[3] = 0;  // traversal pass' chunk cursor (the stop-flag replay clears the word again before it uses it)
    __threadfence_system();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_ray_bin: scatter segments to region buckets, emit sample sort keys, set the per-region sample bitmask.
// Three steps per workgroup: count its segments per region in LDS, reserve one contiguous range per region with a
// single returning atomic, then re-enumerate and scatter through LDS cursors.
// ---------------------------------------------------------------------------------------------------------------------
template <uint32_t kTab>
__global__ void __launch_bounds__(kBinThreads)
  k_ray_bin(MapConst mc, RegionTable rt, BatchScratch bs, const RayWalk *__restrict__ walks, uint32_t n_rays,
            Segment *__restrict__ segments, uint32_t segment_capacity, unsigned long long *__restrict__ hit_keys,
            uint32_t *__restrict__ hit_mask, int ray_shift, int bucket_hits, uint32_t rays_per_block,
            uint32_t tab_mask)
{
  __shared__ LdsRegionTableT<kTab> tab;
  __shared__ uint32_t s_slot[kTab];  // region slot per table entry (kSlotUnassigned: not handed over by the set-up)
  if (bs.info->error & (kErrHashFull | kErrSlotsFull))
  {
    return;  // the batch's set-up overflowed the pool: the host grows it and repeats the batch
  }
  for (uint32_t i = threadIdx.x; i <= tab_mask; i += blockDim.x)
  {
    tab.keys[i] = 0;
    tab.count[i] = 0;
    tab.cursor[i] = 0;
    s_slot[i] = kSlotUnassigned;
  }
  __syncthreads();

  const uint32_t first = blockIdx.x * rays_per_block;
  const uint32_t last = min(first + rays_per_block, n_rays);
  const uint32_t mask_words = uint32_t(mc.region_voxels + 31) >> 5;

  // Step 1: the workgroup's regions with their segment and sample counts come from k_ray_setup (same rays, same LDS
  // table layout); reserve one contiguous range in every region bucket it feeds.
  const uint32_t n_wg_regions = bs.wg_region_count[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < n_wg_regions; i += blockDim.x)
  {
    const WgRegion wr = bs.wg_regions[size_t(blockIdx.x) * kLtabSize + i];
    tab.keys[wr.entry] = wr.key;
    if (wr.count)
    {
      tab.cursor[wr.entry] = bs.seg_offset[wr.hash] + atomicAdd(&bs.seg_cursor[wr.hash], wr.count);
    }
    if (wr.hits)
    {
      // The samples' keys need the region's slot: looked up once per (workgroup, region) here, not once per ray.
      const uint32_t slot = rt.vals[wr.hash];
      s_slot[wr.entry] = slot;
      if (bucket_hits && slot < rt.slot_capacity)  // (a speculatively launched pass may see a batch whose slots ran out)
      {
        tab.count[wr.entry] = atomicAdd(&bs.hit_end[slot], wr.hits);
      }
    }
  }
  __syncthreads();
  // Step 2: sample keys and mask bits.  bucket_hits: the keys go straight into their region's range of the sample
  // list (k_sort_region_hits orders each range); otherwise they are written in ray order for a device-wide sort.
  auto emitSample = [&](uint32_t ray, const RayWalk &rw) {
    unsigned long long hk = kHitInvalid;
    uint32_t pos = ray;
    if ((rw.flags & kRwValid) && (rw.flags & kRwApplySample))
    {
      uint64_t key;
      uint32_t vi;
      sampleVoxel(mc, rw, key, vi);
      const uint32_t e = ltabFind(tab, key, tab_mask);
      uint32_t slot;
      if (e < kLtabSize)
      {
        // (A speculatively launched pass may run on a batch whose region inserts failed -- hash table full: the slot is
        // then unassigned, the host repeats the batch after growing the pool, nothing may be written here.)
        slot = s_slot[e];
        if (bucket_hits)
        {
          pos = atomicAdd(&tab.count[e], 1u);
        }
      }
      else
      {
        const uint32_t h = regionFind(rt, key);
        slot = (h != 0xffffffffu) ? rt.vals[h] : kSlotUnassigned;
        if (bucket_hits && slot < rt.slot_capacity)
        {
          pos = atomicAdd(&bs.hit_end[slot], 1u);  // LDS table overflow in k_ray_setup: counted globally there too
        }
      }
      if (slot < rt.slot_capacity)
      {
        // ray_shift == 1 (NDT / TSDF event streams): the low bit tags the key as a sample (hit) event.
        hk = ((unsigned long long)slot << kHitSlotShift) | ((unsigned long long)vi << kHitRayBits) |
             ((unsigned long long)ray << ray_shift) | (unsigned long long)(ray_shift ? 1u : 0u);
        atomicOr(&hit_mask[size_t(slot) * mask_words + (vi >> 5)], 1u << (vi & 31));
      }
    }
    if (!bucket_hits || hk != kHitInvalid)
    {
      hit_keys[pos] = hk;
    }
  };
  // Step 3: scatter, rays visited by descending extent (see RayOrder).
  __shared__ RayOrder order;
  const uint32_t n_local = last - first;
  buildRayOrder(order, n_local, [&](uint32_t, uint32_t idx) {
    const RayWalk *w = walks + first + idx;
    const int total[3] = { w->total[0], w->total[1], w->total[2] };
    return rayExtentBin(mc, w->flags, total);
  });
  for (uint32_t idx = threadIdx.x; idx < n_local; idx += blockDim.x)
  {
    const uint32_t ray = first + order.perm[idx];
    const RayWalk rw = walks[ray];
    emitSample(ray, rw);  // (the ray's record is loaded once for both its sample key and its segments)
    const RayFix rf = rayFix(mc, rw);
    forEachSegment(mc, rw, true, [&](uint64_t key, const SegmentEntry &entry) {
      const uint32_t e = ltabFind(tab, key, tab_mask);
      uint32_t pos;
      if (e < kLtabSize)
      {
        pos = atomicAdd(&tab.cursor[e], 1u);
      }
      else
      {
        const uint32_t h = regionFind(rt, key);  // table overflow: straight to the global cursor
        if (h == 0xffffffffu)
        {
          return;  // (failed insert of a speculated batch, see above)
        }
        pos = bs.seg_offset[h] + atomicAdd(&bs.seg_cursor[h], 1u);
      }
      if (pos < segment_capacity)
      {
        segments[pos] = makeSegment(mc, rw, rf, ray, entry);
      }
    });
  }
}

/// The number of rays of [rays, rays + n) the map's ray filter passes: what an integrate call reports for its own
/// rays when they only run later, with a collected batch (one workgroup; the count goes straight to pinned memory).
__global__ void __launch_bounds__(1024)
  k_count_passed(MapConst mc, const double *__restrict__ rays, uint32_t n_rays, unsigned ray_flags,
                 uint32_t *__restrict__ host_count)
{
  __shared__ uint32_t s_ok;
  if (threadIdx.x == 0)
  {
    s_ok = 0;
  }
  __syncthreads();
  uint32_t my_ok = 0;
  for (uint32_t ray = threadIdx.x; ray < n_rays; ray += blockDim.x)
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
    my_ok += (rw.flags & kRwPassed) ? 1u : 0u;
  }
  atomicAdd(&s_ok, my_ok);
  __syncthreads();
  if (threadIdx.x == 0)
  {
    *host_count = s_ok;
  }
}

/// Rewrite only the sample (hit) keys of a batch (used when the NDT / TSDF key buffer had to be re-allocated).
__global__ void __launch_bounds__(256)
  k_rekey_samples(MapConst mc, RegionTable rt, const RayWalk *__restrict__ walks, uint32_t n_rays,
                  unsigned long long *__restrict__ hit_keys, int ray_shift)
{
  const uint32_t ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays)
  {
    return;
  }
  const RayWalk rw = walks[ray];
  unsigned long long hk = kHitInvalid;
  if ((rw.flags & kRwValid) && (rw.flags & kRwApplySample))
  {
    int r1[3], l1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      splitGlobal(rw.g0[a] + rwDir(rw, a) * rw.total[a], mc.dim[a], r1[a], l1[a]);
    }
    const uint32_t h = regionFind(rt, packRegionKey(r1[0], r1[1], r1[2]));
    const uint32_t slot = (h != 0xffffffffu) ? rt.vals[h] : kSlotUnassigned;
    if (slot < rt.slot_capacity)
    {
      const uint32_t vi = uint32_t(l1[0] + l1[1] * mc.dim[0] + l1[2] * mc.dim[0] * mc.dim[1]);
      hk = ((unsigned long long)slot << kHitSlotShift) | ((unsigned long long)vi << kHitRayBits) |
           ((unsigned long long)ray << ray_shift) | (unsigned long long)(ray_shift ? 1u : 0u);
    }
  }
  hit_keys[ray] = hk;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_hit_bounds: [begin, end) of each region slot in the sorted hit list.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
  k_hit_bounds(const unsigned long long *__restrict__ sorted, BatchScratch bs, int region_voxels)
{
  const uint32_t n_hits = bs.info->n_hits;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_hits)
  {
    return;
  }
  const unsigned long long group = sorted[i] >> kHitRayBits;
  if (i == 0 || (sorted[i - 1] >> kHitRayBits) != group)
  {
    // First sample of its voxel: entry point for ordering misses against this voxel's samples.  Entries are only
    // ever read for voxels whose mask bit is set in the same batch, so the table needs no clearing.
    const uint32_t slot = uint32_t(group >> kHitVoxelBits);
    const uint32_t vi = uint32_t(group) & ((1u << kHitVoxelBits) - 1u);
    bs.voxel_first_hit[size_t(slot) * size_t(region_voxels) + vi] = i;
  }
  const uint32_t slot = uint32_t(sorted[i] >> kHitSlotShift);
  if (i == 0 || uint32_t(sorted[i - 1] >> kHitSlotShift) != slot)
  {
    bs.hit_begin[slot] = i;
  }
  if (i + 1 == n_hits || uint32_t(sorted[i + 1] >> kHitSlotShift) != slot)
  {
    bs.hit_end[slot] = i + 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_sort_region_hits: one workgroup per touched region orders the region's samples by (voxel, ray) in LDS (bitonic
// network over the next power of two) and records each voxel's first sample.  Replaces a device-wide radix sort of all
// sample keys plus k_hit_bounds when no region holds more than kSortRegionHits samples.
// ---------------------------------------------------------------------------------------------------------------------
/// LDS position of sort element i: one pad element per 32 keeps the power-of-two strides of the network (8 consecutive
/// keys per lane in the last trip of every merge level) off a single group of banks.
__device__ inline uint32_t sortSlot(uint32_t i)
{
  return i + (i >> 5);
}

/// R fused bitonic stages on the 2^R elements they connect (register butterflies between one LDS read and write).
template <int R>
__device__ inline void bitonicFused(unsigned long long *l_keys, uint32_t g, uint32_t s_shift, uint32_t k)
{
  constexpr uint32_t kCount = 1u << R;
  const uint32_t low = g & ((1u << s_shift) - 1u);
  const uint32_t base = ((g >> s_shift) << (s_shift + R)) | low;
  const bool ascending = (base & k) == 0;
  unsigned long long v[kCount];
#pragma unroll
  for (uint32_t m = 0; m < kCount; ++m)
  {
    v[m] = l_keys[sortSlot(base | (m << s_shift))];
  }
#pragma unroll
  for (int t = 0; t < R; ++t)
  {
    const uint32_t d = 1u << (R - 1 - t);
#pragma unroll
    for (uint32_t m = 0; m < kCount; ++m)
    {
      if ((m & d) == 0)
      {
        const unsigned long long a = v[m];
        const unsigned long long c = v[m | d];
        const bool swap = (a > c) == ascending;
        v[m] = swap ? c : a;
        v[m | d] = swap ? a : c;
      }
    }
  }
#pragma unroll
  for (uint32_t m = 0; m < kCount; ++m)
  {
    l_keys[sortSlot(base | (m << s_shift))] = v[m];
  }
}

constexpr uint32_t kSortRegionHits = 8192;
constexpr int kSortThreads = 1024;
#ifndef OHMHIP_SORT_SMALL
#define OHMHIP_SORT_SMALL 2048  // regions with at most this many samples are ordered by 256-thread workgroups (0: off)
#endif
constexpr uint32_t kSortSmallHits = OHMHIP_SORT_SMALL;
constexpr int kSortSmallThreads = 256;

/// kCap / kThreads: the instantiation's LDS capacity in keys and workgroup size; it orders the regions of the list with
/// min_hits < samples <= kCap.  The network of a region with ~10^3 samples keeps 128 lanes busy per fused stage: the
/// 1024-thread, 66 KiB instantiation (two workgroups per CU) spends its time in barriers of mostly idle waves, so regions
/// of at most kSortSmallHits samples -- nearly all of them -- go to a 256-thread, 17 KiB one that runs eight per CU.
template <uint32_t kCap, int kThreads>
__global__ void __launch_bounds__(kThreads)
  k_sort_region_hits(RegionTable rt, BatchScratch bs, const unsigned long long *__restrict__ keys,
                     unsigned long long *__restrict__ sorted, int region_voxels, uint32_t min_hits)
{
  __shared__ unsigned long long l_keys[kCap + kCap / 32];
  // Grid-stride over the list (its length lives on the device: the launch may be issued before the host knows it).
  const uint32_t n_regions = bs.info->n_hit_regions;
  for (uint32_t list_index = blockIdx.x; list_index < n_regions; list_index += gridDim.x)
  {
  const uint32_t h = bs.sort_list[list_index];
  const uint32_t slot = rt.vals[h];
  if (slot >= rt.slot_capacity)
  {
    continue;
  }
  const uint32_t begin = bs.hit_begin[slot];
  const uint32_t n = bs.hit_end[slot] - begin;
  if (n <= min_hits || n > kCap)
  {
    continue;
  }
  uint32_t padded = 64;
  while (padded < n)
  {
    padded <<= 1;
  }
  for (uint32_t i = threadIdx.x; i < padded; i += kThreads)
  {
    l_keys[sortSlot(i)] = (i < n) ? keys[begin + i] : ~0ull;
  }
  __syncthreads();
  // Bitonic network, up to three consecutive compare distances (j, j/2, j/4) fused per LDS round trip: a thread pulls
  // the 8 (4, 2) elements those stages connect into registers, runs the butterflies there and writes them back.
  // The network is LDS-bandwidth bound, so this cuts its cost by the same factor as the traffic (~2.6x).
  for (uint32_t k = 2; k <= padded; k <<= 1)
  {
    uint32_t j = k >> 1;
    while (j > 0)
    {
      // levels fused this trip: r in 1..3, distances j, j/2, .., s = j >> (r - 1)
      const uint32_t levels_left = uint32_t(32 - __clz(int(j)));  // log2(j) + 1
      const uint32_t r = min(3u, levels_left);
      const uint32_t s_shift = levels_left - r;  // log2 of the smallest distance s
      const uint32_t group_count = padded >> r;
      for (uint32_t g = threadIdx.x; g < group_count; g += kThreads)
      {
        if (r == 3)
        {
          bitonicFused<3>(l_keys, g, s_shift, k);
        }
        else if (r == 2)
        {
          bitonicFused<2>(l_keys, g, s_shift, k);
        }
        else
        {
          bitonicFused<1>(l_keys, g, s_shift, k);
        }
      }
      __syncthreads();
      j >>= r;
    }
  }
  for (uint32_t i = threadIdx.x; i < n; i += kThreads)
  {
    const unsigned long long key = l_keys[sortSlot(i)];
    sorted[begin + i] = key;
    if (i == 0 || (l_keys[sortSlot(i - 1)] >> kHitRayBits) != (key >> kHitRayBits))
    {
      // First sample of its voxel: entry point for ordering misses against this voxel's samples.
      const uint32_t vi = uint32_t(key >> kHitRayBits) & ((1u << kHitVoxelBits) - 1u);
      bs.voxel_first_hit[size_t(slot) * size_t(region_voxels) + vi] = begin + i;
    }
  }
  __syncthreads();  // l_keys is reused by the next region
  }  // regions
}

/// Undo the cursor movement of a k_ray_bin pass over the touched regions (segment cursors back to zero, sample cursors
/// back to the start of the region's range) so the pass can be repeated with other launch parameters.
__global__ void __launch_bounds__(256) k_reset_cursors(RegionTable rt, BatchScratch bs)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs.info->n_touched)
  {
    return;
  }
  const uint32_t h = bs.touched[i];
  bs.seg_cursor[h] = 0;
  const uint32_t slot = rt.vals[h];
  if (slot < rt.slot_capacity)
  {
    bs.hit_end[slot] = bs.hit_begin[slot];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Occupancy update functions (bit-for-bit the CPU mapper's per-voxel arithmetic).
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline float fInf()
{
  return __int_as_float(0x7f800000);
}

/// One miss: ohm/RayMapperOccupancy.cpp:143-163 + ohm/VoxelOccupancyCompute.h:110-120 (null_update == false).
__device__ inline float occMiss(const MapConst &mc, unsigned ray_flags, float initial)
{
  const float inf = fInf();
  const bool unobserved = initial == inf;
  const bool is_free = !unobserved && initial < mc.threshold_value;
  const bool is_occ = !unobserved && initial >= mc.threshold_value;
  float adj = mc.miss_value;
  adj = (unobserved && (ray_flags & OHMHIP_RF_EXCLUDE_UNOBSERVED)) ? inf : adj;
  adj = (is_free && (ray_flags & OHMHIP_RF_EXCLUDE_FREE)) ? 0.0f : adj;
  adj = (is_occ && (ray_flags & OHMHIP_RF_EXCLUDE_OCCUPIED)) ? 0.0f : adj;
  const float base = unobserved ? 0.0f : initial;
  adj = (unobserved || (mc.sat_min < initial && initial < mc.sat_max)) ? adj : 0.0f;
  return (base != inf) ? fmaxf(mc.min_value, base + adj) : base;
}

/// One hit: ohm/RayMapperOccupancy.cpp:261-281 + ohm/VoxelOccupancyCompute.h:44-54.
__device__ inline float occHit(const MapConst &mc, unsigned ray_flags, float initial)
{
  const float inf = fInf();
  const bool unobserved = initial == inf;
  const bool is_free = !unobserved && initial < mc.threshold_value;
  const bool is_occ = !unobserved && initial >= mc.threshold_value;
  float adj = mc.hit_value;
  adj = (unobserved && (ray_flags & OHMHIP_RF_EXCLUDE_UNOBSERVED)) ? inf : adj;
  adj = (is_free && (ray_flags & OHMHIP_RF_EXCLUDE_FREE)) ? 0.0f : adj;
  adj = (is_occ && (ray_flags & OHMHIP_RF_EXCLUDE_OCCUPIED)) ? 0.0f : adj;
  const float base = unobserved ? 0.0f : initial;
  adj = (unobserved || (mc.sat_min < initial && initial < mc.sat_max)) ? adj : 0.0f;
  return (base != inf) ? fminf(base + adj, mc.max_value) : base;
}

/// n sequential misses.  The update is a deterministic function of the value alone, so once it reaches a fixed point
/// (the min clamp) the remaining applications are the identity and can be skipped without changing the result.
__device__ inline float occMissN(const MapConst &mc, unsigned ray_flags, float x, uint32_t n)
{
  constexpr unsigned kExcludeFlags = OHMHIP_RF_EXCLUDE_UNOBSERVED | OHMHIP_RF_EXCLUDE_FREE | OHMHIP_RF_EXCLUDE_OCCUPIED;
  if (n == 0)
  {
    return x;
  }
  if (!(ray_flags & kExcludeFlags))
  {
    // Without the exclusion flags only the FIRST miss can meet an unobserved voxel; every later one is occMiss() of an
    // observed value, which is these three operations (same operations, same order: bit identical) -- a voxel that is
    // not yet at the clamp pays them up to ~10 times per batch (a map the sensor is moving through).
    // The first miss is occMiss() with the exclusion flags known to be clear: its three flag selects and the free / occupied
    // classification drop out, the remaining operations are the same in the same order (bit identical; -3 us per C1 batch).
    const bool unobserved = x == fInf();
    const float base = unobserved ? 0.0f : x;
    const float first_adj = (unobserved || (mc.sat_min < x && x < mc.sat_max)) ? mc.miss_value : 0.0f;
    float nx = fmaxf(mc.min_value, base + first_adj);
    if (nx == x)
    {
      return x;
    }
    x = nx;
    for (uint32_t k = 1; k < n; ++k)
    {
      const float adj = (mc.sat_min < x && x < mc.sat_max) ? mc.miss_value : 0.0f;
      nx = fmaxf(mc.min_value, x + adj);
      if (nx == x)
      {
        break;
      }
      x = nx;
    }
    return x;
  }
  for (uint32_t k = 0; k < n; ++k)
  {
    const float nx = occMiss(mc, ray_flags, x);
    if (nx == x)
    {
      break;
    }
    x = nx;
  }
  return x;
}

/// ohm/VoxelMeanCompute.h:134-152 with Vec3 = dvec3, coord_real = double (as the CPU mappers instantiate it).
__device__ inline uint32_t subVoxelUpdate(uint32_t coord, uint32_t point_count, const double v[3], double resolution)
{
  const int mean_positions = (1 << 10) - 1;
  const double mean_resolution = resolution / double(mean_positions);
  const double offset = double(0.5f) * resolution;
  double mean[3];
  mean[0] = int(coord & mean_positions) * mean_resolution - offset;
  mean[1] = int((coord >> 10) & mean_positions) * mean_resolution - offset;
  mean[2] = int((coord >> 20) & mean_positions) * mean_resolution - offset;
  const double one_on_count_plus_one = double(1) / double(point_count + 1);
  uint32_t pattern = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    mean[a] += (v[a] - mean[a]) * one_on_count_plus_one;
    int pos = pointToRegionCoord(mean[a] + offset, mean_resolution);
    pos = (pos >= 0 ? (pos < (1 << 10) ? pos : mean_positions) : 0);
    pattern |= uint32_t(pos) << (10 * a);
  }
  return pattern | (1u << 31);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_region_walk: the hot kernel.
//
// One workgroup (16 waves) per chunk of <= kChunkSegments ray-region segments of ONE region.  The region's miss-count
// tile lives in LDS: one u16 per voxel, 15 bits of count and the top bit holding the voxel's mask flag ("also receives
// samples"), so ONE returning LDS atomic per visit both counts the miss and fetches the flag.  Every lane resumes one
// ray's fp64 walk at the step that enters the region and visits the segment's voxels.  Idle lanes are refilled in
// batches from a workgroup-wide LDS cursor so waves stay mostly full although segments differ in length.
//
// A miss on a masked voxel must be ordered against that voxel's samples.  Such visits are appended to a per-wave LDS
// queue (no atomics: the queue cursor is wave-uniform) and resolved in bursts: against the region's sorted sample keys
// staged in LDS when they fit, otherwise through a global event list (k_flagged_events).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kWalkThreads = 1024;
constexpr int kWalkWaves = kWalkThreads / 64;
constexpr int kQueueCap = 128;     ///< deferred events per wave (8 B each)
#ifndef OHMHIP_LDS_HITS
#define OHMHIP_LDS_HITS 6144
#endif
constexpr int kLdsHits = OHMHIP_LDS_HITS;     ///< a region's sample list is staged in LDS when it has at most this many samples
constexpr uint32_t kIndexShift = 5;  ///< staged samples are indexed by voxel index >> kIndexShift ...
constexpr uint32_t kIndexBuckets = (1u << kHitVoxelBits) >> kIndexShift;  ///< ... in this many buckets (+ 1 end entry)
constexpr int kRefillMinIdle = 20; ///< refill a wave once this many lanes are idle
constexpr uint32_t kTileFlag = 0x8000u;       ///< mask flag inside a u16 tile entry
constexpr uint32_t kTileCountMask = 0x7fffu;  ///< count bits of a u16 tile entry (a chunk adds <= kMaxChunkSegments)
#ifndef OHMHIP_MAX_CHUNK_SEGMENTS
#define OHMHIP_MAX_CHUNK_SEGMENTS 8192
#endif
constexpr uint32_t kMaxChunkSegments = OHMHIP_MAX_CHUNK_SEGMENTS;  ///< bounded by the 15-bit counters and by the LDS order array
constexpr uint32_t kTraceChunks = 4096;  ///< debug trace: records kept per launch
constexpr uint32_t kTraceWords = 32;     ///< debug trace: u64 words per record
constexpr uint32_t kLengthClasses = 128;      ///< segment length histogram bins (lengths above the last bin share it)

/// Physical position of the count tile's logical word `w` (two voxels per word, voxel order).  A word's LDS bank is its
/// index modulo 32, which in voxel order is (x / 2, y & 1): lanes whose rays advance in step through a region -- a
/// lidar's vertical fan of beams has the same x and y in every lane -- would all hit one or two banks.  The tile is
/// therefore stored with the bank bits XOR-ed with the y / z bits of the index (a permutation inside every 32-word row).
__device__ inline uint32_t tileWord(uint32_t w)
{
  return w ^ (((w >> 5) ^ (w >> 10)) & 31u);
}

/// Byte address of the tile word holding the u16 entry at byte offset `va` (= 2 x voxel index).
__device__ inline uint32_t tileAddress(uint32_t va)
{
  return (va & ~3u) ^ ((((va >> 5) ^ (va >> 10)) & (31u << 2)));
}

/// Resolve one deferred miss event: find the first sample of the same voxel with a larger ray index; the miss counts
/// towards the interval before that sample, or towards the voxel's trailing count if there is none.
__device__ inline void resolveFlaggedMiss(unsigned long long key, const BatchScratch &bs,
                                          const unsigned long long *__restrict__ sorted_hits,
                                          uint32_t *__restrict__ miss_counts, uint32_t *__restrict__ interval_counts,
                                          int region_voxels)
{
  const uint32_t slot = uint32_t(key >> kHitSlotShift);
  const uint32_t vi = uint32_t(key >> kHitRayBits) & ((1u << kHitVoxelBits) - 1u);
  const uint32_t he = bs.info->n_hits;
  // Start at the voxel's first sample and step over the (few) samples with a smaller ray index.
  uint32_t lo = bs.voxel_first_hit[size_t(slot) * size_t(region_voxels) + vi];
  while (lo < he && (sorted_hits[lo] >> kHitRayBits) == (key >> kHitRayBits) && sorted_hits[lo] < key)
  {
    ++lo;
  }
  if (lo < he && (sorted_hits[lo] >> kHitRayBits) == (key >> kHitRayBits))
  {
    // The visit was counted in the voxel's miss count by the walk; move it to the interval before that sample.
    // (Integer add / sub commute, so the transient order against the tile flush does not matter.)
    atomicAdd(&interval_counts[lo], 1u);
    atomicSub(&miss_counts[size_t(slot) * size_t(region_voxels) + vi], 1u);
  }
}

/// Drain one wave's deferred-miss queue of (voxel, ray) pairs.  Preferred: order each miss against the region's samples
/// in LDS (binary search over the staged sorted keys, LDS atomics on the interval / trailing counters).  Otherwise
/// append the events to the global list in one coalesced burst (resolved by k_flagged_events, or sorted and replayed
/// for NDT / TSDF).
__device__ inline void flushQueue(const uint2 *queue, uint32_t qcount, unsigned lane, unsigned long long slot_bits,
                                  int ray_shift, bool lds_resolve, const unsigned long long *l_hits,
                                  const uint16_t *l_index, uint32_t n_region_hits, uint32_t *l_intervals,
                                  uint32_t *l_counts,
                                  unsigned long long *__restrict__ events, uint32_t event_capacity,
                                  uint32_t *__restrict__ event_count, int defer_all, const BatchScratch &bs,
                                  const unsigned long long *__restrict__ sorted_hits,
                                  uint32_t *__restrict__ miss_counts, uint32_t *__restrict__ interval_counts,
                                  int region_voxels)
{
  if (lds_resolve)
  {
    for (uint32_t q = lane; q < qcount; q += 64)
    {
      const uint2 e = queue[q];
      const unsigned long long ev =
        slot_bits | ((unsigned long long)e.x << kHitRayBits) | ((unsigned long long)e.y << ray_shift);
      // First staged sample with key > ev: the bucket index narrows the search to the samples of the event's 32
      // voxels (a handful), a binary search finishes it.
      uint32_t lo = l_index[e.x >> kIndexShift], hi = l_index[(e.x >> kIndexShift) + 1u];
      while (lo < hi)
      {
        const uint32_t mid = (lo + hi) >> 1;
        if (l_hits[mid] > ev)
        {
          hi = mid;
        }
        else
        {
          lo = mid + 1;
        }
      }
      // (lo may be the first sample of the next bucket: the voxel test below rejects it)
      if (lo < n_region_hits && (l_hits[lo] >> kHitRayBits) == (ev >> kHitRayBits))
      {
        // Belongs before a later sample of the voxel: move it from the voxel's count to that sample's interval.
        atomicAdd(&l_intervals[lo >> 1], 1u << ((lo & 1u) * 16u));
        atomicSub(&l_counts[tileWord(e.x >> 1)], 1u << ((e.x & 1u) * 16u));
      }
    }
    return;
  }
  uint32_t gbase = 0;
  if (lane == 0)
  {
    gbase = atomicAdd(event_count, qcount);
  }
  gbase = __shfl(gbase, 0);
  for (uint32_t q = lane; q < qcount; q += 64)
  {
    const uint2 e = queue[q];
    const unsigned long long ev =
      slot_bits | ((unsigned long long)e.x << kHitRayBits) | ((unsigned long long)e.y << ray_shift);
    if (gbase + q < event_capacity)
    {
      events[gbase + q] = ev;
    }
    else if (!defer_all)
    {
      resolveFlaggedMiss(ev, bs, sorted_hits, miss_counts, interval_counts, region_voxels);
    }
  }
}

// Explicit lane-mask selects for the walk step (see k_region_walk): `mask` is a wave-wide 64-bit lane mask in SGPRs.
constexpr int kFcmpOlt = 4;   ///< llvm::CmpInst::FCMP_OLT
constexpr int kIcmpSlt = 40;  ///< llvm::CmpInst::ICMP_SLT

__device__ inline int selectI(unsigned long long mask, int if_set, int if_clear)
{
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
}

__device__ inline double selectD(unsigned long long mask, double if_set, double if_clear)
{
  const int lo = selectI(mask, __double2loint(if_set), __double2loint(if_clear));
  const int hi = selectI(mask, __double2hiint(if_set), __double2hiint(if_clear));
  return __hiloint2double(hi, lo);
}

/// value + (lane's mask bit): one add-with-carry-in.
__device__ inline int addMask(int value, unsigned long long mask)
{
  asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(value) : "s"(mask) : "vcc");
  return value;
}

constexpr int kIcmpEq = 32;   ///< llvm::CmpInst::ICMP_EQ
constexpr int kIcmpUlt = 36;  ///< llvm::CmpInst::ICMP_ULT

/// mask ? if_set : 0
__device__ inline uint32_t selectOrZero(unsigned long long mask, uint32_t if_set)
{
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(if_set), "s"(mask));
  return r;
}

__device__ inline uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ inline uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

/// a + b, saturating at 2^32 - 1 (the predictor's candidates never wrap into small values).
__device__ inline uint32_t addSat(uint32_t a, uint32_t b)
{
  uint32_t r;
  asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

/// The same with a wave-uniform second operand (kept in an SGPR).
__device__ inline uint32_t addSatUniform(uint32_t a, uint32_t b)
{
  uint32_t r;
  asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b));
  return r;
}

/// Returning LDS add on the count tile.  Issued as inline assembly so that (a) the tile's address needs no base add (it
/// sits at LDS offset 0: the kernel has no static LDS, checked by the parity tests on every run) and (b) the wait for
/// the returned value is placed by hand, after the walk step (waitTile).
__device__ inline uint32_t tileAdd(uint32_t byte_address, uint32_t value)
{
  uint32_t old;
  asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(old) : "v"(byte_address), "v"(value) : "memory");
  return old;
}

/// 1 << (shift & 31): the hardware shift only reads the low five bits of its shift operand.
__device__ inline uint32_t shiftOne(uint32_t shift)
{
  uint32_t r;
  asm("v_lshlrev_b32_e64 %0, %1, 1" : "=v"(r) : "v"(shift));
  return r;
}

__device__ inline uint32_t waitTile(uint32_t old)
{
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(old) : : "memory");
  return old;
}

/// Local voxel coordinates of voxel index `vi` of a region.
__device__ inline void voxelLocal(const MapConst &mc, uint32_t vi, int &lx, int &ly, int &lz)
{
  const uint32_t dx = uint32_t(mc.dim[0]);
  const uint32_t dxy = dx * uint32_t(mc.dim[1]);
  lz = int(vi / dxy);
  const uint32_t r = vi - uint32_t(lz) * dxy;
  ly = int(r / dx);
  lx = int(r - uint32_t(ly) * dx);
}

/// Steps a ray has taken along each axis when it stands in voxel `vi` of region (rx, ry, rz): the walk moves
/// monotonically away from the start voxel on every axis.
__device__ inline void stepsAtVoxel(const MapConst &mc, const RayWalk &rw, int rx, int ry, int rz, uint32_t vi, int &k0,
                                    int &k1, int &k2)
{
  int lx, ly, lz;
  voxelLocal(mc, vi, lx, ly, lz);
  k0 = abs(rx * mc.dim[0] + lx - rw.g0[0]);
  k1 = abs(ry * mc.dim[1] + ly - rw.g0[1]);
  k2 = abs(rz * mc.dim[2] + lz - rw.g0[2]);
}

/// time_next of one axis after k steps along it (ohm/LineWalkCompute.h:299-301, :375-378).
__device__ inline double timeNext(double init, double delta, int k, int total)
{
  return (k < total) ? ((k == 0) ? init : init + delta * double(k)) : dInf();
}

/// The exact decision of the reference walk for a ray standing in voxel `vi` of region (rx, ry, rz): the axis of the
/// next step.  walkSelectNextAxis (ohm/LineWalkCompute.h:282-289): smallest time_next, ties go to the higher axis.
__device__ inline int exactNextAxis(const MapConst &mc, const RayWalk &rw, int rx, int ry, int rz, uint32_t vi)
{
  int k0, k1, k2;
  stepsAtVoxel(mc, rw, rx, ry, rz, vi, k0, k1, k2);
  const double t0 = timeNext(rw.init[0], rw.delta[0], k0, rw.total[0]);
  const double t1 = timeNext(rw.init[1], rw.delta[1], k1, rw.total[1]);
  const double t2 = timeNext(rw.init[2], rw.delta[2], k2, rw.total[2]);
  const bool m01 = t0 < t1;
  const double t01 = m01 ? t0 : t1;
  const bool m2 = t01 < t2;
  return m2 ? (m01 ? 0 : 1) : 2;
}

/// Kernel parameters of k_region_walk (one struct keeps the template instantiations readable).
struct WalkArgs
{
  MapConst mc;
  BatchScratch bs;
  const Chunk *chunks;
  const Segment *segments;
  const RayWalk *walks;
  const uint64_t *slot_keys;  ///< region key per slot (exact decisions need the region's coordinates)
  const unsigned long long *sorted_hits;
  const uint32_t *hit_mask;
  uint32_t *miss_counts;
  uint32_t *interval_counts;
  unsigned long long *events;
  uint32_t event_capacity;
  uint32_t *event_count;
  int refill_min_idle;
  unsigned dbg;
  int ray_shift;
  int defer_all;     ///< NDT / TSDF: every visit to a masked voxel becomes an event; masked voxels are not counted
  float *occupancy;  ///< non-null: single-chunk regions are applied straight from LDS
  unsigned ray_flags;
  unsigned long long *dbg_counters;
  float *tsdf;       ///< non-null (TSDF mode): single-chunk regions are applied straight from LDS
  uint32_t *chunk_cursor;  ///< device-wide next-chunk cursor (zeroed before the launch)
  uint32_t n_chunks;
  /// NDT / TSDF: this launch repeats a walk whose event list overflowed.  Regions held by a single chunk had their plain
  /// counts applied to the layers by the first launch already: the repeat only regenerates their events.
  int rewalk;
  /// TSDF with weight drop-off: a free-space visit changes the weight by a value that depends on the voxel and the
  /// ray, so no voxel can be counted -- every visit of the batch is an event for the ordered replay.
  int flag_all;
  /// Occupancy maps without mean / secondary layers: a region held by a single chunk whose samples are all staged in
  /// LDS has its samples replayed by the walk's epilogue itself (the ordered sample list, the interval counters and the
  /// trailing counts are all in LDS at that point); the region is marked kSamplesApplied for k_apply_hits.
  int inline_hits;
};

/// Top bit of BatchScratch::hit_begin[slot], set by the walk kernel once it has replayed the region's samples itself
/// (the array is this batch's own copy -- see ohmhip_map.hip: parity -- and k_plan rewrites the entry of every region a
/// batch touches, so the mark lives exactly from the walk to the end of the batch).
constexpr uint32_t kSamplesApplied = 0x80000000u;

constexpr double kTraversalScale = 1099511627776.0;  ///< 2^40 fixed-point units per metre of traversal
constexpr uint32_t kWalkCursorWords = 24;  ///< l_cursor[]: see k_region_walk
#ifndef OHMHIP_WALK_UNROLL
#define OHMHIP_WALK_UNROLL 2
#endif
constexpr int kWalkUnroll = OHMHIP_WALK_UNROLL;  ///< walk steps per loop trip (see the loop)

/// Shape of a walk workgroup.  WalkFull is the one the design was tuned on: 1024 threads own a CU with a 32 768-voxel
/// tile.  WalkHalf (round 6) can serve regions / tiles of up to 16 384 voxels with everything halved -- threads, tile,
/// staged samples, segments per chunk -- so that TWO workgroups share a CU (at most 80.7 of 81.9 KB of LDS each, 8 waves of
/// 128 VGPRs each): one's prologue and epilogue run under the other's walk loop.  The per-thread shares (segments and
/// samples prefetched per thread, tile words per thread in the epilogue) are the same in both.  The host picks it for
/// regions of at most 4 096 voxels (16^3: -10 % per batch), where it was measured to pay (ohmhip_map.hip).
template <int kThreadsT, int kTileVoxelsT, int kLdsHitsT, uint32_t kSegmentsT, int kQueueCapT, int kMinWavesPerEuT>
struct WalkGeometry
{
  static constexpr int kMinWavesPerEu = kMinWavesPerEuT;  ///< __launch_bounds__: 4 keeps WalkHalf at 128 VGPRs (two workgroups per CU)
  static constexpr int kThreads = kThreadsT;
  static constexpr int kWaves = kThreadsT / 64;
  static constexpr int kTileVoxels = kTileVoxelsT;  ///< largest region / tile the shape serves
  static constexpr int kLdsHits = kLdsHitsT;        ///< samples of a region staged in LDS
  static constexpr uint32_t kSegments = kSegmentsT; ///< segments per chunk (LDS order array)
  static constexpr int kQueueCap = kQueueCapT;      ///< deferred events per wave
};
using WalkFull = WalkGeometry<kWalkThreads, 1 << kHitVoxelBits, kLdsHits, kMaxChunkSegments, kQueueCap, 1>;
using WalkHalf = WalkGeometry<kWalkThreads / 2, (1 << kHitVoxelBits) / 2, kLdsHits / 2, kMaxChunkSegments / 2, 96, 4>;
/// kSpecial: the batch contains rays whose end voxel is part of the walk (clipped / kRfEndPointAsFree / TSDF) or
/// kRfExcludeOrigin.  The common case (kSpecial == false) keeps those predicates out of the hot loop: every iteration
/// of an active lane is a miss.
/// kTrace: development instrumentation (OHMHIP_DEBUG_FLAGS 64 / 128): per-chunk time stamps and loop counters.  Compiled
/// out of the production instantiations.
template <bool kSpecial, bool kTrace, typename G = WalkFull>
__global__ void __launch_bounds__(G::kThreads, G::kMinWavesPerEu) k_region_walk(WalkArgs args)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const MapConst &mc = args.mc;
  // Layout: [count tile: ceil(region_voxels / 2) words][queues][staged sample keys][interval counters][cursor]
  // [length histogram][segment order: u16 per segment].  The tile sits at offset 0 so the per-visit atomic needs no
  // base add.
  const uint32_t count_words = uint32_t(mc.region_voxels + 1) >> 1;
  const uint32_t mask_words = uint32_t(mc.region_voxels + 31) >> 5;
  uint32_t *l_counts = lds;
  uint2 *l_queues = reinterpret_cast<uint2 *>(lds + ((count_words + 31u) & ~31u));  // (whole rows: see tileWord)
  unsigned long long *l_hits = reinterpret_cast<unsigned long long *>(l_queues + G::kWaves * G::kQueueCap);
  uint32_t *l_intervals = reinterpret_cast<uint32_t *>(l_hits + G::kLdsHits);  // [G::kLdsHits] u16 interval counters
  // l_cursor[0]: segment cursor, [1]: a fetched chunk's index, [2..5]: its record, [6..7]: its samples, [8..9]: its
  // region key; [12..21]: a second record (start-up only)
  uint32_t *l_cursor = l_intervals + G::kLdsHits / 2;
  uint32_t *l_hist = l_cursor + kWalkCursorWords;
  uint32_t *l_idle = l_hist + kLengthClasses;  // [64] scratch words: where a lane with nothing to visit aims its LDS add
  uint16_t *l_index = reinterpret_cast<uint16_t *>(l_idle + 64);  // [kIndexBuckets + 2] first staged sample per bucket
  uint16_t *l_order = l_index + kIndexBuckets + 2;

  // Persistent workgroups: the launch has one workgroup per CU and each takes chunks from a device-wide cursor until
  // none are left.  A static blockIdx -> chunk binding leaves the hardware's round-robin of workgroups over the 8 XCDs
  // in charge of the balance, and chunk costs vary enough that some XCDs then finish in half the time of others.
  // The chunk record (and the region's range in the sample list) travels with the index through LDS: thread 0 fetches
  // the next one while the other waves are still finishing their loop, so a trip does not start with a chain of
  // dependent global loads.
  const int defer_all = args.defer_all;
  // A workgroup's first chunk is the one with its own index (the list is ordered largest first, so the launch starts on
  // the gridDim.x largest chunks, one per workgroup, whatever order the workgroups arrive in); the shared cursor hands
  // out the chunks behind those.
  auto fetchNextChunk = [&](uint32_t *record, bool first = false) {
    const uint32_t next = first ? blockIdx.x : gridDim.x + atomicAdd(args.chunk_cursor, 1u);
    record[1] = next;
    if (next < args.n_chunks)
    {
      const Chunk c = args.chunks[next];
      record[2] = c.slot;
      record[3] = c.seg_begin;
      record[4] = c.seg_end;
      record[5] = c.hash_index;
      record[6] = defer_all ? 0u : args.bs.hit_begin[c.slot];
      record[7] = defer_all ? 0u : args.bs.hit_end[c.slot];
      const uint64_t key = args.slot_keys[c.slot];
      record[8] = uint32_t(key);
      record[9] = uint32_t(key >> 32);
    }
  };
  // A workgroup holds two chunk records: the chunk it works on and the next one.  While it walks chunk N every thread
  // loads its share of chunk N + 1's prologue inputs (segment lengths, the region's sample keys and mask words) into
  // registers -- issued behind the first lane refill of chunk N, so the loads complete under the walk and the next
  // prologue starts without a memory round trip -- and thread 0 claims chunk N + 2 while the other waves finish their
  // loop.
  constexpr int kSegPerThread = int(G::kSegments) / G::kThreads;
  constexpr int kHitsPerThread = G::kLdsHits / uint32_t(G::kThreads);
  struct ChunkRecord
  {
    uint32_t index, slot, seg_begin, seg_end, hash_index, hb, he, key_lo, key_hi;
  };
  auto readRecord = [&](const uint32_t *record) {
    // (readfirstlane: the values are wave-uniform, so the chunk loop's condition is a scalar branch and the barriers
    // inside the loop are not restructured as if threads could leave at different trips.)
    ChunkRecord r;
    r.index = __builtin_amdgcn_readfirstlane(record[1]);
    r.slot = __builtin_amdgcn_readfirstlane(record[2]);
    r.seg_begin = __builtin_amdgcn_readfirstlane(record[3]);
    r.seg_end = __builtin_amdgcn_readfirstlane(record[4]);
    r.hash_index = __builtin_amdgcn_readfirstlane(record[5]);
    r.hb = __builtin_amdgcn_readfirstlane(record[6]);
    r.he = __builtin_amdgcn_readfirstlane(record[7]);
    r.key_lo = __builtin_amdgcn_readfirstlane(record[8]);
    r.key_hi = __builtin_amdgcn_readfirstlane(record[9]);
    return r;
  };
  unsigned long long pf_hits[kHitsPerThread] = {};
  uint32_t pf_vox[kSegPerThread] = {};
  uint32_t pf_mask = 0;
  // Every global load of a chunk's prologue that depends only on the chunk record, issued back to back: clamped instead
  // of predicated so the loads share one basic block.
  auto prefetchChunk = [&](const ChunkRecord &r) {
    const uint32_t n_hits = r.he - r.hb;
    if (!defer_all && n_hits && n_hits <= uint32_t(G::kLdsHits))
    {
#pragma unroll
      for (int j = 0; j < kHitsPerThread; ++j)
      {
        pf_hits[j] = args.sorted_hits[r.hb + min(threadIdx.x + uint32_t(j) * uint32_t(G::kThreads), n_hits - 1u)];
      }
    }
    pf_mask = args.hit_mask[size_t(r.slot) * mask_words + min(threadIdx.x, mask_words - 1u)];
    const uint32_t n = r.seg_end - r.seg_begin;
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j)
    {
      pf_vox[j] = args.segments[r.seg_begin + min(threadIdx.x + uint32_t(j) * uint32_t(G::kThreads), n - 1u)].vox;
    }
  };
  // The first two records, fetched by two waves at once.
  if (threadIdx.x == 0 || threadIdx.x == 64)
  {
    fetchNextChunk(l_cursor + (threadIdx.x ? kWalkCursorWords / 2 : 0), threadIdx.x == 0);
  }
  __syncthreads();
  ChunkRecord cur = readRecord(l_cursor);
  ChunkRecord next = readRecord(l_cursor + kWalkCursorWords / 2);
  if (next.index < cur.index)
  {
    const ChunkRecord swap = cur;
    cur = next;
    next = swap;
  }
  if (cur.index < args.n_chunks)
  {
    prefetchChunk(cur);
  }
  __syncthreads();  // (everyone has read the records: thread 0 may overwrite the first one)
  while (cur.index < args.n_chunks)
  {
    unsigned long long clk_start = 0;
    if (kTrace)
    {
      clk_start = wall_clock64();
    }
    const uint32_t chunk_index = cur.index;
    Chunk chunk;
    chunk.slot = cur.slot;
    chunk.seg_begin = cur.seg_begin;
    chunk.seg_end = cur.seg_end;
    chunk.hash_index = cur.hash_index;
    const uint32_t hb = cur.hb;
    const uint32_t he = cur.he;
    // Region coordinates (packRegionKey): only the exact decisions use them.
    const int region_x = int(int16_t(cur.key_lo & 0xffffu));
    const int region_y = int(int16_t(cur.key_lo >> 16));
    const int region_z = int(int16_t(cur.key_hi & 0xffffu));
    const uint32_t n_seg = chunk.seg_end - chunk.seg_begin;
    const Segment *chunk_segments = args.segments + chunk.seg_begin;

    // ---- prologue.  One workgroup owns the CU (the tile takes most of its LDS), so nothing overlaps this phase; its
    // ---- inputs are in registers already (prefetchChunk).
    // The region's sorted sample keys are staged in LDS so deferred misses can be ordered against them at LDS latency.
    const uint32_t n_region_hits = he - hb;
    const bool lds_resolve = !defer_all && n_region_hits <= uint32_t(G::kLdsHits);
    unsigned long long my_hits[kHitsPerThread];
#pragma unroll
    for (int j = 0; j < kHitsPerThread; ++j)
    {
      my_hits[j] = pf_hits[j];
    }
    const uint32_t *g_mask = args.hit_mask + size_t(chunk.slot) * mask_words;
    const uint32_t my_mask = pf_mask;
    uint32_t lens[kSegPerThread];
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j)
    {
      lens[j] = min(pf_vox[j] >> kSegVoxelBits, kLengthClasses - 1u);
    }
    if (threadIdx.x < kLengthClasses)
    {
      l_hist[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0)
    {
      l_cursor[0] = 0;
    }
    if (threadIdx.x < 64)
    {
      l_idle[threadIdx.x] = 0;
    }
    const bool stamp = kTrace && threadIdx.x == 0;
    unsigned long long clk_p[6] = { 0, 0, 0, 0, 0, 0 };
    if (stamp)
    {
      clk_p[0] = wall_clock64();
    }
    // Tile entries start at zero count with the voxel's mask flag in the top bit: one mask word covers 16 tile words,
    // half a row of the tile, which tileWord() maps onto half a row again: the 4-word groups permuted by the high bits
    // of the row's XOR constant, the words inside a group by its low two bits.
    for (uint32_t w = threadIdx.x; w < mask_words; w += uint32_t(G::kThreads))
    {
      const uint32_t mword = args.flag_all ? 0xffffffffu : ((w == threadIdx.x) ? my_mask : g_mask[w]);
      const uint32_t swizzle = tileWord(w * 16u) ^ (w * 16u);
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q)
      {
        const uint32_t logical = w * 16u + q * 4u;
        if (logical + 3u < count_words)
        {
          uint32_t v[4];
#pragma unroll
          for (uint32_t r = 0; r < 4; ++r)
          {
            // physical word r of the group holds logical word r ^ (swizzle & 3)
            const uint32_t two = (mword >> ((q * 4u + (r ^ (swizzle & 3u))) * 2u)) & 3u;
            v[r] = ((two & 1u) << 15) | ((two & 2u) << 30);
          }
          *reinterpret_cast<uint4 *>(&l_counts[logical ^ (swizzle & ~3u)]) = make_uint4(v[0], v[1], v[2], v[3]);
        }
        else
        {
          for (uint32_t r = 0; r < 4; ++r)
          {
            if (logical + r < count_words)
            {
              const uint32_t two = (mword >> ((q * 4u + r) * 2u)) & 3u;
              l_counts[tileWord(logical + r)] = ((two & 1u) << 15) | ((two & 2u) << 30);
            }
          }
        }
      }
    }
    if (stamp)
    {
      clk_p[1] = wall_clock64();
    }
    __syncthreads();
    bool prefetched = false;  // wave-uniform: the next chunk's prologue loads have been issued
    if (stamp)
    {
      clk_p[2] = wall_clock64();
    }
    // Longest segments first (counting sort on the voxel count, indices in LDS): lanes refilled together get segments
    // of similar length and so retire together, and the workgroup drains on its SHORTEST segments instead of waiting
    // for a few long stragglers.  The order inside a length class is arbitrary; integer counting does not care.
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j)
    {
      if (threadIdx.x + uint32_t(j) * uint32_t(G::kThreads) < n_seg)
      {
        atomicAdd(&l_hist[lens[j]], 1u);
      }
    }
    if (lds_resolve && n_region_hits)
    {
#pragma unroll
      for (int j = 0; j < kHitsPerThread; ++j)
      {
        const uint32_t i = threadIdx.x + uint32_t(j) * uint32_t(G::kThreads);
        if (i < n_region_hits)
        {
          l_hits[i] = my_hits[j];
        }
        if (i < (n_region_hits + 1) / 2)
        {
          l_intervals[i] = 0;  // two u16 counters per word (a chunk adds at most kMaxChunkSegments to one counter)
        }
      }
    }
    __syncthreads();
    if (stamp)
    {
      clk_p[3] = wall_clock64();
    }
    if (lds_resolve && n_region_hits)
    {
      // Bucket index over the staged samples (sorted by voxel): l_index[b] = first sample of a voxel in bucket >= b.
#pragma unroll
      for (int j = 0; j < kHitsPerThread; ++j)
      {
        const uint32_t i = threadIdx.x + uint32_t(j) * uint32_t(G::kThreads);
        if (i < n_region_hits)
        {
          auto bucketOf = [](unsigned long long key) {
            return (uint32_t(key >> kHitRayBits) & ((1u << kHitVoxelBits) - 1u)) >> kIndexShift;
          };
          const uint32_t b = bucketOf(my_hits[j]);
          const uint32_t first = (i == 0) ? 0u : bucketOf(l_hits[i - 1]) + 1u;
          for (uint32_t k = first; k <= b; ++k)
          {
            l_index[k] = uint16_t(i);
          }
          if (i + 1 == n_region_hits)
          {
            for (uint32_t k = b + 1u; k <= kIndexBuckets; ++k)
            {
              l_index[k] = uint16_t(n_region_hits);
            }
          }
        }
      }
    }
    if (threadIdx.x < 64)
    {
      // Exclusive scan over the classes in DESCENDING length order: lane l owns classes 127 - 2l and 126 - 2l.
      const uint32_t hi_class = kLengthClasses - 1u - 2u * threadIdx.x;
      const uint32_t a = l_hist[hi_class];
      const uint32_t b = l_hist[hi_class - 1u];
      uint32_t incl = a + b;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1)
      {
        const uint32_t up = __shfl_up(incl, d);
        incl += (int(threadIdx.x) >= d) ? up : 0u;
      }
      const uint32_t excl = incl - (a + b);
      l_hist[hi_class] = excl;
      l_hist[hi_class - 1u] = excl + a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j)
    {
      const uint32_t i = threadIdx.x + uint32_t(j) * uint32_t(G::kThreads);
      if (i < n_seg)
      {
        l_order[atomicAdd(&l_hist[lens[j]], 1u)] = uint16_t(i);
      }
    }
    __syncthreads();
    if (stamp)
    {
      clk_p[4] = wall_clock64();
      if (chunk_index < kTraceChunks)
      {
        unsigned long long *rec = args.dbg_counters + 16 + size_t(chunk_index) * kTraceWords;
        rec[20] = clk_p[0];
        rec[21] = clk_p[1];
        rec[22] = clk_p[2];
        rec[23] = clk_p[3];
      }
    }

    const unsigned lane = laneId();
    const unsigned wave = threadIdx.x >> 6;
    uint2 *queue = l_queues + wave * G::kQueueCap;
    const int dimx = mc.dim[0];
    const int dimxy = mc.dim[0] * mc.dim[1];
    const unsigned long long slot_bits = (unsigned long long)chunk.slot << kHitSlotShift;
    const int ray_shift = args.ray_shift;
    const int refill_min_idle = args.refill_min_idle;
    const bool refill_only = kTrace && (args.dbg & 16u) != 0;
    const uint32_t fix_margin = mc.fix_margin;
    const uint32_t idle_address = uint32_t(reinterpret_cast<char *>(l_idle + lane) - reinterpret_cast<char *>(lds));

    // Per-lane walk state (all named scalars: no run-time indexed arrays).
    int left = 0;  // voxels this lane still has to visit in its segment (<= 0: idle)
    uint32_t f0 = 0, f1 = 0, f2 = 0, d0 = 0, d1 = 0, d2 = 0;  // predictor: next step time / step delta per axis
    int sx = 0, sy = 0, sz = 0;  // change of `va` per step along each axis
    uint32_t va = 0;             // byte offset of the current voxel's u16 tile entry (2 x voxel index)
    uint32_t ray = 0;
    uint32_t skip = 0;      // kSpecial only: first voxel is not visited (kRfExcludeOrigin)
    uint32_t end_last = 0;  // kSpecial only: the segment's last voxel is the ray's end voxel
    uint32_t qcount = 0;     // wave-uniform
    bool exhausted = false;  // wave-uniform
    uint32_t dbg_iters = 0, dbg_active = 0, dbg_refills = 0, dbg_fm = 0, dbg_slow = 0;  // wave-uniform (kTrace)
    uint32_t dbg_s2 = 0, dbg_s3 = 0, dbg_sl = 0, dbg_sp = 0;
    unsigned long long clk_loop = 0;
    if (kTrace)
    {
      clk_loop = wall_clock64();
    }

    // The walk loop.  A wave's trip is a chain of dependent hops (LDS round trip, mask algebra on the scalar unit,
    // branches), and with four waves per SIMD the chain, not instruction issue, sets the pace.  So one trip takes
    // kWalkUnroll steps per lane: one refill / exit test per trip, the steps' LDS adds in flight together, their
    // returned flags tested after the last step.  A lane whose segment ends inside a trip idles for the rest of it.
    int refill_threshold = refill_min_idle;  // idle lanes that trigger a refill; 64 once the chunk has no segments left
    while (true)
    {
      // ---- refill idle lanes (wave-uniform decision) ------------------------------------------------------------------
      const unsigned long long am = __ballot(left > 0);
      const int n_idle = 64 - __popcll(am);
      if (__builtin_expect(n_idle >= refill_threshold, 0))
      {
        if (exhausted)
        {
          break;  // every lane idle and nothing left to hand out
        }
        if (kTrace)
        {
          ++dbg_refills;
        }
        uint32_t base = 0;
        if (lane == 0)
        {
          base = atomicAdd(l_cursor, uint32_t(n_idle));
        }
        base = __builtin_amdgcn_readfirstlane(base);
        exhausted = base + uint32_t(n_idle) >= n_seg;
        refill_threshold = exhausted ? 64 : refill_threshold;
        const unsigned long long idle = ~am;
        const uint32_t mine =
          base + __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
        if (left <= 0 && mine < n_seg)
        {
          const uint4 *rec = reinterpret_cast<const uint4 *>(chunk_segments + l_order[mine]);
          const uint4 ra = rec[0];
          const uint4 rb = rec[1];
          f0 = ra.x;
          f1 = ra.y;
          f2 = ra.z;
          va = (ra.w & ((1u << kSegVoxelBits) - 1u)) << 1;
          left = int(ra.w >> kSegVoxelBits);
          d0 = rb.x & ~kSegNegative;
          d1 = rb.y & ~kSegNegative;
          d2 = rb.z & ~kSegNegative;
          sx = (rb.x & kSegNegative) ? -2 : 2;
          sy = (rb.y & kSegNegative) ? -2 * dimx : 2 * dimx;
          sz = (rb.z & kSegNegative) ? -2 * dimxy : 2 * dimxy;
          ray = rb.w & kSegRayMask;
          if (kSpecial)
          {
            skip = (rb.w & kSegSkipFirst) ? 1u : 0u;
            end_last = (rb.w & kSegEnd) ? 1u : 0u;
          }
          left = refill_only ? 0 : left;
        }
        if (!prefetched)
        {
          // First refill of the chunk: the lanes' records are on their way, now queue the next chunk's prologue loads
          // behind them (loads return in order, so nothing in this chunk ever waits for these).
          prefetched = true;
          if (next.index < args.n_chunks)
          {
            prefetchChunk(next);
          }
        }
      }

      uint32_t olds[kWalkUnroll];     // tile word returned by each step's LDS add
      uint32_t visited[kWalkUnroll];  // `va` of each step's voxel
#pragma unroll
      for (int u = 0; u < kWalkUnroll; ++u)
      {
        // ---- visit: count the miss and fetch the voxel's mask flag with one returning LDS atomic.  Masked voxels
        // ---- (which also receive samples) are counted too; the ordering pass moves such a miss to an interval counter
        // ---- when a later sample of the voxel exists.  `va` is the byte offset of the voxel's u16 tile entry
        // ---- (2 x voxel index): word address = va & ~3, and the shifts only read the low five bits of their shift
        // ---- operand, so (va << 3) selects bit 0 or 16 of the word for the count and (.. | 15) bit 15 or 31 for the
        // ---- flag.  A lane with nothing to visit adds to its own scratch word instead (no exec-mask juggling; the
        // ---- scratch words start every chunk at zero and a lane idles for far fewer than 2^15 steps of a chunk, so
        // ---- their flag bits stay clear).
        // kSpecial: the ray's end voxel (last voxel of a kSegEnd segment) is always visited; kRfExcludeOrigin drops
        // the first voxel of the ray otherwise.
        const bool at_end = kSpecial && end_last && left == 1;
        const bool visit = kSpecial ? (left > 0 && (at_end || !skip)) : (left > 0);
        visited[u] = va;
        olds[u] = tileAdd(visit ? tileAddress(va) : idle_address, shiftOne(va << 3));
        if (kSpecial)
        {
          skip = 0;
        }
        if (kTrace)
        {
          ++dbg_iters;
          dbg_active += uint32_t(__popcll(__ballot(visit)));
        }

        int stride;
        {
          // ---- one walk step from the fixed-point predictor (see Segment), taken by every lane.  The smallest
          // ---- candidate is trusted when it lies inside the region's range (below kFixMaxDelta) and leads the second
          // ---- smallest by more than the accumulated truncation error; a visiting lane that cannot trust it asks the
          // ---- reference's fp64 arithmetic (rare: near-ties, ray ends that disagree with their keys, degenerate rays).
          const uint32_t fmin = umin3(f0, f1, f2);
          const uint32_t fmed = umed3(f0, f1, f2);
          const uint32_t limit = min(fmed, kFixMaxDelta);
          const uint32_t lead = addSatUniform(fmin, fix_margin);
          const unsigned long long certain = __builtin_amdgcn_uicmp(lead, limit, kIcmpUlt);
          unsigned long long a0 = __builtin_amdgcn_uicmp(f0, fmin, kIcmpEq);
          unsigned long long a2 = __builtin_amdgcn_uicmp(f2, fmin, kIcmpEq);
          // (a lane on its segment's LAST voxel takes a step nobody uses -- its predictors are parked when the ray ends
          // there, which would send every ray of a TSDF / end-point-as-free batch through the exact path once for nothing)
          const unsigned long long slow = __ballot(left > 1) & ~certain;
          if (__builtin_expect(slow != 0, 0))
          {
            int axis = 1;
            if ((slow >> lane) & 1ull)
            {
              uint32_t r = ray;
              asm volatile("" : "+v"(r));  // keeps the record's address arithmetic inside this (rare) block
              axis = exactNextAxis(mc, args.walks[r], region_x, region_y, region_z, va >> 1);
            }
            a0 = (a0 & ~slow) | (slow & __ballot(axis == 0));
            a2 = (a2 & ~slow) | (slow & __ballot(axis == 2));
            if (kTrace)
            {
              ++dbg_slow;
              dbg_s2 += uint32_t(__popcll(slow & __ballot(left == 2)));
              dbg_s3 += uint32_t(__popcll(slow & __ballot(left == 3)));
              dbg_sl += uint32_t(__popcll(slow));
              dbg_sp += uint32_t(__popcll(slow & __ballot((d0 | d1 | d2) == 0u)));
            }
          }
          const unsigned long long a1 = ~(a0 | a2);
          f0 = addSat(f0, selectOrZero(a0, d0));
          f1 = addSat(f1, selectOrZero(a1, d1));
          f2 = addSat(f2, selectOrZero(a2, d2));
          stride = selectI(a2, sz, selectI(a0, sx, sy));
        }
        va += uint32_t(stride);
        left -= 1;
      }

      // ---- deferred ordering of misses on masked voxels.  The returned tile words are consumed after the trip's last
      // ---- step, so the LDS round trips are covered by the step arithmetic (waitTile carries the s_waitcnt).
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kWalkUnroll; ++u)
      {
        olds[u] = waitTile(olds[u]);  // (the first one waits; LDS operations return in order)
      }
#pragma unroll
      for (int u = 0; u < kWalkUnroll; ++u)
      {
        // (a lane that did not visit holds its scratch word, whose flag bits are clear)
        const bool flagged = __builtin_amdgcn_ubfe(olds[u], (visited[u] << 3) | 15u, 1u) != 0;
        const unsigned long long fm = __ballot(flagged);
        if (kTrace)
        {
          dbg_fm += fm ? 1u : 0u;
        }
        if (fm)
        {
          if (flagged)
          {
            const uint32_t pos =
              qcount + __builtin_amdgcn_mbcnt_hi(uint32_t(fm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(fm), 0u));
            queue[pos] = make_uint2(visited[u] >> 1, ray);
          }
          qcount += uint32_t(__popcll(fm));
          if (qcount > uint32_t(G::kQueueCap - 64))
          {
            flushQueue(queue, qcount, lane, slot_bits, ray_shift, lds_resolve, l_hits, l_index, n_region_hits,
                       l_intervals, l_counts, args.events, args.event_capacity, args.event_count, defer_all, args.bs,
                       args.sorted_hits, args.miss_counts, args.interval_counts, mc.region_voxels);
            qcount = 0;
          }
        }
      }
    }

    if (kTrace && lane == 0)
    {
      const unsigned long long clk_end_loop = wall_clock64();
      if (chunk_index < kTraceChunks)
      {
        unsigned long long *rec = args.dbg_counters + 16 + size_t(chunk_index) * kTraceWords;
        if (wave < 15)
        {
          rec[2 + wave] = clk_end_loop;
        }
        if (wave == 0)
        {
          rec[0] = n_seg | ((unsigned long long)((chunk.hash_index >> 31) & 1u) << 32);
          rec[1] = clk_loop;
          rec[18] = clk_start;
          rec[17] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |
                    ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);  // HW_ID | XCC_ID
        }
      }
      if (args.dbg & 128u)
      {
        atomicAdd(&args.dbg_counters[0], (unsigned long long)dbg_iters);
        atomicAdd(&args.dbg_counters[1], (unsigned long long)dbg_active);
        atomicAdd(&args.dbg_counters[2], (unsigned long long)dbg_refills);
        atomicAdd(&args.dbg_counters[3], (unsigned long long)dbg_fm);
        atomicAdd(&args.dbg_counters[4], (unsigned long long)dbg_slow);
        atomicAdd(&args.dbg_counters[5], (unsigned long long)dbg_s2);
        atomicAdd(&args.dbg_counters[6], (unsigned long long)dbg_s3);
        atomicAdd(&args.dbg_counters[7], (unsigned long long)dbg_sl);
        atomicAdd(&args.dbg_counters[8], (unsigned long long)dbg_sp);
      }
    }
    // Final queue flush.
    if (qcount)
    {
      flushQueue(queue, qcount, lane, slot_bits, ray_shift, lds_resolve, l_hits, l_index, n_region_hits, l_intervals,
                 l_counts, args.events, args.event_capacity, args.event_count, defer_all, args.bs, args.sorted_hits,
                 args.miss_counts, args.interval_counts, mc.region_voxels);
    }
    if (threadIdx.x == 0)
    {
      fetchNextChunk(l_cursor);  // the chunk after the next one; overlaps with the other waves finishing their loop
    }
    __syncthreads();
    const ChunkRecord after_next = readRecord(l_cursor);
    if (stamp && chunk_index < kTraceChunks)
    {
      args.dbg_counters[16 + size_t(chunk_index) * kTraceWords + 24] = wall_clock64();  // epilogue start
    }

    const bool inline_hits = args.inline_hits && lds_resolve && n_region_hits > 0u && args.occupancy && !args.rewalk &&
                             (chunk.hash_index & 0x80000000u);
    if (lds_resolve && !inline_hits)
    {
      for (uint32_t i = threadIdx.x; i < n_region_hits; i += uint32_t(G::kThreads))
      {
        const uint32_t c = (l_intervals[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu;
        if (c)
        {
          atomicAdd(&args.interval_counts[hb + i], c);
        }
      }
    }
    uint32_t *g_counts = args.miss_counts + size_t(chunk.slot) * size_t(mc.region_voxels);
    if (args.rewalk && (chunk.hash_index & 0x80000000u))
    {
      __syncthreads();
      cur = next;
      next = after_next;
      continue;
    }
    if (args.occupancy && (chunk.hash_index & 0x80000000u))
    {
      // This chunk holds ALL of the region's segments for the batch: apply the miss counts to the log-odds layer
      // straight from LDS (no count round trip through HBM).  Voxels which also receive samples keep their count for
      // the ordered replay (occupancy: k_apply_hits; NDT: their visits are events, the tile entry is not used).
      float *g_occ = args.occupancy + size_t(chunk.slot) * size_t(mc.region_voxels);
      // Load pass / update pass, so the loads of the voxels a thread updates are in flight together (a load -> update ->
      // store loop would pay the memory latency once per touched word, and nothing else runs on this CU to hide it);
      // in two halves, which keeps the kernel's register peak below the walk loop's budget.
      constexpr uint32_t kWordsPerThread = uint32_t(G::kTileVoxels) / 2u / uint32_t(G::kThreads) / 2u;
      const bool even_voxels = (mc.region_voxels & 1) == 0;
      for (uint32_t half = 0; half < 2u; ++half)
      {
      uint32_t words[kWordsPerThread];
      float2 values[kWordsPerThread];
#pragma unroll
      for (uint32_t j = 0; j < kWordsPerThread; ++j)
      {
        const uint32_t i = threadIdx.x + (half * kWordsPerThread + j) * uint32_t(G::kThreads);
        const uint32_t flagged_w = (i < count_words) ? l_counts[tileWord(i)] : 0u;
        uint32_t w = flagged_w;
        // Keep only the entries applied here: unflagged voxels with a count.
        w = (w & kTileFlag) ? (w & 0xffff0000u) : w;
        w = (w & (kTileFlag << 16)) ? (w & 0x0000ffffu) : w;
        if (!defer_all && !inline_hits)
        {
          // Voxels which also receive samples keep their count for the ordered replay (k_apply_hits).
          if ((flagged_w & kTileFlag) && (flagged_w & kTileCountMask))
          {
            atomicAdd(&g_counts[2 * i], flagged_w & kTileCountMask);
          }
          if ((flagged_w & (kTileFlag << 16)) && ((flagged_w >> 16) & kTileCountMask))
          {
            atomicAdd(&g_counts[2 * i + 1], (flagged_w >> 16) & kTileCountMask);
          }
        }
        words[j] = w;
        values[j] = make_float2(0.0f, 0.0f);
        if (w)
        {
          if (even_voxels)
          {
            values[j] = *reinterpret_cast<const float2 *>(&g_occ[2 * i]);
          }
          else
          {
            values[j].x = g_occ[2 * i];
            values[j].y = (2 * i + 1 < uint32_t(mc.region_voxels)) ? g_occ[2 * i + 1] : 0.0f;
          }
        }
      }
#pragma unroll
      for (uint32_t j = 0; j < kWordsPerThread; ++j)
      {
        const uint32_t i = threadIdx.x + (half * kWordsPerThread + j) * uint32_t(G::kThreads);
        const uint32_t w = words[j];
        if (w)
        {
          const uint32_t n0 = w & kTileCountMask;
          const uint32_t n1 = (w >> 16) & kTileCountMask;
          // (a voxel at its clamp -- most of a settled map's free space -- does not move: nothing to write)
          if (n0)
          {
            const float v = occMissN(mc, args.ray_flags, values[j].x, n0);
            if (v != values[j].x)
            {
              g_occ[2 * i] = v;
            }
          }
          if (n1)
          {
            const float v = occMissN(mc, args.ray_flags, values[j].y, n1);
            if (v != values[j].y)
            {
              g_occ[2 * i + 1] = v;
            }
          }
        }
      }
      if (stamp && chunk_index < kTraceChunks)
      {
        args.dbg_counters[16 + size_t(chunk_index) * kTraceWords + 25 + half] = wall_clock64();
      }
      }  // halves
      if (inline_hits)
      {
        // Ordered replay of the region's samples, one lane per voxel with samples (the head of its run in the sorted
        // list): misses before each sample from the interval counters, the sample, the trailing misses from the tile.
        // These voxels are disjoint from the ones the passes above wrote.
        for (uint32_t i = threadIdx.x; i < n_region_hits; i += uint32_t(G::kThreads))
        {
          const unsigned long long key = l_hits[i];
          const unsigned long long group = key >> kHitRayBits;
          if (i > 0u && (l_hits[i - 1u] >> kHitRayBits) == group)
          {
            continue;
          }
          const uint32_t vi = uint32_t(group) & ((1u << kHitVoxelBits) - 1u);
          float x = g_occ[vi];
          for (uint32_t j = i; j < n_region_hits && (l_hits[j] >> kHitRayBits) == group; ++j)
          {
            x = occMissN(mc, args.ray_flags, x, (l_intervals[j >> 1] >> ((j & 1u) * 16u)) & 0xffffu);
            x = occHit(mc, args.ray_flags, x);
          }
          const uint32_t w = l_counts[tileWord(vi >> 1)];
          x = occMissN(mc, args.ray_flags, x, (w >> ((vi & 1u) * 16u)) & kTileCountMask);
          g_occ[vi] = x;
        }
        if (stamp && chunk_index < kTraceChunks)
        {
          args.dbg_counters[16 + size_t(chunk_index) * kTraceWords + 27] = wall_clock64();
        }
        if (threadIdx.x == 0)
        {
          atomicOr(&args.bs.hit_begin[chunk.slot], kSamplesApplied);
        }
      }
      if (stamp && chunk_index < kTraceChunks)
      {
        args.dbg_counters[16 + size_t(chunk_index) * kTraceWords + 19] = wall_clock64();
      }
      __syncthreads();
      cur = next;
      next = after_next;
      continue;
    }
    if (args.tsdf && (chunk.hash_index & 0x80000000u))
    {
      // TSDF, region held by this one chunk: voxels that only saw free-space visits (count n, not flagged) end at
      // weight = min(weight + n, max_weight), distance = truncation distance (see k_apply_counts_tsdf) -- applied here
      // straight from LDS; flagged voxels are replayed from their events.
      float2 *g_tsdf = reinterpret_cast<float2 *>(args.tsdf) + size_t(chunk.slot) * size_t(mc.region_voxels);
      constexpr uint32_t kTsdfBatch = 4;
      for (uint32_t first = threadIdx.x; first < count_words; first += kTsdfBatch * uint32_t(G::kThreads))
      {
        uint32_t words[kTsdfBatch];
        float2 values[2 * kTsdfBatch];
#pragma unroll
        for (uint32_t j = 0; j < kTsdfBatch; ++j)
        {
          const uint32_t i = first + j * uint32_t(G::kThreads);
          uint32_t w = (i < count_words) ? l_counts[tileWord(i)] : 0u;
          w = (w & kTileFlag) ? (w & 0xffff0000u) : w;
          w = (w & (kTileFlag << 16)) ? (w & 0x0000ffffu) : w;
          words[j] = w;
          values[2 * j] = make_float2(0.0f, 0.0f);
          values[2 * j + 1] = make_float2(0.0f, 0.0f);
          if (w & kTileCountMask)
          {
            values[2 * j] = g_tsdf[2 * i];
          }
          if ((w >> 16) & kTileCountMask)
          {
            values[2 * j + 1] = g_tsdf[2 * i + 1];
          }
        }
#pragma unroll
        for (uint32_t j = 0; j < kTsdfBatch; ++j)
        {
          const uint32_t i = first + j * uint32_t(G::kThreads);
          const uint32_t n0 = words[j] & kTileCountMask;
          const uint32_t n1 = (words[j] >> 16) & kTileCountMask;
          if (n0)
          {
            const float wn = values[2 * j].x + float(n0);
            g_tsdf[2 * i] = make_float2((mc.tsdf_max_weight < wn) ? mc.tsdf_max_weight : wn, mc.tsdf_trunc);
          }
          if (n1)
          {
            const float wn = values[2 * j + 1].x + float(n1);
            g_tsdf[2 * i + 1] = make_float2((mc.tsdf_max_weight < wn) ? mc.tsdf_max_weight : wn, mc.tsdf_trunc);
          }
        }
      }
      if (stamp && chunk_index < kTraceChunks)
      {
        args.dbg_counters[16 + size_t(chunk_index) * kTraceWords + 19] = wall_clock64();
      }
      __syncthreads();
      cur = next;
      next = after_next;
      continue;
    }
    // Flush the tile: integer adds, so the merge across chunks of one region is order independent.  (NDT / TSDF:
    // entries of masked voxels are skipped -- their visits travel as events.)
    for (uint32_t i = threadIdx.x; i < count_words; i += uint32_t(G::kThreads))
    {
      const uint32_t w = l_counts[tileWord(i)];
      if (w & (kTileCountMask | (kTileCountMask << 16)))
      {
#pragma unroll
        for (uint32_t half = 0; half < 2; ++half)
        {
          const uint32_t entry = (w >> (16u * half)) & 0xffffu;
          const uint32_t n = entry & kTileCountMask;
          if (n && !(defer_all && (entry & kTileFlag)))
          {
            atomicAdd(&g_counts[2 * i + half], n);
          }
        }
      }
    }
    if (stamp && chunk_index < kTraceChunks)
    {
      args.dbg_counters[16 + size_t(chunk_index) * kTraceWords + 19] = wall_clock64();
    }
    // The tile is reused by the next trip: everyone must be done reading it.
    __syncthreads();
    cur = next;
    next = after_next;
  }  // chunk loop
}

/// Resolve the deferred miss events (grid-stride; the event count lives in device memory).
__global__ void __launch_bounds__(256)
  k_flagged_events(BatchScratch bs, const unsigned long long *__restrict__ events, uint32_t event_capacity,
                   const uint32_t *__restrict__ event_count, const unsigned long long *__restrict__ sorted_hits,
                   uint32_t *__restrict__ miss_counts, uint32_t *__restrict__ interval_counts, int region_voxels,
                   uint32_t *__restrict__ host_event_count)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    *host_event_count = *event_count;  // pinned: sizes the next batch's event list
  }
  const uint32_t n = min(*event_count, event_capacity);
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
  {
    resolveFlaggedMiss(events[i], bs, sorted_hits, miss_counts, interval_counts, region_voxels);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_apply_hits: one lane per sorted hit; the first hit of each voxel group replays the whole group in ray order.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline void applyHits(uint32_t i, const MapConst &mc, const RegionTable &rt, const BatchScratch &bs,
                                 unsigned ray_flags, const unsigned long long *__restrict__ sorted,
                                 uint32_t *__restrict__ interval_counts, uint32_t *__restrict__ miss_counts,
                                 const double *__restrict__ rays, float *__restrict__ occupancy,
                                 uint32_t *__restrict__ mean, const SecondaryLayers &sec,
                                 const RayWalk *__restrict__ walks)
{
  const uint32_t n_hits = bs.info->n_hits;
  if (i >= n_hits)
  {
    return;
  }
  const unsigned long long key = sorted[i];
  const unsigned long long group = key >> kHitRayBits;  // slot | voxel
  if (i > 0 && (sorted[i - 1] >> kHitRayBits) == group)
  {
    return;  // not the head of its voxel group
  }
  const uint32_t slot = uint32_t(key >> kHitSlotShift);
  if (bs.hit_begin[slot] & kSamplesApplied)
  {
    return;  // the walk kernel replayed this region's samples itself (WalkArgs::inline_hits)
  }
  const uint32_t vi = uint32_t(key >> kHitRayBits) & ((1u << kHitVoxelBits) - 1u);
  const size_t gi = size_t(slot) * size_t(mc.region_voxels) + vi;
  float x = occupancy[gi];

  uint32_t mcoord = 0, mcount = 0;
  double centre[3] = { 0, 0, 0 };
  if (mean)
  {
    mcoord = mean[2 * gi];
    mcount = mean[2 * gi + 1];
    int16_t rk[3];
    unpackRegionKey(rt.slot_keys[slot], rk);
    const int lx = int(vi % uint32_t(mc.dim[0]));
    const int ly = int((vi / uint32_t(mc.dim[0])) % uint32_t(mc.dim[1]));
    const int lz = int(vi / uint32_t(mc.dim[0] * mc.dim[1]));
    centre[0] = globalVoxelCentreAxis(mc, 0, int(rk[0]) * mc.dim[0] + lx);
    centre[1] = globalVoxelCentreAxis(mc, 1, int(rk[1]) * mc.dim[1] + ly);
    centre[2] = globalVoxelCentreAxis(mc, 2, int(rk[2]) * mc.dim[2] + lz);
  }

  uint32_t packed_normal = sec.incident ? sec.incident[gi] : 0u;
  float traversal_add = 0.0f;
  float traversal = sec.traversal ? sec.traversal[gi] : 0.0f;
  uint32_t last_ray = 0;
  for (uint32_t j = i; j < n_hits && (sorted[j] >> kHitRayBits) == group; ++j)
  {
    x = occMissN(mc, ray_flags, x, interval_counts[j]);
    interval_counts[j] = 0;
    x = occHit(mc, ray_flags, x);
    const uint32_t ray = uint32_t(sorted[j] & ((1ull << kHitRayBits) - 1ull));
    last_ray = ray;
    if (sec.incident)
    {
      // ohm/RayMapperOccupancy.cpp:319-325: incident ray = start - end (converted to float), weight = sample count
      // before this sample (0 without a mean layer).
      const float dir[3] = { float(rays[size_t(ray) * 6 + 0] - rays[size_t(ray) * 6 + 3]),
                             float(rays[size_t(ray) * 6 + 1] - rays[size_t(ray) * 6 + 4]),
                             float(rays[size_t(ray) * 6 + 2] - rays[size_t(ray) * 6 + 5]) };
      packed_normal = updateIncidentNormal(packed_normal, dir, mean ? mcount : 0u);
    }
    if (sec.traversal)
    {
      // ohm/RayMapperOccupancy.cpp:299-305: remaining ray length inside the sample voxel.
      const double dx = rays[size_t(ray) * 6 + 3] - rays[size_t(ray) * 6 + 0];
      const double dy = rays[size_t(ray) * 6 + 4] - rays[size_t(ray) * 6 + 1];
      const double dz = rays[size_t(ray) * 6 + 5] - rays[size_t(ray) * 6 + 2];
      traversal += float(sqrt((dx * dx + dy * dy) + dz * dz) - lastExitRange(walks, ray));
    }
    if (mean)
    {
      // NOTE: uses the ray's sample as submitted; the clip filter never applies a hit to a moved end point.
      const double local[3] = { rays[size_t(ray) * 6 + 3] - centre[0], rays[size_t(ray) * 6 + 4] - centre[1],
                                rays[size_t(ray) * 6 + 5] - centre[2] };
      mcoord = subVoxelUpdate(mcoord, mcount, local, mc.resolution);
      ++mcount;
    }
  }
  // Misses after the last hit.
  x = occMissN(mc, ray_flags, x, miss_counts[gi]);
  miss_counts[gi] = 0;
  occupancy[gi] = x;
  if (mean)
  {
    mean[2 * gi] = mcoord;
    mean[2 * gi + 1] = mcount;
  }
  (void)traversal_add;
  if (sec.incident)
  {
    sec.incident[gi] = packed_normal;
  }
  if (sec.traversal)
  {
    sec.traversal[gi] = traversal;
  }
  if (sec.touch_time && sec.timestamps)
  {
    // The CPU mapper overwrites the touch time at every sample: the last sample in ray order wins
    // (ohm/RayMapperOccupancy.cpp:313-317).
    sec.touch_time[gi] = encodeVoxelTouchTime(sec.time_base, sec.timestamps[last_ray]);
  }
}

__global__ void __launch_bounds__(256)
  k_apply_hits(MapConst mc, RegionTable rt, BatchScratch bs, unsigned ray_flags,
               const unsigned long long *__restrict__ sorted, uint32_t *__restrict__ interval_counts,
               uint32_t *__restrict__ miss_counts, const double *__restrict__ rays, float *__restrict__ occupancy,
               uint32_t *__restrict__ mean, SecondaryLayers sec, const RayWalk *__restrict__ walks)
{
  applyHits(blockIdx.x * blockDim.x + threadIdx.x, mc, rt, bs, ray_flags, sorted, interval_counts, miss_counts, rays,
            occupancy, mean, sec, walks);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_apply_counts: one block per touched region: apply plain miss counts, clear scratch.
// ---------------------------------------------------------------------------------------------------------------------
/// `skip_masked`: voxels whose mask bit is set are somebody else's (NDT: the ordered replay; occupancy, when this runs
/// beside applyHits in one launch: applyHits) -- their counts are neither applied nor touched.
__device__ inline void applyCounts(uint32_t region_index, const MapConst &mc, const RegionTable &rt,
                                   const BatchScratch &bs, unsigned ray_flags, uint32_t *__restrict__ miss_counts,
                                   uint32_t *__restrict__ hit_mask, float *__restrict__ occupancy, int clear_mask,
                                   uint32_t *__restrict__ hit_miss_counts, uint32_t direct_chunk_segments,
                                   int skip_masked, int preserve_masked, float *__restrict__ traversal = nullptr,
                                   unsigned long long *__restrict__ traversal_acc = nullptr)
{
  const uint32_t h = bs.touched[region_index];
  const uint32_t slot = rt.vals[h];
  const size_t base = size_t(slot) * size_t(mc.region_voxels);
  // Regions with a single chunk were applied by the walk kernel itself (direct_chunk_segments != 0).
  const bool applied_by_walk = direct_chunk_segments && bs.seg_count[h] > 0 && bs.seg_count[h] <= direct_chunk_segments;
  if (!applied_by_walk)
  {
    // 4 voxels per lane per step: 16-byte loads of the count and log-odds layers (region blocks are 16-byte aligned
    // whenever region_voxels is a multiple of 4; otherwise fall back to scalar accesses).
    const uint32_t n4 = (mc.region_voxels % 4 == 0) ? uint32_t(mc.region_voxels) / 4u : 0u;
    uint4 *counts4 = reinterpret_cast<uint4 *>(miss_counts + base);
    float4 *occ4 = reinterpret_cast<float4 *>(occupancy + base);
    const uint32_t *mask = hit_mask + size_t(slot) * (uint32_t(mc.region_voxels + 31) >> 5);
    for (uint32_t q = threadIdx.x; q < n4; q += blockDim.x)
    {
      uint4 n = counts4[q];
      if (n.x | n.y | n.z | n.w)
      {
        uint32_t bits = 0;
        if (skip_masked)
        {
          // NDT: voxels holding samples are updated by the ordered replay only; their tile entry is not used.
          bits = (mask[(4 * q) >> 5] >> ((4 * q) & 31)) & 15u;
          n.x = (bits & 1u) ? 0u : n.x;
          n.y = (bits & 2u) ? 0u : n.y;
          n.z = (bits & 4u) ? 0u : n.z;
          n.w = (bits & 8u) ? 0u : n.w;
        }
        const float4 before = occ4[q];
        float4 o = before;
        o.x = n.x ? occMissN(mc, ray_flags, o.x, n.x) : o.x;
        o.y = n.y ? occMissN(mc, ray_flags, o.y, n.y) : o.y;
        o.z = n.z ? occMissN(mc, ray_flags, o.z, n.z) : o.z;
        o.w = n.w ? occMissN(mc, ray_flags, o.w, n.w) : o.w;
        // (voxels at their clamp do not move: a settled map's free space needs no log-odds write)
        const bool moved = o.x != before.x || o.y != before.y || o.z != before.z || o.w != before.w;
        if (preserve_masked && bits)
        {
          // A masked voxel's log-odds and count belong to applyHits, which may be at work on them right now: only the
          // other voxels of the group are written.
          float *o1 = occupancy + base + 4 * q;
          uint32_t *c1 = miss_counts + base + 4 * q;
          if (!(bits & 1u)) { o1[0] = o.x; c1[0] = 0; }
          if (!(bits & 2u)) { o1[1] = o.y; c1[1] = 0; }
          if (!(bits & 4u)) { o1[2] = o.z; c1[2] = 0; }
          if (!(bits & 8u)) { o1[3] = o.w; c1[3] = 0; }
        }
        else
        {
          if (moved)
          {
            occ4[q] = o;
          }
          counts4[q] = make_uint4(0, 0, 0, 0);
        }
        if (hit_miss_counts)
        {
          // NDT-TM: every plain miss increments HitMissCount::miss_count (ohm/RayMapperNdt.cpp:171-178).
          hit_miss_counts[2 * (base + 4 * q + 0) + 1] += n.x;
          hit_miss_counts[2 * (base + 4 * q + 1) + 1] += n.y;
          hit_miss_counts[2 * (base + 4 * q + 2) + 1] += n.z;
          hit_miss_counts[2 * (base + 4 * q + 3) + 1] += n.w;
        }
      }
    }
    for (uint32_t vi = 4 * n4 + threadIdx.x; vi < uint32_t(mc.region_voxels); vi += blockDim.x)
    {
      uint32_t n = miss_counts[base + vi];
      if (n && skip_masked && ((mask[vi >> 5] >> (vi & 31)) & 1u))
      {
        if (!preserve_masked)
        {
          miss_counts[base + vi] = 0;
        }
        n = 0;
      }
      if (n)
      {
        occupancy[base + vi] = occMissN(mc, ray_flags, occupancy[base + vi], n);
        miss_counts[base + vi] = 0;
        if (hit_miss_counts)
        {
          hit_miss_counts[2 * (base + vi) + 1] += n;
        }
      }
    }
  }
  if (traversal_acc)
  {
    // This batch's ray lengths through the region's voxels (k_region_traversal, traversal_kernels.h), summed exactly.
    for (uint32_t vi = threadIdx.x; vi < uint32_t(mc.region_voxels); vi += blockDim.x)
    {
      const unsigned long long sum = traversal_acc[base + vi];
      if (sum)
      {
        traversal[base + vi] = float(double(traversal[base + vi]) + double(sum) * (1.0 / kTraversalScale));
        traversal_acc[base + vi] = 0;
      }
    }
  }
  if (clear_mask)
  {
    __syncthreads();  // (the loops above read the mask)
    const uint32_t mask_words = uint32_t(mc.region_voxels + 31) >> 5;
    for (uint32_t i = threadIdx.x; i < mask_words; i += blockDim.x)
    {
      hit_mask[size_t(slot) * mask_words + i] = 0;
    }
  }
  if (threadIdx.x == 0)
  {
    bs.seg_count[h] = 0;
    bs.seg_cursor[h] = 0;
    bs.touched_flag[h] = 0;
  }
}

__global__ void __launch_bounds__(1024)
  k_apply_counts(MapConst mc, RegionTable rt, BatchScratch bs, unsigned ray_flags, uint32_t *__restrict__ miss_counts,
                 uint32_t *__restrict__ hit_mask, float *__restrict__ occupancy, int clear_mask,
                 uint32_t *__restrict__ hit_miss_counts, uint32_t direct_chunk_segments, int skip_masked,
                 float *__restrict__ traversal, unsigned long long *__restrict__ traversal_acc)
{
  applyCounts(blockIdx.x, mc, rt, bs, ray_flags, miss_counts, hit_mask, occupancy, clear_mask, hit_miss_counts,
              direct_chunk_segments, skip_masked, 0, traversal, traversal_acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// The apply phase of occupancy-only maps (round 5).  The walk kernel applies every region it holds in ONE chunk itself --
// counts and samples, straight from LDS -- so in the steady state of C1 the two kernels above found work in ~80 of
// 1243 regions and spent their 46 us on workgroups and lanes that looked at a region or a sample and left.  k_plan now
// lists the regions that DO need them (BatchScratch::apply_counts_list / apply_hits_list): the kernels below run over
// those lists only, and the per-region bookkeeping every touched region needs moves to k_batch_cleanup.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kApplyListParts = 16;  ///< workgroups sharing one listed region's count application (whole mask words each)

/// One launch for both lists (256-thread workgroups; the first hit_blocks of them replay samples, the others apply
/// counts): a listed region's samples and its plain counts touch DIFFERENT voxels -- the count part skips every voxel
/// whose sample-mask bit is set, those counts are the sample replay's (which applies and clears them) -- so the two run
/// side by side instead of one behind the other (22 + 17 us as two launches: both are chains of dependent loads).
///   sample part: workgroup (r, p) replays samples [256 p, 256 p + 256) of listed region r -- one lane per sample, the
///     head of a voxel's run replays the run (applyHits); blocks_per_region covers the batch's densest region
///   count part: workgroup (r, p) applies 1 / kApplyListParts of listed region r's counts (whole mask words)
__global__ void __launch_bounds__(256)
  k_apply_lists(MapConst mc, RegionTable rt, BatchScratch bs, unsigned ray_flags,
                const unsigned long long *__restrict__ sorted, uint32_t *__restrict__ interval_counts,
                uint32_t *__restrict__ miss_counts, const uint32_t *__restrict__ hit_mask,
                const double *__restrict__ rays, float *__restrict__ occupancy, uint32_t hit_blocks,
                uint32_t blocks_per_region, unsigned long long hit_tiles)
{
  if (blockIdx.x < hit_blocks)
  {
    // (region, 256-sample tile) pairs, strided over the sample part's workgroups: hit_blocks == hit_tiles except for
    // skewed multi-million-ray batches, whose product of listed regions and tiles of the densest one is capped by the
    // host (the product used to be the grid, in 32 bits).
    for (unsigned long long tile = blockIdx.x; tile < hit_tiles; tile += hit_blocks)
    {
      const uint32_t slot = rt.vals[bs.apply_hits_list[uint32_t(tile / blocks_per_region)]];
      if (slot >= rt.slot_capacity)
      {
        continue;
      }
      const uint32_t begin = bs.hit_begin[slot] & ~kSamplesApplied;
      const uint32_t i = begin + uint32_t(tile % blocks_per_region) * 256u + threadIdx.x;
      if (i < bs.hit_end[slot])
      {
        SecondaryLayers none{};
        applyHits(i, mc, rt, bs, ray_flags, sorted, interval_counts, miss_counts, rays, occupancy, nullptr, none,
                  nullptr);
      }
    }
    return;
  }
  const uint32_t block = blockIdx.x - hit_blocks;
  const uint32_t h = bs.apply_counts_list[block / kApplyListParts];
  const uint32_t part = block % kApplyListParts;
  const uint32_t slot = rt.vals[h];
  if (slot >= rt.slot_capacity)
  {
    return;
  }
  const size_t base = size_t(slot) * size_t(mc.region_voxels);
  const uint32_t words = uint32_t(mc.region_voxels + 31) >> 5;
  const uint32_t *mask = hit_mask + size_t(slot) * words;
  const uint32_t v_lo = uint32_t(uint64_t(words) * part / kApplyListParts) * 32u;
  const uint32_t v_hi = min(uint32_t(uint64_t(words) * (part + 1u) / kApplyListParts) * 32u, uint32_t(mc.region_voxels));
  if (mc.region_voxels % 4 == 0)
  {
    uint4 *counts4 = reinterpret_cast<uint4 *>(miss_counts + base);
    float4 *occ4 = reinterpret_cast<float4 *>(occupancy + base);
    for (uint32_t q = v_lo / 4u + threadIdx.x; q < v_hi / 4u; q += blockDim.x)
    {
      uint4 n = counts4[q];
      if (n.x | n.y | n.z | n.w)
      {
        const uint32_t bits = (mask[(4u * q) >> 5] >> ((4u * q) & 31u)) & 15u;
        n.x = (bits & 1u) ? 0u : n.x;
        n.y = (bits & 2u) ? 0u : n.y;
        n.z = (bits & 4u) ? 0u : n.z;
        n.w = (bits & 8u) ? 0u : n.w;
        const float4 before = occ4[q];
        float4 o = before;
        o.x = n.x ? occMissN(mc, ray_flags, o.x, n.x) : o.x;
        o.y = n.y ? occMissN(mc, ray_flags, o.y, n.y) : o.y;
        o.z = n.z ? occMissN(mc, ray_flags, o.z, n.z) : o.z;
        o.w = n.w ? occMissN(mc, ray_flags, o.w, n.w) : o.w;
        // (only the voxels that are this part's: a masked voxel's log-odds and count may be under the sample replay's
        // hands right now)
        float *o1 = occupancy + base + 4 * q;
        uint32_t *c1 = miss_counts + base + 4 * q;
        if (n.x) { if (o.x != before.x) { o1[0] = o.x; } c1[0] = 0; }
        if (n.y) { if (o.y != before.y) { o1[1] = o.y; } c1[1] = 0; }
        if (n.z) { if (o.z != before.z) { o1[2] = o.z; } c1[2] = 0; }
        if (n.w) { if (o.w != before.w) { o1[3] = o.w; } c1[3] = 0; }
      }
    }
    return;
  }
  for (uint32_t vi = v_lo + threadIdx.x; vi < v_hi; vi += blockDim.x)
  {
    const uint32_t n = miss_counts[base + vi];
    if (n && !((mask[vi >> 5] >> (vi & 31u)) & 1u))
    {
      occupancy[base + vi] = occMissN(mc, ray_flags, occupancy[base + vi], n);
      miss_counts[base + vi] = 0;
    }
  }
}

/// Every touched region back to its idle state: the sample mask cleared, the per-batch scratch reset (what applyCounts
/// does for a region when k_apply_counts runs over all of them).
__global__ void __launch_bounds__(256)
  k_batch_cleanup(MapConst mc, RegionTable rt, BatchScratch bs, uint32_t n_touched, uint32_t *__restrict__ hit_mask)
{
  const uint32_t mask_words = uint32_t(mc.region_voxels + 31) >> 5;
  for (uint32_t r = blockIdx.x; r < n_touched; r += gridDim.x)
  {
    const uint32_t h = bs.touched[r];
    const uint32_t slot = rt.vals[h];
    if (slot < rt.slot_capacity)
    {
      for (uint32_t w = threadIdx.x; w < mask_words; w += blockDim.x)
      {
        hit_mask[size_t(slot) * mask_words + w] = 0;
      }
    }
    if (threadIdx.x == 0)
    {
      bs.seg_count[h] = 0;
      bs.seg_cursor[h] = 0;
      bs.touched_flag[h] = 0;
    }
  }
}

/// dst[i] &= mask
__global__ void k_and_u32(uint32_t *dst, uint32_t mask, size_t count)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
  {
    dst[i] &= mask;
  }
}

/// dst[index[i]] |= bits  (indices may repeat)
__global__ void k_or_at_u32(uint32_t *dst, const uint32_t *__restrict__ index, size_t count, uint32_t bits)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
  {
    atomicOr(&dst[index[i]], bits);
  }
}

/// A list of independent byte copies done by ONE launch: pool slot <-> pinned host record (the device reads / writes
/// the mapped host memory itself, so an eviction or re-admission of hundreds of regions is a single PCIe-saturating
/// kernel instead of one copy-engine call per region and layer: measured 10 GB/s with the calls, their per-call
/// overhead dominating 256 KiB copies), or slot -> slot inside the pool (compaction).
struct CopyJob
{
  const char *src;
  char *dst;
  uint64_t bytes;
};

constexpr uint32_t kCopyBlocksPerJob = 16;

__global__ void __launch_bounds__(256) k_copy_jobs(const CopyJob *__restrict__ jobs, uint32_t n_jobs)
{
  const uint32_t job_index = blockIdx.x / kCopyBlocksPerJob;
  const uint32_t part = blockIdx.x % kCopyBlocksPerJob;
  if (job_index >= n_jobs)
  {
    return;
  }
  const CopyJob job = jobs[job_index];
  const bool aligned = ((reinterpret_cast<uintptr_t>(job.src) | reinterpret_cast<uintptr_t>(job.dst)) & 15u) == 0;
  const uint64_t vectors = aligned ? job.bytes / 16u : 0u;
  const uint4 *src = reinterpret_cast<const uint4 *>(job.src);
  uint4 *dst = reinterpret_cast<uint4 *>(job.dst);
  // (interleaved over the job's blocks so that the blocks of a job stream neighbouring lines)
  for (uint64_t i = uint64_t(part) * 256u + threadIdx.x; i < vectors; i += uint64_t(kCopyBlocksPerJob) * 256u)
  {
    dst[i] = src[i];
  }
  if (part == 0)
  {
    for (uint64_t i = vectors * 16u + threadIdx.x; i < job.bytes; i += 256u)
    {
      job.dst[i] = job.src[i];
    }
  }
}

/// The same list of copies done by a SMALL persistent grid (the background write-back of the spill path): `gridDim.x`
/// workgroups walk the (job, part) pairs with a stride, so the launch holds at most that many CUs while batches run --
/// the walk kernel needs whole CUs, and a flood of short copy workgroups over all of them stalls it.
__global__ void __launch_bounds__(256) k_copy_jobs_few(const CopyJob *__restrict__ jobs, uint32_t n_jobs)
{
  const uint32_t units = n_jobs * kCopyBlocksPerJob;
  for (uint32_t unit = blockIdx.x; unit < units; unit += gridDim.x)
  {
    const CopyJob job = jobs[unit / kCopyBlocksPerJob];
    const uint32_t part = unit % kCopyBlocksPerJob;
    const bool aligned = ((reinterpret_cast<uintptr_t>(job.src) | reinterpret_cast<uintptr_t>(job.dst)) & 15u) == 0;
    const uint64_t vectors = aligned ? job.bytes / 16u : 0u;
    const uint4 *src = reinterpret_cast<const uint4 *>(job.src);
    uint4 *dst = reinterpret_cast<uint4 *>(job.dst);
    for (uint64_t i = uint64_t(part) * 256u + threadIdx.x; i < vectors; i += uint64_t(kCopyBlocksPerJob) * 256u)
    {
      dst[i] = src[i];
    }
    if (part == 0)
    {
      for (uint64_t i = vectors * 16u + threadIdx.x; i < job.bytes; i += 256u)
      {
        job.dst[i] = job.src[i];
      }
    }
  }
}

/// use[2 * index[i]] is touched with `stamp` (touchRegionUse)
__global__ void k_touch_use_at(uint32_t *use, const uint32_t *__restrict__ index, size_t count, uint32_t stamp)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
  {
    touchRegionUse(use, index[i], stamp);
  }
}

/// pairs = (slot, stamp): the slot's "use before the gap" becomes stamp (a region back from the host store)
__global__ void k_set_prev_use(uint32_t *use, const uint32_t *__restrict__ pairs, size_t count)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
  {
    use[2 * size_t(pairs[2 * i]) + 1] = pairs[2 * i + 1];
  }
}

/// Fill a float layer with a value (pool initialisation: occupancy clears to +inf, ohm/DefaultLayer.cpp:87-91).
__global__ void k_fill_u32(uint32_t *dst, uint32_t value, size_t count)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
  {
    dst[i] = value;
  }
}
}  // namespace ohmhip

#endif  // OHMHIP_OCCUPANCY_KERNELS_H
