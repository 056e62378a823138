// Traversal layer: the length of every ray inside every voxel it visits, summed per voxel
// (ohm/RayMapperOccupancy.cpp:166-173, ohmgpu/gpu/RegionUpdate.cl: the `traversal` argument of the voxel visit).
//
//   k_region_traversal   1 workgroup / chunk   a second, fp64 walk of the chunk's segments into a 32-bit LDS tile
//
// The count walk (k_region_walk) decides its steps from a fixed-point predictor and never knows a step's time; the
// traversal layer needs exactly that -- exit range minus enter range of every visit -- so maps with the layer run this
// kernel after it, over the same chunk list.  Every lane repeats the reference's fp64 walk of its segment
// (ohm/LineWalkCompute.h:282-307: time_next recomputed from the step count, smallest wins, ties to the higher axis) and
// adds float(exit - enter), as the CPU mapper does, to the voxel's word of an LDS tile in fixed point.  No global atomic
// per visit (round 2: one 64-bit global atomic per visit, 24.7 ms for C1):
//   * the tile word counts units of 2^-unit_bits metres (2^-28 m = 3.7 nm unless the voxels are metres wide), rounded to
//     nearest per visit; the LDS add returns the word's old value and that is looked at after the step arithmetic, so
//     the LDS round trip is covered (measured: 795 -> 705 us for C1);
//   * a word that wraps sends one carry of 2^32 units to the voxel's 64-bit global accumulator (rare: a 0.1 m visit is
//     2.7e7 units, so one visit in ~160 carries);
//   * the tile is flushed into the same accumulators when the chunk is done; applyCounts folds them into the float layer
//     once per batch.  Integer sums: the result does not depend on the order of chunks, lanes or carries.
#ifndef OHMHIP_TRAVERSAL_KERNELS_H
#define OHMHIP_TRAVERSAL_KERNELS_H

#include "occupancy_kernels.h"

namespace ohmhip
{
constexpr int kTraversalUnroll = 2;  ///< walk steps per loop trip

struct TraversalArgs
{
  MapConst mc;
  const Chunk *chunks;
  const Segment *segments;
  const RayWalk *walks;
  const uint64_t *slot_keys;
  unsigned long long *traversal_acc;  ///< [slot][voxel] sums of this batch in units of 1 / kTraversalScale metres
  int unit_bits;                      ///< tile unit = 2^-unit_bits metres, <= 28 (traversalUnitBits)
  int refill_min_idle;
  /// Persistent workgroups (round 6), like k_region_walk: one per CU, the first chunk is the workgroup's own index, the
  /// following ones come from this device-wide cursor (zeroed by k_plan) -- the list is ordered largest first, so the
  /// launch ends on its smallest chunks instead of on whatever the hardware's round-robin over the XCDs left for last.
  uint32_t *chunk_cursor;
  uint32_t n_chunks;
};

/// Largest tile unit exponent for which a single visit (at most a voxel diagonal long) stays below 2^31 units.
inline int traversalUnitBits(double resolution)
{
  int bits = 28;
  while (bits > 0 && resolution * 1.7320508075688772 * double(1ull << bits) >= 2147483648.0)
  {
    --bits;
  }
  return bits;
}

inline size_t traversalLdsBytes(const MapConst &mc)
{
  return (size_t(mc.region_voxels) + 16u) * sizeof(uint32_t);  // tile + [0] segment cursor, [1..8] idle words, [9] next chunk
}

__global__ void __launch_bounds__(kWalkThreads) k_region_traversal(TraversalArgs args)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t l_tile[];
  const MapConst &mc = args.mc;
  const uint32_t n_voxels = uint32_t(mc.region_voxels);
  uint32_t *l_cursor = l_tile + n_voxels;
  for (uint32_t chunk_index = blockIdx.x; chunk_index < args.n_chunks;)
  {
  const Chunk chunk = args.chunks[chunk_index];
  for (uint32_t i = threadIdx.x; i < n_voxels; i += blockDim.x)
  {
    l_tile[i] = 0;
  }
  if (threadIdx.x == 0)
  {
    *l_cursor = 0;
    l_cursor[9] = gridDim.x + atomicAdd(args.chunk_cursor, 1u);  // the chunk after this one (fetched under the walk)
  }
  __syncthreads();

  int16_t rk[3];
  unpackRegionKey(args.slot_keys[chunk.slot], rk);
  const int region_x = rk[0], region_y = rk[1], region_z = rk[2];
  const uint32_t n_seg = chunk.seg_end - chunk.seg_begin;
  const Segment *chunk_segments = args.segments + chunk.seg_begin;
  unsigned long long *acc = args.traversal_acc + size_t(chunk.slot) * size_t(n_voxels);
  const float unit_scale = float(1ull << args.unit_bits);
  const uint32_t idle_address = (n_voxels + 1u + (threadIdx.x & 7u)) << 2;  // spare words behind the tile and the cursor
  const unsigned long long carry = 1ull << (32 + 40 - args.unit_bits);
  const int dimx = mc.dim[0];
  const int dimxy = mc.dim[0] * mc.dim[1];
  const uint32_t lane = threadIdx.x & 63u;
  const double inf = dInf();

  // Lane state: the reference's walk of one segment.
  int left = 0;
  uint32_t vi = 0;
  int sx = 0, sy = 0, sz = 0;
  int k0 = 0, k1 = 0, k2 = 0, tot0 = 0, tot1 = 0, tot2 = 0;
  double i0 = 0, i1 = 0, i2 = 0, e0 = 0, e1 = 0, e2 = 0;
  double t0 = 0, t1 = 0, t2 = 0, t_enter = 0, ray_len = 0;
  uint32_t skip = 0, end_last = 0;
  bool exhausted = false;
  int refill_threshold = args.refill_min_idle;
  while (true)
  {
    // ---- refill idle lanes (wave-uniform decision, as in k_region_walk) ------------------------------------------------
    const unsigned long long am = __ballot(left > 0);
    const int n_idle = 64 - __popcll(am);
    if (__builtin_expect(n_idle >= refill_threshold, 0))
    {
      if (exhausted)
      {
        break;
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
        const uint4 *rec = reinterpret_cast<const uint4 *>(chunk_segments + mine);
        const uint32_t vox = rec[0].w;
        const uint4 rb = rec[1];
        vi = vox & ((1u << kSegVoxelBits) - 1u);
        left = int(vox >> kSegVoxelBits);
        sx = (rb.x & kSegNegative) ? -1 : 1;
        sy = (rb.y & kSegNegative) ? -dimx : dimx;
        sz = (rb.z & kSegNegative) ? -dimxy : dimxy;
        skip = (rb.w & kSegSkipFirst) ? 1u : 0u;
        end_last = (rb.w & kSegEnd) ? 1u : 0u;
        const RayWalk rw = args.walks[rb.w & kSegRayMask];
        stepsAtVoxel(mc, rw, region_x, region_y, region_z, vi, k0, k1, k2);
        i0 = rw.init[0];
        i1 = rw.init[1];
        i2 = rw.init[2];
        e0 = rw.delta[0];
        e1 = rw.delta[1];
        e2 = rw.delta[2];
        tot0 = rw.total[0];
        tot1 = rw.total[1];
        tot2 = rw.total[2];
        t0 = timeNext(i0, e0, k0, tot0);
        t1 = timeNext(i1, e1, k1, tot1);
        t2 = timeNext(i2, e2, k2, tot2);
        // The step which entered this region is the latest step taken so far.
        double te = (k0 > 0) ? stepTime(i0, e0, k0) : 0.0;
        const double te1 = (k1 > 0) ? stepTime(i1, e1, k1) : 0.0;
        const double te2 = (k2 > 0) ? stepTime(i2, e2, k2) : 0.0;
        te = (te1 > te) ? te1 : te;
        te = (te2 > te) ? te2 : te;
        t_enter = te;
        ray_len = rw.length;
      }
    }

    uint32_t olds[kTraversalUnroll], adds[kTraversalUnroll], visited[kTraversalUnroll];
#pragma unroll
    for (int u = 0; u < kTraversalUnroll; ++u)
    {
      // ---- one voxel: exit range == time of the next step (the ray's length at an end voxel that is part of the
      // ---- walk); kRfExcludeOrigin drops the first voxel of the ray (its range still passes).  A lane with nothing to
      // ---- visit adds zero to a spare word: no exec-mask juggling around the LDS operation.
      const bool active = left > 0;
      const bool at_end = end_last && left == 1;
      const bool visit = active && (at_end || !skip);
      skip = 0;
      const unsigned long long m01 = __builtin_amdgcn_fcmp(t0, t1, kFcmpOlt);
      const double t01 = selectD(m01, t0, t1);
      const unsigned long long m2 = __builtin_amdgcn_fcmp(t01, t2, kFcmpOlt);
      double t_exit = selectD(m2, t01, t2);
      t_exit = at_end ? ray_len : t_exit;
      // (the CPU mapper adds float(exit - enter): that float is what is summed.  The scale is a power of two, so the
      // product is exact and the conversion rounds to the nearest unit.)
      const uint32_t add = visit ? __float2uint_rn(float(t_exit - t_enter) * unit_scale) : 0u;
      adds[u] = add;
      visited[u] = vi;
      olds[u] = tileAdd(visit ? (vi << 2) : idle_address, add);
      t_enter = active ? t_exit : t_enter;
      // ---- the reference's step, branch free, taken by every lane (an idle lane's state is dead, and the step after a
      // ---- segment's last voxel is never used).
      const unsigned long long a0 = m2 & m01;
      const unsigned long long a1 = m2 & ~m01;
      const unsigned long long a2 = ~m2;
      k0 = addMask(k0, a0);
      k1 = addMask(k1, a1);
      k2 = addMask(k2, a2);
      const unsigned long long g0 = __builtin_amdgcn_sicmp(k0, tot0, kIcmpSlt);
      const unsigned long long g1 = __builtin_amdgcn_sicmp(k1, tot1, kIcmpSlt);
      const unsigned long long g2 = __builtin_amdgcn_sicmp(k2, tot2, kIcmpSlt);
      const double n0 = selectD(g0, i0 + e0 * double(k0), inf);
      const double n1 = selectD(g1, i1 + e1 * double(k1), inf);
      const double n2 = selectD(g2, i2 + e2 * double(k2), inf);
      t0 = selectD(a0, n0, t0);
      t1 = selectD(a1, n1, t1);
      t2 = selectD(a2, n2, t2);
      vi += uint32_t(selectI(a2, sz, selectI(a0, sx, sy)));
      left -= 1;
    }
    // ---- the returned words are looked at after the trip's last step: the LDS round trips are covered by the step
    // ---- arithmetic (waitTile carries the s_waitcnt).
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < kTraversalUnroll; ++u)
    {
      olds[u] = waitTile(olds[u]);
    }
#pragma unroll
    for (int u = 0; u < kTraversalUnroll; ++u)
    {
      if (__builtin_expect(uint32_t(olds[u] + adds[u]) < adds[u], 0))
      {
        atomicAdd(&acc[visited[u]], carry);  // the tile word wrapped
      }
    }
  }

  __syncthreads();
  const int flush_shift = 40 - args.unit_bits;
  for (uint32_t i = threadIdx.x; i < n_voxels; i += blockDim.x)
  {
    const uint32_t sum = l_tile[i];
    if (sum)
    {
      atomicAdd(&acc[i], (unsigned long long)(sum) << flush_shift);
    }
  }
  chunk_index = __builtin_amdgcn_readfirstlane(l_cursor[9]);
  __syncthreads();  // (the tile and the cursor words are reused by the next chunk)
  }  // chunks
}
}  // namespace ohmhip

#endif  // OHMHIP_TRAVERSAL_KERNELS_H
