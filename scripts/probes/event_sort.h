// event_sort.h -- PROTOTYPE, measured and not adopted (profiles/r04_sort_probe.txt): an ordering of the NDT /
// TSDF event keys of a batch (ohm_amd/csrc/replay_kernels.h) for event lists of up to a few million keys -- a
// least-significant-digit radix sort written for this key layout, meant to replace rocPRIM's one-sweep sort where the list
// is small enough to be cut into at most 512 workgroup tiles.  Correct on every case of
// scripts/probes/evsort_probe.hip, and no faster than the library where it matters: 0.174 ms against 0.187 ms on C2's
// 1.3 M keys -- materialising a histogram per tile (4096 digits x 212 tiles per pass) costs what the look-back fills of
// the one-sweep sort cost.  Kept next to its probe as the record of the attempt.
//
// Why: on a list of C2's size (1.3 M keys) the library sort is latency bound -- 7 digit passes of ~20 us, each preceded
// by two ~5 us fills of its look-back state, plus the histogram kernels: ~240 us of a 1.1 ms batch
// (profiles/r04_profile_c2_ndt.txt).  The keys are [slot | voxel:15 | ray:28 | sample bit]; a batch of n rays uses only
// ceil(log2 n) + 1 of the low 29 bits, so the digits are laid over the bits that can differ (C2: 21 + 24 bits = four
// passes of 10-12 bits) and a pass is three small launches with no fills:
//   k_evsort_rank     per tile (512 threads x K keys, in registers): every key's stable rank among the keys of its digit
//                     inside the tile -- waves match equal digits with ballots (no LDS atomics), per-wave digit counters
//                     in LDS, then a prefix over the waves; the tile's digit histogram goes to hist[digit][tile]
//   k_evsort_scan     per digit: exclusive prefix over the tiles (one wave per digit) and the digit's total
//   k_evsort_scatter  per tile: exclusive scan of the digit totals in LDS + the tile's prefix = the digit's first output
//                     position for this tile; key -> dst[base[digit] + rank]
// Stable, so the passes compose; invalid keys (all ones) carry the largest digit in every pass and end up last, as
// with the library sort.  Larger lists (C3: 24 M keys) stay with rocPRIM, whose tuned configuration moves them at
// 2.7 TB/s (profiles/r04_sort_probe.txt).
#ifndef OHMHIP_EVENT_SORT_H
#define OHMHIP_EVENT_SORT_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ohmhip
{
constexpr int kEvSortThreads = 512;
constexpr int kEvSortWaves = kEvSortThreads / 64;
constexpr uint32_t kEvSortMaxBits = 12;     ///< widest digit: 4096 bins, 64 KiB of u16 wave counters
constexpr uint32_t kEvSortMaxTiles = 512;   ///< tiles of a list (one wave scans a digit's tile counts: 8 per lane)
constexpr uint32_t kEvSortMaxKeysPerThread = 24;

/// Stable rank of every key among the keys of its digit inside its tile (u16: a tile holds at most 12 288 keys), and the
/// tile's digit histogram.  Key i of the tile belongs to wave w = i / (64 K), round j, lane l: consecutive keys sit in
/// consecutive lanes, rounds follow each other, waves follow each other -- tile order.
template <int K>
__global__ void __launch_bounds__(kEvSortThreads)
  k_evsort_rank(const unsigned long long *__restrict__ src, uint32_t n, uint32_t shift, uint32_t bits, uint32_t n_tiles,
                uint32_t *__restrict__ hist, uint16_t *__restrict__ ranks)
{
  extern __shared__ uint16_t evsort_counters[];  // [kEvSortWaves][bins]
  const uint32_t bins = 1u << bits;
  {
    uint32_t *words = reinterpret_cast<uint32_t *>(evsort_counters);
    for (uint32_t i = threadIdx.x; i < (uint32_t(kEvSortWaves) * bins) / 2u; i += kEvSortThreads)
    {
      words[i] = 0u;
    }
  }
  __syncthreads();
  const unsigned lane = __lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t wave_first = blockIdx.x * uint32_t(kEvSortThreads * K) + wave * uint32_t(64 * K);
  volatile uint16_t *mine = evsort_counters + wave * bins;
  uint16_t before[K];  // keys of the same digit ahead of this one inside the wave's part
  uint16_t digit[K];
#pragma unroll
  for (int j = 0; j < K; ++j)
  {
    const uint32_t i = wave_first + uint32_t(j) * 64u + lane;
    const bool valid = i < n;
    const uint32_t d = valid ? uint32_t(src[i] >> shift) & (bins - 1u) : 0u;
    // lanes holding the same digit (valid lanes only): one ballot per digit bit
    unsigned long long same = __ballot(valid);
    for (uint32_t b = 0; b < bits; ++b)
    {
      const bool set = ((d >> b) & 1u) != 0u;
      const unsigned long long has = __ballot(set);
      same &= set ? has : ~has;
    }
    const uint32_t ahead = uint32_t(__popcll(same & ((1ull << lane) - 1ull)));
    const uint32_t seen = valid ? uint32_t(mine[d]) : 0u;  // (every lane of the group reads before its first lane writes)
    __builtin_amdgcn_wave_barrier();
    if (valid && ahead == 0u)
    {
      mine[d] = uint16_t(seen + uint32_t(__popcll(same)));
    }
    __builtin_amdgcn_wave_barrier();
    before[j] = uint16_t(seen + ahead);
    digit[j] = uint16_t(d);
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves (in place) and the tile's total
  for (uint32_t bin = threadIdx.x; bin < bins; bin += kEvSortThreads)
  {
    uint32_t running = 0;
#pragma unroll
    for (int w = 0; w < kEvSortWaves; ++w)
    {
      const uint32_t c = evsort_counters[uint32_t(w) * bins + bin];
      evsort_counters[uint32_t(w) * bins + bin] = uint16_t(running);
      running += c;
    }
    hist[size_t(bin) * n_tiles + blockIdx.x] = running;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < K; ++j)
  {
    const uint32_t i = wave_first + uint32_t(j) * 64u + lane;
    if (i < n)
    {
      ranks[i] = uint16_t(uint32_t(evsort_counters[wave * bins + digit[j]]) + uint32_t(before[j]));
    }
  }
}

/// Per digit: exclusive prefix of its tile counts (hist[digit][tile], n_tiles <= 512) written tile-major --
/// prefix[tile][digit] -- so a scatter workgroup reads its row contiguously; totals[digit] = the digit's key count.
__global__ void __launch_bounds__(256)
  k_evsort_scan(const uint32_t *__restrict__ hist, uint32_t bins, uint32_t n_tiles, uint32_t *__restrict__ prefix,
                uint32_t *__restrict__ totals)
{
  const uint32_t bin = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (bin >= bins)
  {
    return;
  }
  const unsigned lane = __lane_id();
  constexpr uint32_t kPerLane = kEvSortMaxTiles / 64u;
  uint32_t v[kPerLane];
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k)
  {
    const uint32_t t = lane * kPerLane + k;
    v[k] = (t < n_tiles) ? hist[size_t(bin) * n_tiles + t] : 0u;
    sum += v[k];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t up = __shfl_up(incl, d);
    incl += (int(lane) >= d) ? up : 0u;
  }
  uint32_t running = incl - sum;
#pragma unroll
  for (uint32_t k = 0; k < kPerLane; ++k)
  {
    const uint32_t t = lane * kPerLane + k;
    if (t < n_tiles)
    {
      prefix[size_t(t) * bins + bin] = running;
    }
    running += v[k];
  }
  if (lane == 63u)
  {
    totals[bin] = incl;
  }
}

/// key -> dst[first position of its digit for this tile + rank inside the tile].
template <int K>
__global__ void __launch_bounds__(kEvSortThreads)
  k_evsort_scatter(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst, uint32_t n,
                   uint32_t shift, uint32_t bits, const uint32_t *__restrict__ prefix, const uint32_t *__restrict__ totals,
                   const uint16_t *__restrict__ ranks)
{
  __shared__ uint32_t s_base[1u << kEvSortMaxBits];
  __shared__ uint32_t s_wave[kEvSortWaves];
  const uint32_t bins = 1u << bits;
  const unsigned lane = __lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  // exclusive scan of the digit totals: every thread owns `per` consecutive digits
  const uint32_t per = (bins + kEvSortThreads - 1u) / kEvSortThreads;
  const uint32_t first_bin = threadIdx.x * per;
  uint32_t own = 0;
  for (uint32_t k = 0; k < per; ++k)
  {
    own += (first_bin + k < bins) ? totals[first_bin + k] : 0u;
  }
  uint32_t incl = own;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t up = __shfl_up(incl, d);
    incl += (int(lane) >= d) ? up : 0u;
  }
  if (lane == 63u)
  {
    s_wave[wave] = incl;
  }
  __syncthreads();
  uint32_t running = incl - own;
  for (uint32_t w = 0; w < wave; ++w)
  {
    running += s_wave[w];
  }
  const uint32_t *row = prefix + size_t(blockIdx.x) * bins;
  for (uint32_t k = 0; k < per; ++k)
  {
    const uint32_t bin = first_bin + k;
    if (bin < bins)
    {
      s_base[bin] = running + row[bin];
      running += totals[bin];
    }
  }
  __syncthreads();
  const uint32_t wave_first = blockIdx.x * uint32_t(kEvSortThreads * K) + wave * uint32_t(64 * K);
#pragma unroll
  for (int j = 0; j < K; ++j)
  {
    const uint32_t i = wave_first + uint32_t(j) * 64u + lane;
    if (i < n)
    {
      const unsigned long long key = src[i];
      const uint32_t d = uint32_t(key >> shift) & (bins - 1u);
      dst[s_base[d] + uint32_t(ranks[i])] = key;
    }
  }
}

struct EvSortPass
{
  uint32_t shift, bits;
};

/// The digit passes for keys whose low field (bits [0, low_field)) uses only its lowest `low_used` bits and whose
/// significant bits end at `end_bit`: the low part and the part from `low_field` up are each cut into the fewest digits of
/// at most kEvSortMaxBits, evenly.  Returns the number of passes (<= 8).
inline int evSortPlan(uint32_t low_used, uint32_t low_field, uint32_t end_bit, EvSortPass passes[8])
{
  int n = 0;
  auto cut = [&](uint32_t first, uint32_t count) {
    if (count == 0)
    {
      return;
    }
    const uint32_t parts = (count + kEvSortMaxBits - 1u) / kEvSortMaxBits;
    uint32_t at = first;
    for (uint32_t p = 0; p < parts; ++p)
    {
      const uint32_t bits = (count - (at - first) + (parts - p) - 1u) / (parts - p);
      passes[n++] = EvSortPass{ at, bits };
      at += bits;
    }
  };
  cut(0, low_used < low_field ? low_used : low_field);
  cut(low_field, end_bit > low_field ? end_bit - low_field : 0u);
  return n;
}

/// Keys per thread (a template parameter of the tile kernels) for a list of n keys: the smallest instantiated value
/// that cuts the list into at most kEvSortMaxTiles tiles; 0 when the list is too long for this sort.
inline int evSortKeysPerThread(size_t n)
{
  const int choices[] = { 2, 3, 6, 8, 12, 16, 24 };
  for (int k : choices)
  {
    if ((n + size_t(kEvSortThreads) * size_t(k) - 1) / (size_t(kEvSortThreads) * size_t(k)) <= kEvSortMaxTiles)
    {
      return k;
    }
  }
  return 0;
}

/// Bytes of scratch the sort needs for n keys: ranks (u16 per key) + hist + prefix (u32 [4096][512] each) + totals.
inline size_t evSortScratchBytes(size_t n)
{
  const size_t table = sizeof(uint32_t) * (size_t(1) << kEvSortMaxBits) * kEvSortMaxTiles;
  return ((sizeof(uint16_t) * n + 255) & ~size_t(255)) + 2 * table + sizeof(uint32_t) * (size_t(1) << kEvSortMaxBits);
}

template <int K>
inline void evSortLaunchPass(const unsigned long long *src, unsigned long long *dst, uint32_t n, EvSortPass pass,
                             uint32_t n_tiles, uint32_t *hist, uint32_t *prefix, uint32_t *totals, uint16_t *ranks,
                             hipStream_t stream)
{
  const uint32_t bins = 1u << pass.bits;
  hipLaunchKernelGGL((k_evsort_rank<K>), dim3(n_tiles), dim3(kEvSortThreads), sizeof(uint16_t) * kEvSortWaves * bins, stream,
                     src, n, pass.shift, pass.bits, n_tiles, hist, ranks);
  hipLaunchKernelGGL(k_evsort_scan, dim3((bins + 3u) / 4u), dim3(256), 0, stream, hist, bins, n_tiles, prefix, totals);
  hipLaunchKernelGGL((k_evsort_scatter<K>), dim3(n_tiles), dim3(kEvSortThreads), 0, stream, src, dst, n, pass.shift,
                     pass.bits, prefix, totals, ranks);
}

/// Sort n keys (n < 2^32) from `a`, ping-ponging with `b`; returns the buffer that holds the result (a or b), or nullptr
/// when the list is too long for this sort (nothing was launched).  `scratch`: evSortScratchBytes(n) bytes.
inline unsigned long long *evSortKeys(unsigned long long *a, unsigned long long *b, size_t n, uint32_t low_used,
                                      uint32_t low_field, uint32_t end_bit, void *scratch, hipStream_t stream)
{
  const int k = evSortKeysPerThread(n);
  if (k == 0 || n == 0 || n > 0xfffffff0ull)
  {
    return n == 0 ? a : nullptr;
  }
  EvSortPass passes[8];
  const int n_passes = evSortPlan(low_used, low_field, end_bit, passes);
  const uint32_t n_tiles = uint32_t((n + size_t(kEvSortThreads) * size_t(k) - 1) / (size_t(kEvSortThreads) * size_t(k)));
  char *at = static_cast<char *>(scratch);
  uint16_t *ranks = reinterpret_cast<uint16_t *>(at);
  at += (sizeof(uint16_t) * n + 255) & ~size_t(255);
  const size_t table = sizeof(uint32_t) * (size_t(1) << kEvSortMaxBits) * kEvSortMaxTiles;
  uint32_t *hist = reinterpret_cast<uint32_t *>(at);
  uint32_t *prefix = reinterpret_cast<uint32_t *>(at + table);
  uint32_t *totals = reinterpret_cast<uint32_t *>(at + 2 * table);
  unsigned long long *src = a, *dst = b;
  for (int p = 0; p < n_passes; ++p)
  {
    switch (k)
    {
    case 2: evSortLaunchPass<2>(src, dst, uint32_t(n), passes[p], n_tiles, hist, prefix, totals, ranks, stream); break;
    case 3: evSortLaunchPass<3>(src, dst, uint32_t(n), passes[p], n_tiles, hist, prefix, totals, ranks, stream); break;
    case 6: evSortLaunchPass<6>(src, dst, uint32_t(n), passes[p], n_tiles, hist, prefix, totals, ranks, stream); break;
    case 8: evSortLaunchPass<8>(src, dst, uint32_t(n), passes[p], n_tiles, hist, prefix, totals, ranks, stream); break;
    case 12: evSortLaunchPass<12>(src, dst, uint32_t(n), passes[p], n_tiles, hist, prefix, totals, ranks, stream); break;
    case 16: evSortLaunchPass<16>(src, dst, uint32_t(n), passes[p], n_tiles, hist, prefix, totals, ranks, stream); break;
    default: evSortLaunchPass<24>(src, dst, uint32_t(n), passes[p], n_tiles, hist, prefix, totals, ranks, stream); break;
    }
    unsigned long long *t = src;
    src = dst;
    dst = t;
  }
  return src;
}
}  // namespace ohmhip

#endif  // OHMHIP_EVENT_SORT_H
