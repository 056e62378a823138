// evsort_probe.hip -- scripts/probes/event_sort.h (a prototype, not part of the library) against std::sort and against rocPRIM's tuned one-sweep sort, on keys
// shaped like a batch's event list: [slot | voxel:15 | ray:28 | sample bit], a few per cent invalid (all ones).
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/probes/evsort_probe scripts/probes/evsort_probe.hip
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "event_sort.h"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                            \
  do                                                                        \
  {                                                                         \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess)                                                   \
    {                                                                       \
      std::printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

static unsigned bitsFor(unsigned long long v)
{
  unsigned b = 0;
  while (v)
  {
    ++b;
    v >>= 1;
  }
  return b;
}

int runCase(size_t n, unsigned n_rays, unsigned n_slots, double invalid_share)
{
  std::vector<unsigned long long> host(n);
  unsigned long long x = 88172645463325252ull + n;
  auto next = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
  };
  for (size_t i = 0; i < n; ++i)
  {
    const unsigned long long r = next();
    if (double(r & 0xffff) / 65536.0 < invalid_share)
    {
      host[i] = ~0ull;
      continue;
    }
    const unsigned long long slot = (r >> 16) % n_slots, voxel = (r >> 40) & 0x7fff, ray = next() % n_rays, s = next() & 1u;
    host[i] = (slot << 44) | (voxel << 29) | (ray << 1) | s;
  }
  std::vector<unsigned long long> expect(host);
  std::sort(expect.begin(), expect.end());
  unsigned slot_bits = 1;
  while ((1u << slot_bits) <= n_slots)
  {
    ++slot_bits;
  }
  const unsigned end_bit = std::min(64u, 44u + slot_bits + 1u);
  const unsigned low_used = bitsFor(n_rays - 1) + 1;

  unsigned long long *a = nullptr, *b = nullptr;
  void *scratch = nullptr;
  CHECK(hipMalloc(&a, n * 8 + 8));
  CHECK(hipMalloc(&b, n * 8 + 8));
  CHECK(hipMalloc(&scratch, ohmhip::evSortScratchBytes(n)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  unsigned long long *result = nullptr;
  for (int rep = 0; rep < 5; ++rep)
  {
    CHECK(hipMemcpy(a, host.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0, nullptr));
    result = ohmhip::evSortKeys(a, b, n, low_used, 29, end_bit, scratch, nullptr);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  bool ok = false;
  if (result)
  {
    std::vector<unsigned long long> out(n);
    CHECK(hipMemcpy(out.data(), result, n * 8, hipMemcpyDeviceToHost));
    ok = out == expect;
  }
  // rocPRIM, as the library calls it
  using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, size_t(1) << 15>;
  size_t temp_bytes = 0;
  CHECK(rocprim::radix_sort_keys<Config>(nullptr, temp_bytes, a, b, n, 0, end_bit, nullptr));
  void *temp = nullptr;
  CHECK(hipMalloc(&temp, temp_bytes));
  float best_lib = 1e9f;
  for (int rep = 0; rep < 5; ++rep)
  {
    CHECK(hipMemcpy(a, host.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0, nullptr));
    CHECK(rocprim::radix_sort_keys<Config>(temp, temp_bytes, a, b, n, 0, end_bit, nullptr));
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best_lib = std::min(best_lib, ms);
  }
  ohmhip::EvSortPass passes[8];
  const int n_passes = ohmhip::evSortPlan(low_used, 29, end_bit, passes);
  std::printf("n %9zu rays %8u slots %6u: event_sort %7.3f ms (%d passes, K %2d) %s   rocPRIM %7.3f ms\n", n, n_rays, n_slots,
              best, n_passes, ohmhip::evSortKeysPerThread(n), result ? (ok ? "sorted" : "WRONG") : "declined", best_lib);
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  CHECK(hipFree(scratch));
  CHECK(hipFree(temp));
  return (result && !ok) ? 1 : 0;
}

int main()
{
  int bad = 0;
  bad += runCase(1300000, 1000000, 121, 0.02);
  bad += runCase(300000, 250000, 40, 0.05);
  bad += runCase(3000000, 1000000, 4465, 0.01);
  bad += runCase(5000, 4096, 7, 0.1);
  bad += runCase(1, 1, 1, 0.0);
  bad += runCase(70001, 65537, 1000000, 0.0);
  bad += runCase(3200000, 4000000, 4465, 0.0);  // too long: declined
  std::printf(bad ? "FAILED\n" : "EVSORT_OK\n");
  return bad;
}
