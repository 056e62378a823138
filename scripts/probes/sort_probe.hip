// sort_probe.hip -- rocPRIM one-sweep radix sort of 64-bit keys with 8 / 9 / 10 radix bits (11 does not fit: its histogram kernel wants 192 KiB of LDS) per pass: does a wider
// digit (fewer passes over the NDT / TSDF event keys) pay on gfx950?  Build: hipcc -O3 --offload-arch=gfx950.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                   \
  do                                                                               \
  {                                                                                \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess)                                                          \
    {                                                                              \
      std::printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

template <unsigned kBits, unsigned kBlock, unsigned kItems, unsigned kHistBlock = 256, unsigned kHistItems = 12,
          rocprim::block_radix_rank_algorithm kRank = rocprim::block_radix_rank_algorithm::default_algorithm>
int run(const char *name, const std::vector<unsigned long long> &host, unsigned end_bit, const std::vector<unsigned long long> &expect)
{
  using Onesweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<kHistBlock, kHistItems>, rocprim::kernel_config<kBlock, kItems>, kBits, kRank>;
  using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, Onesweep, size_t(1) << 15>;
  const size_t n = host.size();
  unsigned long long *a = nullptr, *b = nullptr;
  CHECK(hipMalloc(&a, n * 8));
  CHECK(hipMalloc(&b, n * 8));
  size_t temp_bytes = 0;
  CHECK(rocprim::radix_sort_keys<Config>(nullptr, temp_bytes, a, b, n, 0, end_bit, nullptr));
  void *temp = nullptr;
  CHECK(hipMalloc(&temp, temp_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep)
  {
    CHECK(hipMemcpy(a, host.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0, nullptr));
    CHECK(rocprim::radix_sort_keys<Config>(temp, temp_bytes, a, b, n, 0, end_bit, nullptr));
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  std::vector<unsigned long long> out(n);
  CHECK(hipMemcpy(out.data(), b, n * 8, hipMemcpyDeviceToHost));
  const bool ok = out == expect;
  std::printf("%-34s n %9zu bits %2u: %8.3f ms  %s\n", name, n, end_bit, best, ok ? "sorted" : "WRONG");
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  CHECK(hipFree(temp));
  return ok ? 0 : 1;
}

int runDefault(const std::vector<unsigned long long> &host, unsigned end_bit, const std::vector<unsigned long long> &expect)
{
  // what the library uses today: rocPRIM's tuned one-sweep configuration for the architecture
  using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, size_t(1) << 15>;
  const size_t n = host.size();
  unsigned long long *a = nullptr, *b = nullptr;
  CHECK(hipMalloc(&a, n * 8));
  CHECK(hipMalloc(&b, n * 8));
  size_t temp_bytes = 0;
  CHECK(rocprim::radix_sort_keys<Config>(nullptr, temp_bytes, a, b, n, 0, end_bit, nullptr));
  void *temp = nullptr;
  CHECK(hipMalloc(&temp, temp_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep)
  {
    CHECK(hipMemcpy(a, host.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0, nullptr));
    CHECK(rocprim::radix_sort_keys<Config>(temp, temp_bytes, a, b, n, 0, end_bit, nullptr));
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  std::vector<unsigned long long> out(n);
  CHECK(hipMemcpy(out.data(), b, n * 8, hipMemcpyDeviceToHost));
  std::printf("%-34s n %9zu bits %2u: %8.3f ms  %s\n", "rocPRIM default (tuned)", n, end_bit, best, out == expect ? "sorted" : "WRONG");
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  CHECK(hipFree(temp));
  return 0;
}

int main()
{
  struct Case
  {
    size_t n;
    unsigned bits;
  };
  const Case cases[] = { { 1300000, 53 }, { 24000000, 58 } };
  for (const Case &c : cases)
  {
    std::vector<unsigned long long> host(c.n);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < c.n; ++i)
    {
      x ^= x << 13;
      x ^= x >> 7;
      x ^= x << 17;
      host[i] = x & ((1ull << c.bits) - 1ull);
    }
    std::vector<unsigned long long> expect(host);
    std::sort(expect.begin(), expect.end());
    int bad = 0;
    constexpr auto kMatch = rocprim::block_radix_rank_algorithm::match;
    bad += runDefault(host, c.bits, expect);
    bad += run<8, 256, 12, 256, 12, kMatch>("radix 8  (256 x 12, match)", host, c.bits, expect);
    bad += run<8, 512, 12, 256, 12, kMatch>("radix 8  (512 x 12, match)", host, c.bits, expect);
    bad += run<8, 1024, 6, 256, 12, kMatch>("radix 8  (1024 x 6, match)", host, c.bits, expect);
    bad += run<9, 512, 12, 256, 12, kMatch>("radix 9  (512 x 12, match)", host, c.bits, expect);
    bad += run<9, 512, 8, 256, 12, kMatch>("radix 9  (512 x 8, match)", host, c.bits, expect);
    bad += run<9, 1024, 6, 256, 12, kMatch>("radix 9  (1024 x 6, match)", host, c.bits, expect);
    bad += run<9, 1024, 7, 256, 12, kMatch>("radix 9  (1024 x 7, match)", host, c.bits, expect);
    bad += run<10, 512, 12, 256, 12, kMatch>("radix 10 (512 x 12, match)", host, c.bits, expect);
    bad += run<10, 1024, 6, 256, 12, kMatch>("radix 10 (1024 x 6, match)", host, c.bits, expect);
    bad += run<10, 1024, 7, 256, 12, kMatch>("radix 10 (1024 x 7, match)", host, c.bits, expect);
    if (bad)
    {
      std::printf("some configuration sorted wrongly\n");
    }
  }
  return 0;
}
