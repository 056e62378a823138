// LDS cost of the walk kernel's per-visit operation on a CU filled like k_region_walk (1 x 1024 threads, 64 KiB tile):
// cycles per wave-instruction per CU for returning / non-returning adds and reads, random vs conflict-free addresses,
// dependent (one in flight per wave) vs 4 in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kWords = 16384;

__device__ inline uint32_t hash32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// mode: 0 add_rtn, 1 add (no return), 2 read
// pattern: 0 random word, 1 lane-linear (conflict free), 2 random but bank == lane % 32 (conflict free, scattered rows)
template <int kMode, int kPattern, int kInFlight>
__global__ void __launch_bounds__(1024) probe(int iters, uint32_t *sink)
{
  extern __shared__ uint32_t tile[];
  for (int i = threadIdx.x; i < kWords; i += 1024) tile[i] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it)
  {
    uint32_t addr[kInFlight], old[kInFlight];
#pragma unroll
    for (int k = 0; k < kInFlight; ++k)
    {
      seed = seed * 1664525u + 1013904223u;
      const uint32_t h = hash32(seed);
      uint32_t w;
      if (kPattern == 0) w = h & (kWords - 1);
      else if (kPattern == 1) w = (lane + 64u * uint32_t(it * kInFlight + k)) & (kWords - 1);
      else w = ((h & (kWords / 32 - 1)) * 32u) | (lane & 31u);
      addr[k] = w * 4u;
    }
#pragma unroll
    for (int k = 0; k < kInFlight; ++k)
    {
      if (kMode == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(old[k]) : "v"(addr[k]), "v"(1u) : "memory");
      else if (kMode == 1) { asm volatile("ds_add_u32 %0, %1" : : "v"(addr[k]), "v"(1u) : "memory"); old[k] = 0; }
      else asm volatile("ds_read_b32 %0, %1" : "=v"(old[k]) : "v"(addr[k]) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) acc += old[k];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <typename K>
void run(const char *name, K kernel, int in_flight, uint32_t *sink, int threads)
{
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kWords * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), kWords * 4, 0, 10, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), kWords * 4, 0, iters, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double instr = double(threads / 64) * iters * in_flight;  // wave-instructions per CU
  printf("%-44s waves/CU %2d  %8.3f ms  %7.2f cycles per wave-instruction per CU (2.4 GHz)\n", name, threads / 64, ms,
         ms * 1e-3 * 2.4e9 / instr);
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint32_t *sink;
  hipMalloc(&sink, 64);
  for (int threads : { 1024, 256 })
  {
    run("add_rtn random        1 in flight", probe<0, 0, 1>, 1, sink, threads);
    run("add_rtn random        4 in flight", probe<0, 0, 4>, 4, sink, threads);
    run("add_rtn lane-linear   1 in flight", probe<0, 1, 1>, 1, sink, threads);
    run("add_rtn lane-linear   4 in flight", probe<0, 1, 4>, 4, sink, threads);
    run("add_rtn own-bank rand 1 in flight", probe<0, 2, 1>, 1, sink, threads);
    run("add_rtn own-bank rand 4 in flight", probe<0, 2, 4>, 4, sink, threads);
    run("add     random        4 in flight", probe<1, 0, 4>, 4, sink, threads);
    run("add     own-bank rand 4 in flight", probe<1, 2, 4>, 4, sink, threads);
    run("read    random        4 in flight", probe<2, 0, 4>, 4, sink, threads);
    run("read    own-bank rand 4 in flight", probe<2, 2, 4>, 4, sink, threads);
  }
  return 0;
}
