// Per-XCD speed probe: persistent workgroups take work items from a device-wide cursor; each item is a fixed amount of
// (a) fp64 ALU work, (b) dependent global loads (pointer chase), (c) device-scope atomics.  Prints items per XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(1024) probe(int mode, unsigned *cursor, unsigned n_items, unsigned *per_xcc,
                                              const unsigned *chase, unsigned chase_mask, unsigned *sink,
                                              unsigned long long *ticks)
{
  extern __shared__ unsigned lds[];
  const unsigned xcc = __builtin_amdgcn_s_getreg(63508) & 0xf;
  if (threadIdx.x == 0)
  {
    lds[0] = atomicAdd(cursor, 1u);
  }
  __syncthreads();
  unsigned item = __builtin_amdgcn_readfirstlane(lds[0]);
  while (item < n_items)
  {
    const unsigned long long t0 = wall_clock64();
    if (mode == 0)
    {
      double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
      for (int i = 0; i < 2000; ++i)
      {
        x = x * 1.0000001 + y;
        y = y * 0.9999999 + 1e-9;
      }
      if (x == 123.0)
      {
        sink[0] = 1;
      }
    }
    else if (mode == 1)
    {
      unsigned p = (item * 7919u + threadIdx.x * 104729u) & chase_mask;
      for (int i = 0; i < 64; ++i)
      {
        p = chase[p];
      }
      if (p == 0xffffffffu)
      {
        sink[0] = 1;
      }
    }
    else
    {
      for (int i = 0; i < 64; ++i)
      {
        atomicAdd(&sink[((item * 1024u + threadIdx.x) * 64u + i) & chase_mask], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
      atomicAdd(&per_xcc[xcc], 1u);
      atomicAdd(&ticks[xcc], wall_clock64() - t0);
      lds[0] = atomicAdd(cursor, 1u);
    }
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(lds[0]);
  }
}

int main(int argc, char **argv)
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t lds_bytes = argc > 1 ? size_t(atoi(argv[1])) : 150 * 1024;
  const unsigned n = 1u << 26;  // 256 MiB of u32
  std::vector<unsigned> h(n);
  unsigned long long s = 88172645463325252ull;
  for (unsigned i = 0; i < n; ++i)
  {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = unsigned(s) & (n - 1);
  }
  printf("host init done\n");
  unsigned *d_chase, *d_sink, *d_cursor, *d_per;
  unsigned long long *d_ticks;
  hipMalloc(&d_chase, n * 4);
  hipMalloc(&d_sink, n * 4);
  hipMalloc(&d_cursor, 4);
  hipMalloc(&d_per, 64);
  hipMalloc(&d_ticks, 128);
  hipMemcpy(d_chase, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(d_sink, 0, n * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
  const char *names[3] = { "fp64 alu", "pointer chase", "device atomics" };
  for (int rep = 0; rep < 2; ++rep)
  {
    for (int mode = 0; mode < 3; ++mode)
    {
      hipMemset(d_cursor, 0, 4);
      hipMemset(d_per, 0, 64);
      hipMemset(d_ticks, 0, 128);
      printf("launch mode %d\n", mode);
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(probe, dim3(256), dim3(1024), lds_bytes, 0, mode, d_cursor, 4096u, d_per, d_chase, n - 1, d_sink,
                         d_ticks);
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      unsigned per[16];
      unsigned long long tk[16];
      hipMemcpy(per, d_per, 64, hipMemcpyDeviceToHost);
      hipMemcpy(tk, d_ticks, 128, hipMemcpyDeviceToHost);
      printf("%-14s %8.3f ms items/xcc:", names[mode], ms);
      for (int x = 0; x < 8; ++x)
      {
        printf(" %4u", per[x]);
      }
      printf("  ticks/item:");
      for (int x = 0; x < 8; ++x)
      {
        printf(" %6llu", per[x] ? tk[x] / per[x] : 0ull);
      }
      printf("\n");
    }
  }
  return 0;
}
