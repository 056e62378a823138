// Development probe: host-to-device rates for a 48 MB pinned block -- one hipMemcpyAsync, 1.5 MiB pieces, and a kernel
// that reads the mapped pinned memory itself.  Build: hipcc --offload-arch=gfx950 -O3 -o h2d_probe h2d_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
  {
    dst[i] = src[i];
  }
}

int main()
{
  const size_t bytes = size_t(48) << 20;
  char *h = nullptr, *d = nullptr;
  hipHostMalloc(reinterpret_cast<void **>(&h), bytes, hipHostMallocDefault);
  hipMalloc(reinterpret_cast<void **>(&d), bytes);
  std::memset(h, 1, bytes);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  void *hd = nullptr;
  hipHostGetDevicePointer(&hd, h, 0);
  auto time_it = [&](const char *name, auto fn) {
    fn();
    hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 10; ++i)
    {
      fn();
    }
    hipStreamSynchronize(s);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 10;
    std::printf("%-40s %.3f ms  %.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  time_it("hipMemcpyAsync, one 48 MB copy", [&] { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); });
  time_it("hipMemcpyAsync, 1.5 MiB pieces", [&] {
    const size_t piece = size_t(3) << 19;
    for (size_t at = 0; at < bytes; at += piece)
    {
      hipMemcpyAsync(d + at, h + at, piece, hipMemcpyHostToDevice, s);
    }
  });
  time_it("hipMemcpyAsync, 12 MiB pieces", [&] {
    const size_t piece = size_t(12) << 20;
    for (size_t at = 0; at < bytes; at += piece)
    {
      hipMemcpyAsync(d + at, h + at, piece, hipMemcpyHostToDevice, s);
    }
  });
  time_it("kernel reading pinned memory, 256 blocks", [&] {
    hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s, static_cast<const uint4 *>(hd), reinterpret_cast<uint4 *>(d), bytes / 16);
  });
  time_it("kernel reading pinned memory, 64 blocks", [&] {
    hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s, static_cast<const uint4 *>(hd), reinterpret_cast<uint4 *>(d), bytes / 16);
  });
  time_it("kernel, 1.5 MiB pieces, 16 blocks each", [&] {
    const size_t piece = size_t(3) << 19;
    for (size_t at = 0; at < bytes; at += piece)
    {
      hipLaunchKernelGGL(k_copy, dim3(16), dim3(256), 0, s, reinterpret_cast<const uint4 *>(static_cast<char *>(hd) + at), reinterpret_cast<uint4 *>(d + at), piece / 16);
    }
  });
  return 0;
}
