// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of k_region_walk
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other patterns are
// "uncalibrated").  Each kernel moves a KNOWN number of bytes from a buffer far larger than the 256 MiB Infinity Cache:
//   stream16   : 16 B per lane, coalesced                       (the guide's case)
//   gather32   : one 32-byte record per lane at a random 32-byte-aligned offset (the walk's segment records)
//   gather8    : one 8-byte word per lane at a random offset     (the direct-apply epilogue's float2 loads, sparse)
//   store8     : one 8-byte store per lane, coalesced            (the epilogue's stores)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); compare with the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint32_t hash32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ void stream16(const uint4 *src, size_t n, uint4 *sink)
{
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
  {
    const uint4 v = src[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u) sink[0] = acc;
}

__global__ void gather32(const uint4 *src, size_t records, size_t n, uint4 *sink)
{
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
  {
    const size_t r = (size_t(hash32(uint32_t(i))) * 2654435761ull + hash32(uint32_t(i >> 7))) % records;
    const uint4 a = src[2 * r], b = src[2 * r + 1];
    acc.x ^= a.x ^ b.x; acc.y ^= a.y ^ b.y;
  }
  if (acc.x == 0x12345678u) sink[0] = acc;
}

__global__ void gather8(const uint2 *src, size_t words, size_t n, uint4 *sink)
{
  uint2 acc = make_uint2(0, 0);
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
  {
    const size_t r = (size_t(hash32(uint32_t(i))) * 2654435761ull + hash32(uint32_t(i >> 5))) % words;
    const uint2 a = src[r];
    acc.x ^= a.x; acc.y ^= a.y;
  }
  if (acc.x == 0x12345678u) sink[0] = make_uint4(acc.x, acc.y, 0, 0);
}

__global__ void store8(uint2 *dst, size_t n)
{
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
  {
    dst[i] = make_uint2(uint32_t(i), 7u);
  }
}

int main()
{
  const size_t bytes = size_t(2) << 30;  // 2 GiB buffer
  void *buf = nullptr;
  uint4 *sink = nullptr;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  const dim3 grid(4096), block(256);
  const size_t n16 = bytes / 16;
  hipLaunchKernelGGL(stream16, grid, block, 0, 0, static_cast<const uint4 *>(buf), n16, sink);
  hipDeviceSynchronize();
  printf("stream16  bytes read    %zu\n", n16 * 16);
  const size_t n_gather = size_t(32) << 20;  // 32 M records
  hipLaunchKernelGGL(gather32, grid, block, 0, 0, static_cast<const uint4 *>(buf), bytes / 32, n_gather, sink);
  hipDeviceSynchronize();
  printf("gather32  bytes read    %zu\n", n_gather * 32);
  hipLaunchKernelGGL(gather8, grid, block, 0, 0, static_cast<const uint2 *>(buf), bytes / 8, n_gather, sink);
  hipDeviceSynchronize();
  printf("gather8   bytes read    %zu\n", n_gather * 8);
  hipLaunchKernelGGL(store8, grid, block, 0, 0, static_cast<uint2 *>(buf), bytes / 8);
  hipDeviceSynchronize();
  printf("store8    bytes written %zu\n", bytes);
  return 0;
}
