// event_probe.hip -- what an event between two dependent kernels costs on gfx950 / ROCm 7.2, and whether the stop event
// of hipExtLaunchKernelGGL (bound to the kernel's own completion signal) removes that cost.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/event_probe scripts/probes/event_probe.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_spin(unsigned long long *out, unsigned cycles)
{
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles)
  {
  }
  if (threadIdx.x == 0 && blockIdx.x == 0)
  {
    out[0] = wall_clock64();
  }
}

#define CK(x)                                                                        \
  do                                                                                 \
  {                                                                                  \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess)                                                            \
    {                                                                                \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);     \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

int main()
{
  unsigned long long *d = nullptr;
  CK(hipMalloc(&d, 64));
  hipStream_t s, f;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&f, hipStreamNonBlocking));
  const int n = 400;
  std::vector<hipEvent_t> ev(2 * n + 2);
  for (auto &e : ev)
  {
    CK(hipEventCreate(&e));
  }
  const unsigned spin = 2000;  // 100 MHz ticks: 20 us per kernel
  auto run = [&](int mode, const char *name) -> int {
    // warm
    for (int i = 0; i < 10; ++i)
    {
      hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, spin);
    }
    CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i)
    {
      switch (mode)
      {
      case 0:  // back to back
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, spin);
        break;
      case 1:  // an event record after every kernel
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, spin);
        CK(hipEventRecord(ev[i], s));
        break;
      case 2:  // two records after every kernel
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, spin);
        CK(hipEventRecord(ev[2 * i], s));
        CK(hipEventRecord(ev[2 * i + 1], s));
        break;
      case 3:  // the kernel's own stop event
        hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, nullptr, ev[i], 0, d, spin);
        break;
      case 4:  // start and stop event on the launch
        hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, d, spin);
        break;
      case 5:  // stop event + another stream waiting on it (a kernel there per step)
        hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, nullptr, ev[i], 0, d, spin);
        CK(hipStreamWaitEvent(f, ev[i], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, f, d + 1, 100u);
        break;
      case 6:  // event record + another stream waiting on it
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, spin);
        CK(hipEventRecord(ev[i], s));
        CK(hipStreamWaitEvent(f, ev[i], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, f, d + 1, 100u);
        break;
      }
    }
    CK(hipStreamSynchronize(s));
    CK(hipStreamSynchronize(f));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%-44s %8.2f us per step (kernel spins %.0f us)\n", name, us / n, spin / 100.0);
    return 0;
  };
  if (run(0, "kernels back to back")) return 1;
  if (run(1, "hipEventRecord after each")) return 1;
  if (run(2, "two hipEventRecord after each")) return 1;
  if (run(3, "hipExtLaunchKernelGGL stop event")) return 1;
  if (run(4, "hipExtLaunchKernelGGL start + stop event")) return 1;
  if (run(5, "stop event + cross-stream wait")) return 1;
  if (run(6, "hipEventRecord + cross-stream wait")) return 1;
  // Semantics: elapsed time between the stop events of two consecutive launches = the second kernel's duration (+ gap)?
  hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, nullptr, ev[0], 0, d, spin);
  hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, nullptr, ev[1], 0, d, 5 * spin);
  hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, ev[2], ev[3], 0, d, 3 * spin);
  CK(hipStreamSynchronize(s));
  float a = 0, b = 0, c = 0;
  CK(hipEventElapsedTime(&a, ev[0], ev[1]));
  CK(hipEventElapsedTime(&b, ev[2], ev[3]));
  CK(hipEventElapsedTime(&c, ev[1], ev[3]));
  std::printf("elapsed stop(k1)->stop(k2) %.1f us (k2 spins %.0f); start(k3)->stop(k3) %.1f us (k3 spins %.0f); "
              "stop(k2)->stop(k3) %.1f us\n", a * 1e3, 5 * spin / 100.0, b * 1e3, 3 * spin / 100.0, c * 1e3);
  // query semantics: a stop event is not complete before its kernel is
  hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, nullptr, ev[4], 0, d, 50 * spin);
  const hipError_t q0 = hipEventQuery(ev[4]);
  CK(hipEventSynchronize(ev[4]));
  const hipError_t q1 = hipEventQuery(ev[4]);
  std::printf("query while running: %s; after synchronize: %s\n", hipGetErrorName(q0), hipGetErrorName(q1));
  return 0;
}
