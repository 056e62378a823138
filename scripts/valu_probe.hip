// Issue cost of the VALU instructions the walk step uses, on a CU filled like k_region_walk (1 x 1024 threads per CU):
// each kernel runs `iters` trips of 32 independent instructions of one kind; cycles/instr = time / (iters * 32 * waves
// per SIMD) * clock.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define PROBE(name, ASM, DECL, CONSTRAINTS)                                     \
  __global__ void __launch_bounds__(1024) name(int iters, double *sink)        \
  {                                                                             \
    DECL;                                                                       \
    for (int i = 0; i < iters; ++i)                                             \
    {                                                                           \
      asm volatile(REP32(ASM "\n") : CONSTRAINTS);                              \
    }                                                                           \
    if (threadIdx.x == 1023 && d0 == 1234.5)                                    \
    {                                                                           \
      sink[0] = d0 + d1 + double(i0) + double(i1);                              \
    }                                                                           \
  }

#define DECLS double d0 = threadIdx.x, d1 = 3.0 + threadIdx.x, d2 = 0; int i0 = threadIdx.x, i1 = 7; \
  unsigned long long m = 0
#define CONS "+v"(d0), "+v"(d1), "+v"(d2), "+v"(i0), "+v"(i1), "+s"(m) : : "vcc"

PROBE(p_add_f64, "v_add_f64 %2, %0, %1", DECLS, CONS)
PROBE(p_mul_f64, "v_mul_f64 %2, %0, %1", DECLS, CONS)
PROBE(p_fma_f64, "v_fma_f64 %2, %0, %1, %2", DECLS, CONS)
PROBE(p_cvt_f64_i32, "v_cvt_f64_i32 %2, %3", DECLS, CONS)
PROBE(p_cmp_f64, "v_cmp_lt_f64 %5, %0, %1", DECLS, CONS)
PROBE(p_cmp_i32, "v_cmp_lt_i32 %5, %3, %4", DECLS, CONS)
PROBE(p_cndmask, "v_cndmask_b32_e64 %3, %3, %4, %5", DECLS, CONS)
PROBE(p_add_u32, "v_add_u32 %3, %3, %4", DECLS, CONS)
PROBE(p_addc, "v_addc_co_u32_e64 %3, vcc, 0, %3, %5", DECLS, CONS)
PROBE(p_mov_b32, "v_mov_b32 %3, %4", DECLS, CONS)
PROBE(p_lshl, "v_lshlrev_b32 %3, 1, %4", DECLS, CONS)

template <typename K>
void run(const char *name, K kernel, double *sink)
{
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(kernel, dim3(256), dim3(1024), 0, 0, 10, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kernel, dim3(256), dim3(1024), 0, 0, iters, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  // per SIMD: 4 waves x iters x 32 instructions
  const double instr = 4.0 * iters * 32.0;
  printf("%-16s %8.3f ms  %6.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  double *sink;
  hipMalloc(&sink, 64);
  run("v_add_f64", p_add_f64, sink);
  run("v_mul_f64", p_mul_f64, sink);
  run("v_fma_f64", p_fma_f64, sink);
  run("v_cvt_f64_i32", p_cvt_f64_i32, sink);
  run("v_cmp_lt_f64", p_cmp_f64, sink);
  run("v_cmp_lt_i32", p_cmp_i32, sink);
  run("v_cndmask_b32", p_cndmask, sink);
  run("v_add_u32", p_add_u32, sink);
  run("v_addc_co_u32", p_addc, sink);
  run("v_mov_b32", p_mov_b32, sink);
  run("v_lshlrev_b32", p_lshl, sink);
  return 0;
}
