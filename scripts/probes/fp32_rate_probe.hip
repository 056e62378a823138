// Issue rate of the instructions a "rank" formulation of the walk step would be built from: the
// voxel of crossing k from T = f_a + k d_a and two reciprocal multiplies -- fp32 fma / floor / convert, or 32-bit
// fixed-point multiplies -- next to the full-rate integer instruction the production step is made of (v_xor_b32).
// Same method as scripts/valu_probe.hip: independent streams of ONE instruction kind over 8 rotating destinations,
// one workgroup of 256 * w threads per CU (w waves per SIMD); cycles per instruction per SIMD at 2.4 GHz nominal.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/probes/fp32_rate_probe scripts/probes/fp32_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x

#define THR_KERNEL(name, OP)                                                      \
  __global__ void __launch_bounds__(1024) name(int iters, double *sink)          \
  {                                                                               \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
             a7 = a0 + 7, b = 0x3f800001u, c = threadIdx.x * 7u;                  \
    for (int i = 0; i < iters; ++i)                                               \
    {                                                                             \
      asm volatile(REP4(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7))          \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "v"(c));                                             \
    }                                                                             \
    if (threadIdx.x == 1023 && double(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) == 1234.5) \
    {                                                                             \
      sink[0] = double(a0);                                                       \
    }                                                                             \
  }
// operands: %0..%7 rotating VGPRs, %8 %9 VGPR inputs
#define T_XOR(n) "v_xor_b32 %" #n ", %8, %" #n "\n"
#define T_FMA_F32(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define T_FMAC_F32(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define T_MUL_F32(n) "v_mul_f32 %" #n ", %8, %" #n "\n"
#define T_ADD_F32(n) "v_add_f32 %" #n ", %8, %" #n "\n"
#define T_FLOOR_F32(n) "v_floor_f32 %" #n ", %8\n"
#define T_FRACT_F32(n) "v_fract_f32 %" #n ", %8\n"
#define T_CVT_I32_F32(n) "v_cvt_i32_f32 %" #n ", %8\n"
#define T_CVT_F32_U32(n) "v_cvt_f32_u32 %" #n ", %9\n"
#define T_MED3_F32(n) "v_med3_f32 %" #n ", %8, %9, %" #n "\n"
#define T_MAX3_F32(n) "v_max3_f32 %" #n ", %8, %9, %" #n "\n"
#define T_MAX_F32(n) "v_max_f32 %" #n ", %8, %" #n "\n"
#define T_RCP_F32(n) "v_rcp_f32 %" #n ", %8\n"
#define T_MUL_HI_U32(n) "v_mul_hi_u32 %" #n ", %9, %" #n "\n"
#define T_MUL_LO_U32(n) "v_mul_lo_u32 %" #n ", %9, %" #n "\n"
#define T_MUL_U32_U24(n) "v_mul_u32_u24 %" #n ", %9, %" #n "\n"
#define T_MAD_U32_U24(n) "v_mad_u32_u24 %" #n ", %9, %8, %" #n "\n"
#define T_MUL_HI_U24(n) "v_mul_hi_u32_u24 %" #n ", %9, %" #n "\n"
THR_KERNEL(thr_xor, T_XOR)
THR_KERNEL(thr_fma_f32, T_FMA_F32)
THR_KERNEL(thr_fmac_f32, T_FMAC_F32)
THR_KERNEL(thr_mul_f32, T_MUL_F32)
THR_KERNEL(thr_add_f32, T_ADD_F32)
THR_KERNEL(thr_floor_f32, T_FLOOR_F32)
THR_KERNEL(thr_fract_f32, T_FRACT_F32)
THR_KERNEL(thr_cvt_i32_f32, T_CVT_I32_F32)
THR_KERNEL(thr_cvt_f32_u32, T_CVT_F32_U32)
THR_KERNEL(thr_med3_f32, T_MED3_F32)
THR_KERNEL(thr_max3_f32, T_MAX3_F32)
THR_KERNEL(thr_max_f32, T_MAX_F32)
THR_KERNEL(thr_rcp_f32, T_RCP_F32)
THR_KERNEL(thr_mul_hi_u32, T_MUL_HI_U32)
THR_KERNEL(thr_mul_lo_u32, T_MUL_LO_U32)
THR_KERNEL(thr_mul_u32_u24, T_MUL_U32_U24)
THR_KERNEL(thr_mad_u32_u24, T_MAD_U32_U24)
THR_KERNEL(thr_mul_hi_u24, T_MUL_HI_U24)

// packed fp32: two fma per lane per instruction (64-bit register pairs)
__global__ void __launch_bounds__(1024) thr_pk_fma_f32(int iters, double *sink)
{
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double b = 1.0000001, c = 0.5;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP4("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n"
                      "v_pk_fma_f32 %3, %8, %9, %3\n v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n"
                      "v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b), "v"(c));
  }
  if (threadIdx.x == 1023 && a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1234.5)
  {
    sink[0] = a0;
  }
}

template <typename K>
void run(const char *name, K kernel, double *sink)
{
  const int iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  printf("%-18s", name);
  for (int waves = 1; waves <= 4; waves *= 2)
  {
    const int threads = 256 * waves;
    hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 0, 0, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 0, 0, iters, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double per_wave = ms * 1e-3 * 2.4e9 / (double(iters) * 32.0);
    printf("  w%d %6.2f/%5.2f", waves, per_wave, per_wave / waves);
  }
  printf("\n");
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  double *sink;
  hipMalloc(&sink, 64);
  printf("cycles per instruction at 2.4 GHz: per wave / per SIMD, at w = 1, 2, 4 waves per SIMD\n");
  run("v_xor_b32", thr_xor, sink);
  run("v_fma_f32", thr_fma_f32, sink);
  run("v_fmac_f32", thr_fmac_f32, sink);
  run("v_pk_fma_f32", thr_pk_fma_f32, sink);
  run("v_mul_f32", thr_mul_f32, sink);
  run("v_add_f32", thr_add_f32, sink);
  run("v_max_f32", thr_max_f32, sink);
  run("v_floor_f32", thr_floor_f32, sink);
  run("v_fract_f32", thr_fract_f32, sink);
  run("v_cvt_i32_f32", thr_cvt_i32_f32, sink);
  run("v_cvt_f32_u32", thr_cvt_f32_u32, sink);
  run("v_med3_f32", thr_med3_f32, sink);
  run("v_max3_f32", thr_max3_f32, sink);
  run("v_rcp_f32", thr_rcp_f32, sink);
  run("v_mul_hi_u32", thr_mul_hi_u32, sink);
  run("v_mul_lo_u32", thr_mul_lo_u32, sink);
  run("v_mul_u32_u24", thr_mul_u32_u24, sink);
  run("v_mul_hi_u32_u24", thr_mul_hi_u24, sink);
  run("v_mad_u32_u24", thr_mad_u32_u24, sink);
  return 0;
}
