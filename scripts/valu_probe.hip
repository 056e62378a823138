// Issue cost and dependent-chain latency of the VALU / LDS instructions the walk step uses, measured at 1..8 waves per
// SIMD (is a wave64 VALU instruction 2 or 4 cycles, and how many waves hide a dependent chain?).
//
// Every kernel runs `iters` trips of 32 instructions of one kind: either ONE dependent chain (each instruction reads the
// previous result) or EIGHT independent chains interleaved.  Launch shapes: one workgroup of 256 * w threads per CU puts
// w waves on every SIMD (w = 1..4); 2 workgroups of 1024 threads per CU gives 8.  Reported:
//   cyc/instr/wave = time * clock / (iters * 32)          -- what one wave sees between its own instructions
//   cyc/instr/SIMD = time * clock / (iters * 32 * waves)  -- the SIMD's issue interval when it is the bound
// (clock: 2.4 GHz nominal; the chip may run lower under load -- ratios between rows are what matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// dependent chains -----------------------------------------------------------------------------------------------------
#define DEP_KERNEL(name, ASM, TYPE, INIT)                                         \
  __global__ void __launch_bounds__(1024) name(int iters, double *sink)          \
  {                                                                               \
    TYPE a = INIT, b = (TYPE)3;                                                   \
    for (int i = 0; i < iters; ++i)                                               \
    {                                                                             \
      asm volatile(REP32(ASM "\n") : "+v"(a) : "v"(b));                           \
    }                                                                             \
    if (threadIdx.x == 1023 && double(a) == 1234.5)                               \
    {                                                                             \
      sink[0] = double(a);                                                        \
    }                                                                             \
  }

// eight independent chains, interleaved -----------------------------------------------------------------------------------
#define IND_KERNEL(name, OP, TYPE, INIT)                                          \
  __global__ void __launch_bounds__(1024) name(int iters, double *sink)          \
  {                                                                               \
    TYPE a0 = INIT, a1 = INIT + 1, a2 = INIT + 2, a3 = INIT + 3, a4 = INIT + 4, a5 = INIT + 5, a6 = INIT + 6, \
         a7 = INIT + 7, b = (TYPE)3;                                              \
    for (int i = 0; i < iters; ++i)                                               \
    {                                                                             \
      asm volatile(REP4(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7))          \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b));                                                     \
    }                                                                             \
    if (threadIdx.x == 1023 && double(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) == 1234.5) \
    {                                                                             \
      sink[0] = double(a0);                                                       \
    }                                                                             \
  }

#define ADD_U32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define FMA_F64(n) "v_fma_f64 %" #n ", %" #n ", %8, %8\n"
#define ADD_F64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define MIN3_U32(n) "v_min3_u32 %" #n ", %" #n ", %8, %8\n"
#define ADD_SAT(n) "v_add_u32_e64 %" #n ", %" #n ", %8 clamp\n"

DEP_KERNEL(dep_add_u32, "v_add_u32 %0, %0, %1", unsigned, threadIdx.x)
DEP_KERNEL(dep_add_sat, "v_add_u32_e64 %0, %0, %1 clamp", unsigned, threadIdx.x)
DEP_KERNEL(dep_min3_u32, "v_min3_u32 %0, %0, %1, %1", unsigned, threadIdx.x)
DEP_KERNEL(dep_fma_f64, "v_fma_f64 %0, %0, %1, %1", double, double(threadIdx.x))
DEP_KERNEL(dep_add_f64, "v_add_f64 %0, %0, %1", double, double(threadIdx.x))
IND_KERNEL(ind_add_u32, ADD_U32, unsigned, threadIdx.x)
IND_KERNEL(ind_add_sat, ADD_SAT, unsigned, threadIdx.x)
IND_KERNEL(ind_min3_u32, MIN3_U32, unsigned, threadIdx.x)
IND_KERNEL(ind_fma_f64, FMA_F64, double, double(threadIdx.x))
IND_KERNEL(ind_add_f64, ADD_F64, double, double(threadIdx.x))

// VALU -> SGPR mask -> VALU round trip (the walk step's v_cmp / v_cndmask pairs): one dependent chain
__global__ void __launch_bounds__(1024) dep_cmp_cndmask(int iters, double *sink)
{
  unsigned a = threadIdx.x, b = 3;
  unsigned long long m = 0;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP8(REP4("v_cmp_lt_u32_e64 %1, %0, %2\n v_cndmask_b32_e64 %0, %0, %2, %1\n")) : "+v"(a), "+s"(m) : "v"(b));
  }
  if (threadIdx.x == 1023 && a == 77777u)
  {
    sink[0] = double(a) + double(m);
  }
}

// returning LDS atomic, dependent on its own result through the address (the walk's add -> flag test -> next address
// chain without the arithmetic), conflict free: lane l works on word l of the wave's 64-word row
__global__ void __launch_bounds__(1024) dep_ds_add_rtn(int iters, double *sink)
{
  __shared__ unsigned tile[16 * 64 * 2];
  tile[threadIdx.x] = 0;
  tile[threadIdx.x + 1024] = 0;
  __syncthreads();
  unsigned addr = threadIdx.x * 4u, one = 1, old = 0;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP32("ds_add_rtn_u32 %1, %0, %2\n s_waitcnt lgkmcnt(0)\n v_and_or_b32 %0, %1, 0, %0\n")
                 : "+v"(addr), "+v"(old)
                 : "v"(one)
                 : "memory");
  }
  if (threadIdx.x == 1023 && old == 77777u)
  {
    sink[0] = double(old);
  }
}

// the same with two adds in flight per wave (second one to the row's other half)
__global__ void __launch_bounds__(1024) dep_ds_add_rtn_x2(int iters, double *sink)
{
  __shared__ unsigned tile[16 * 64 * 2];
  tile[threadIdx.x] = 0;
  tile[threadIdx.x + 1024] = 0;
  __syncthreads();
  unsigned addr = threadIdx.x * 4u, addr2 = threadIdx.x * 4u + 4096u, one = 1, old = 0, old2 = 0;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP8(REP4("ds_add_rtn_u32 %2, %0, %4\n ds_add_rtn_u32 %3, %1, %4\n s_waitcnt lgkmcnt(0)\n"
                           "v_and_or_b32 %0, %2, 0, %0\n v_and_or_b32 %1, %3, 0, %1\n"))
                 : "+v"(addr), "+v"(addr2), "+v"(old), "+v"(old2)
                 : "v"(one)
                 : "memory");
  }
  if (threadIdx.x == 1023 && old + old2 == 77777u)
  {
    sink[0] = double(old);
  }
}


// ---- throughput of single instruction kinds with no dependency between them (8 rotating destinations) ---------------
#define THR_KERNEL(name, OP)                                                      \
  __global__ void __launch_bounds__(1024) name(int iters, double *sink)          \
  {                                                                               \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
             a7 = a0 + 7, b = 3, c = threadIdx.x * 7u;                            \
    unsigned long long m = 0x5555555555555555ull, m1 = 0, m2 = 0, m3 = 0;         \
    for (int i = 0; i < iters; ++i)                                               \
    {                                                                             \
      asm volatile(REP4(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7))          \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m1), \
                     "+s"(m2), "+s"(m3)                                           \
                   : "v"(b), "v"(c), "s"(m)                                       \
                   : "vcc");                                                      \
    }                                                                             \
    if (threadIdx.x == 1023 && double(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + double(m1 + m2 + m3) == 1234.5) \
    {                                                                             \
      sink[0] = double(a0);                                                       \
    }                                                                             \
  }
// operands: %0..%7 rotating VGPRs, %8 %9 %10 SGPR pairs (written), %11 %12 VGPR inputs, %13 SGPR mask input
#define T_CMP_E64(n) "v_cmp_lt_u32_e64 %8, %" #n ", %11\n"
#define T_CMP_VCC(n) "v_cmp_lt_u32_e32 vcc, %" #n ", %11\n"
#define T_CND_E64(n) "v_cndmask_b32_e64 %" #n ", %11, %12, %13\n"
#define T_CND_VCC(n) "v_cndmask_b32_e32 %" #n ", %11, %12, vcc\n"
#define T_CND_ZERO(n) "v_cndmask_b32_e64 %" #n ", 0, %12, %13\n"
#define T_AND_OR(n) "v_and_or_b32 %" #n ", %11, %12, %" #n "\n"
#define T_BFI(n) "v_bfi_b32 %" #n ", %11, %12, %" #n "\n"
#define T_MED3(n) "v_med3_u32 %" #n ", %11, %12, %" #n "\n"
#define T_MIN3(n) "v_min3_u32 %" #n ", %11, %12, %" #n "\n"
#define T_ADD3(n) "v_add3_u32 %" #n ", %11, %12, %" #n "\n"
#define T_LSHL_ADD(n) "v_lshl_add_u32 %" #n ", %11, 2, %" #n "\n"
#define T_MIN(n) "v_min_u32 %" #n ", %11, %" #n "\n"
#define T_XOR(n) "v_xor_b32 %" #n ", %11, %" #n "\n"
#define T_LSHR(n) "v_lshrrev_b32 %" #n ", 5, %11\n"
#define T_BITOP3(n) "v_bitop3_b32 %" #n ", %11, %12, %" #n " bitop3:0x6c\n"
#define T_ADD_SGPR(n) "v_add_u32 %" #n ", s20, %" #n "\n"
#define T_MOV(n) "v_mov_b32 %" #n ", %11\n"
#define T_ADDC(n) "v_addc_co_u32_e64 %" #n ", %9, 0, %" #n ", %13\n"
#define T_SUBREV(n) "v_sub_u32 %" #n ", %11, %" #n "\n"
#define T_CMPX(n) "v_cmpx_lt_u32_e64 %" #n ", %11\n s_mov_b64 exec, -1\n"
THR_KERNEL(thr_cmp_e64, T_CMP_E64)
THR_KERNEL(thr_cmp_vcc, T_CMP_VCC)
THR_KERNEL(thr_cnd_e64, T_CND_E64)
THR_KERNEL(thr_cnd_vcc, T_CND_VCC)
THR_KERNEL(thr_cnd_zero, T_CND_ZERO)
THR_KERNEL(thr_and_or, T_AND_OR)
THR_KERNEL(thr_bfi, T_BFI)
THR_KERNEL(thr_med3, T_MED3)
THR_KERNEL(thr_min3, T_MIN3)
THR_KERNEL(thr_add3, T_ADD3)
THR_KERNEL(thr_lshl_add, T_LSHL_ADD)
THR_KERNEL(thr_min, T_MIN)
THR_KERNEL(thr_xor, T_XOR)
THR_KERNEL(thr_lshr, T_LSHR)
THR_KERNEL(thr_bitop3, T_BITOP3)
THR_KERNEL(thr_add_sgpr, T_ADD_SGPR)
THR_KERNEL(thr_mov, T_MOV)
THR_KERNEL(thr_addc, T_ADDC)
THR_KERNEL(thr_sub, T_SUBREV)

// scalar ALU: a dependent chain of s_and / s_or on 64-bit masks (the walk's mask algebra), all waves of the CU share
// ONE scalar unit
__global__ void __launch_bounds__(1024) dep_salu(int iters, double *sink)
{
  unsigned long long m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = 0x3333;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP32("s_xor_b64 %0, %0, %1\n") : "+s"(m) : "s"(n) : "scc");
  }
  if (threadIdx.x == 1023 && m == 77777u)
  {
    sink[0] = double(m);
  }
}
// a wave-uniform branch per 4 VALU instructions (never taken)
__global__ void __launch_bounds__(1024) valu_branch(int iters, double *sink)
{
  unsigned a = threadIdx.x, b = 3;
  unsigned long long m = 0;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP8("v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %2\n s_cmp_eq_u64 %1, 1\n s_cbranch_scc1 1f\n")
                 "1:\n" : "+v"(a), "+s"(m) : "v"(b) : "scc");
  }
  if (threadIdx.x == 1023 && a == 77777u)
  {
    sink[0] = double(a);
  }
}

// the pair the compiler emits for `cond ? a : b`: VOPC compare into VCC, VOP2 select reading VCC
__global__ void __launch_bounds__(1024) pair_cmp_cnd_vcc(int iters, double *sink)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 3, c = threadIdx.x * 7u;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP4("v_cmp_lt_u32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %4, %5, vcc\n"
                      "v_cmp_lt_u32_e32 vcc, %1, %4\n v_cndmask_b32_e32 %1, %4, %5, vcc\n"
                      "v_cmp_lt_u32_e32 vcc, %2, %4\n v_cndmask_b32_e32 %2, %4, %5, vcc\n"
                      "v_cmp_lt_u32_e32 vcc, %3, %4\n v_cndmask_b32_e32 %3, %4, %5, vcc\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
  }
  if (threadIdx.x == 1023 && a0 + a1 + a2 + a3 == 77777u)
  {
    sink[0] = double(a0);
  }
}
// the same with two plain VALU instructions between the compare and the select
__global__ void __launch_bounds__(1024) pair_cmp_gap_cnd_vcc(int iters, double *sink)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 3, c = threadIdx.x * 7u;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP8("v_cmp_lt_u32_e32 vcc, %0, %4\n v_xor_b32 %1, %4, %1\n v_xor_b32 %2, %4, %2\n"
                      "v_cndmask_b32_e32 %0, %4, %5, vcc\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
  }
  if (threadIdx.x == 1023 && a0 + a1 + a2 + a3 == 77777u)
  {
    sink[0] = double(a0);
  }
}
// select through explicit SGPR pairs (what selectI does), compare and select independent of each other
__global__ void __launch_bounds__(1024) pair_cmp_cnd_sgpr(int iters, double *sink)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 3, c = threadIdx.x * 7u;
  unsigned long long m0 = 0, m1 = 0;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP4("v_cmp_lt_u32_e64 %6, %0, %4\n v_cndmask_b32_e64 %0, %4, %5, %6\n"
                      "v_cmp_lt_u32_e64 %7, %1, %4\n v_cndmask_b32_e64 %1, %4, %5, %7\n"
                      "v_cmp_lt_u32_e64 %6, %2, %4\n v_cndmask_b32_e64 %2, %4, %5, %6\n"
                      "v_cmp_lt_u32_e64 %7, %3, %4\n v_cndmask_b32_e64 %3, %4, %5, %7\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "s"(m0), "s"(m1));
  }
  if (threadIdx.x == 1023 && a0 + a1 + a2 + a3 == 77777u)
  {
    sink[0] = double(a0) + double(m0 + m1);
  }
}
// full-rate select from a VGPR mask: sign mask by arithmetic shift, v_bitop3 as (mask & x) | (~mask & y)
__global__ void __launch_bounds__(1024) select_bitop3(int iters, double *sink)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 3, c = threadIdx.x * 7u, t = 0;
  for (int i = 0; i < iters; ++i)
  {
    asm volatile(REP4("v_sub_u32 %6, %0, %4\n v_ashrrev_i32 %6, 31, %6\n v_bitop3_b32 %0, %6, %4, %5 bitop3:0xca\n"
                      "v_sub_u32 %6, %1, %4\n v_ashrrev_i32 %6, 31, %6\n v_bitop3_b32 %1, %6, %4, %5 bitop3:0xca\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(t));
  }
  if (threadIdx.x == 1023 && a0 + a1 + a2 + a3 == 77777u)
  {
    sink[0] = double(a0);
  }
}

template <typename K>
void run(const char *name, K kernel, double *sink, double instr_per_trip = 32.0)
{
  const int iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  printf("%-18s", name);
  for (int waves = 1; waves <= 8; waves = (waves < 4) ? waves + 1 : waves * 2)
  {
    const int threads = (waves <= 4) ? 256 * waves : 1024;
    const int grid = (waves <= 4) ? 256 : 512;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, iters, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double per_wave = ms * 1e-3 * 2.4e9 / (double(iters) * instr_per_trip);
    printf("  w%d %6.2f/%5.2f", waves, per_wave, per_wave / waves);
  }
  printf("\n");
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  double *sink;
  hipMalloc(&sink, 64);
  printf("cycles per instruction at 2.4 GHz: per wave / per SIMD, at w = 1, 2, 3, 4, 8 waves per SIMD\n");
  run("dep v_add_u32", dep_add_u32, sink);
  run("ind v_add_u32", ind_add_u32, sink);
  run("dep v_add_u32 clamp", dep_add_sat, sink);
  run("ind v_add_u32 clamp", ind_add_sat, sink);
  run("dep v_min3_u32", dep_min3_u32, sink);
  run("ind v_min3_u32", ind_min3_u32, sink);
  run("dep v_fma_f64", dep_fma_f64, sink);
  run("ind v_fma_f64", ind_fma_f64, sink);
  run("dep v_add_f64", dep_add_f64, sink);
  run("ind v_add_f64", ind_add_f64, sink);
  run("dep cmp+cndmask", dep_cmp_cndmask, sink, 64.0);
  run("dep ds_add_rtn", dep_ds_add_rtn, sink, 32.0);
  run("dep ds_add_rtn x2", dep_ds_add_rtn_x2, sink, 64.0);
  printf("independent streams, one instruction kind each\n");
  run("v_cmp_e64->sgpr", thr_cmp_e64, sink);
  run("v_cmp_e32->vcc", thr_cmp_vcc, sink);
  run("v_cndmask_e64 sgpr", thr_cnd_e64, sink);
  run("v_cndmask_e32 vcc", thr_cnd_vcc, sink);
  run("v_cndmask 0,v,s", thr_cnd_zero, sink);
  run("v_and_or_b32", thr_and_or, sink);
  run("v_bfi_b32", thr_bfi, sink);
  run("v_med3_u32", thr_med3, sink);
  run("v_min3_u32", thr_min3, sink);
  run("v_add3_u32", thr_add3, sink);
  run("v_lshl_add_u32", thr_lshl_add, sink);
  run("v_min_u32", thr_min, sink);
  run("v_xor_b32", thr_xor, sink);
  run("v_lshrrev_b32", thr_lshr, sink);
  run("v_bitop3_b32", thr_bitop3, sink);
  run("v_add_u32 sgpr", thr_add_sgpr, sink);
  run("v_mov_b32", thr_mov, sink);
  run("v_addc_co_u32", thr_addc, sink);
  run("v_sub_u32", thr_sub, sink);
  run("cmp_e32+cnd_e32 vcc", pair_cmp_cnd_vcc, sink);
  run("cmp,2 valu,cnd vcc", pair_cmp_gap_cnd_vcc, sink);
  run("cmp_e64+cnd_e64 sgpr", pair_cmp_cnd_sgpr, sink);
  run("sub,ashr,bitop3 sel", select_bitop3, sink, 24.0);
  run("dep s_xor_b64", dep_salu, sink);
  run("3 valu + cmp + br", valu_branch, sink, 40.0);
  return 0;
}
