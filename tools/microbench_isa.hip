// tools/microbench_isa.hip -- clean per-instruction issue rates on gfx950 for the instructions the MSM kernels are
// made of.  tools/microbench.hip wraps ONE instruction per asm statement, and hipcc pads every asm statement with an
// s_nop, so its rows are rates of (instruction + s_nop); here a loop body is a SINGLE asm block of 32 instructions
// over 8 independent registers, so nothing but the instruction under test (and the loop's s_add/s_cmp/s_cbranch,
// 3 scalar instructions per 32) is issued.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_isa.hip -o tools/microbench_isa.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 1024;

// 8 x I(0..7) repeated 4 times = 32 instructions
#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define REP32(I) REP8(I) REP8(I) REP8(I) REP8(I)

#define DEF32(NAME, I)                                                                                       \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                                       \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                                    \
    uint32_t x = tid * 2654435761u + seed, y = x ^ 0x9e3779b9u;                                              \
    uint32_t a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;     \
    for (int it = 0; it < ITERS; it++)                                                                       \
      asm volatile(REP32(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(x), "v"(y) : "vcc", "s20", "s21");                                                  \
    out[tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                        \
  }
#define DEF64(NAME, I)                                                                                       \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                                       \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                                    \
    uint32_t x = tid * 2654435761u + seed, y = x ^ 0x9e3779b9u;                                              \
    uint64_t a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;     \
    uint64_t z = ((uint64_t)y << 32) | x;                                                                    \
    for (int it = 0; it < ITERS; it++)                                                                       \
      asm volatile(REP32(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(x), "v"(y), "v"(z) : "vcc", "s20", "s21");                                          \
    out[tid] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);                                            \
  }

#define S(x) #x
#define I_MOV(n) "v_mov_b32 %" S(n) ", %8\n\t"
#define I_ADD(n) "v_add_u32 %" S(n) ", %" S(n) ", %8\n\t"
#define I_SUB(n) "v_sub_u32 %" S(n) ", %" S(n) ", %8\n\t"
#define I_AND(n) "v_and_b32 %" S(n) ", %" S(n) ", %8\n\t"
#define I_ANDK(n) "v_and_b32 %" S(n) ", 0xfffffff, %" S(n) "\n\t"
#define I_SHR(n) "v_lshrrev_b32 %" S(n) ", 28, %" S(n) "\n\t"
#define I_SHL(n) "v_lshlrev_b32 %" S(n) ", 1, %" S(n) "\n\t"
#define I_ALIGN(n) "v_alignbit_b32 %" S(n) ", %8, %" S(n) ", 28\n\t"
#define I_ADD3(n) "v_add3_u32 %" S(n) ", %" S(n) ", %8, %9\n\t"
#define I_OR3(n) "v_or3_b32 %" S(n) ", %" S(n) ", %8, %9\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" S(n) ", %" S(n) ", 1, %8\n\t"
#define I_CNDMASK(n) "v_cndmask_b32 %" S(n) ", %" S(n) ", %8, vcc\n\t"
#define I_CMP(n) "v_cmp_eq_u32 s[20:21], %" S(n) ", %8\n\t"
#define I_MULLO(n) "v_mul_lo_u32 %" S(n) ", %" S(n) ", %8\n\t"
#define I_MULHI(n) "v_mul_hi_u32 %" S(n) ", %" S(n) ", %8\n\t"
#define I_MAD24(n) "v_mad_u32_u24 %" S(n) ", %" S(n) ", %8, %9\n\t"
#define I_XAD(n) "v_xad_u32 %" S(n) ", %" S(n) ", %8, %9\n\t"
#define I_BFE(n) "v_bfe_u32 %" S(n) ", %" S(n) ", 3, 28\n\t"
#define I_MAD64(n) "v_mad_u64_u32 %" S(n) ", vcc, %8, %9, %" S(n) "\n\t"
#define I_MAD64S(n) "v_mad_u64_u32 %" S(n) ", s[20:21], %8, %9, %" S(n) "\n\t"
#define I_MAD64_NOP(n) "v_mad_u64_u32 %" S(n) ", vcc, %8, %9, %" S(n) "\n\ts_nop 0\n\t"
#define I_MAD64_AND(n) "v_mad_u64_u32 %" S(n) ", vcc, %8, %9, %" S(n) "\n\tv_and_b32 %8, %8, %8\n\t"
#define I_SHR64(n) "v_lshrrev_b64 %" S(n) ", 28, %" S(n) "\n\t"
#define I_ADD64(n) "v_lshl_add_u64 %" S(n) ", %" S(n) ", 0, %10\n\t"
#define I_ADDCO(n) "v_add_co_u32 %" S(n) ", vcc, %" S(n) ", %8\n\t"
#define I_ADDC(n) "v_addc_co_u32 %" S(n) ", vcc, %" S(n) ", %8, vcc\n\t"
// one dependent chain (register 0 only): issue-to-issue latency of a dependent v_mad_u64_u32
#define I_MAD64_DEP(n) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"

DEF32(k_mov, I_MOV)
DEF32(k_add, I_ADD)
DEF32(k_sub, I_SUB)
DEF32(k_and, I_AND)
DEF32(k_andk, I_ANDK)
DEF32(k_shr, I_SHR)
DEF32(k_shl, I_SHL)
DEF32(k_align, I_ALIGN)
DEF32(k_add3, I_ADD3)
DEF32(k_or3, I_OR3)
DEF32(k_lshladd, I_LSHLADD)
DEF32(k_cndmask, I_CNDMASK)
DEF32(k_cmp, I_CMP)
DEF32(k_mullo, I_MULLO)
DEF32(k_mulhi, I_MULHI)
DEF32(k_mad24, I_MAD24)
DEF32(k_xad, I_XAD)
DEF32(k_bfe, I_BFE)
DEF32(k_addco, I_ADDCO)
DEF32(k_addc, I_ADDC)
DEF64(k_mad64, I_MAD64)
DEF64(k_mad64s, I_MAD64S)
DEF64(k_mad64_nop, I_MAD64_NOP)
DEF64(k_mad64_and, I_MAD64_AND)
DEF64(k_mad64_dep, I_MAD64_DEP)
DEF64(k_shr64, I_SHR64)
DEF64(k_add64, I_ADD64)

template <class K>
static double time_kernel(K kern, int nb, int block, uint32_t* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(nb), dim3(block), 0, 0, out, 7u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(nb), dim3(block), 0, 0, out, 7u);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3 * 1e-3;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;  // Hz (nominal)
  uint32_t* out;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f}\n", prop.gcnArchName, cus, clk / 1e6);
#define RUN(NAME, K, PER)                                                                                             \
  for (int wps : {1, 2, 8}) {                                                                                         \
    const int nb = cus * wps;                                                                                         \
    const double t = time_kernel(K, nb, 256, out);                                                                    \
    const double ops = (double)nb * 256 * ITERS * 32 * PER;                                                           \
    printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"T_lane_ops_per_s\": %.2f, \"cycles_per_wave_instr_at_nominal_clock\": %.2f}\n", \
           NAME, wps, ops / t / 1e12, clk * cus * 4 / (ops / 64 / t));                                                \
  }
  RUN("v_mov_b32", k_mov, 1)
  RUN("v_add_u32", k_add, 1)
  RUN("v_sub_u32", k_sub, 1)
  RUN("v_and_b32", k_and, 1)
  RUN("v_and_b32(literal)", k_andk, 1)
  RUN("v_lshrrev_b32", k_shr, 1)
  RUN("v_lshlrev_b32", k_shl, 1)
  RUN("v_alignbit_b32", k_align, 1)
  RUN("v_add3_u32", k_add3, 1)
  RUN("v_or3_b32", k_or3, 1)
  RUN("v_lshl_add_u32", k_lshladd, 1)
  RUN("v_cndmask_b32", k_cndmask, 1)
  RUN("v_cmp_eq_u32(sgpr dst)", k_cmp, 1)
  RUN("v_mul_lo_u32", k_mullo, 1)
  RUN("v_mul_hi_u32", k_mulhi, 1)
  RUN("v_mad_u32_u24", k_mad24, 1)
  RUN("v_xad_u32", k_xad, 1)
  RUN("v_bfe_u32", k_bfe, 1)
  RUN("v_add_co_u32", k_addco, 1)
  RUN("v_addc_co_u32", k_addc, 1)
  RUN("v_mad_u64_u32", k_mad64, 1)
  RUN("v_mad_u64_u32(sgpr carry-out)", k_mad64s, 1)
  RUN("v_mad_u64_u32+s_nop(per pair)", k_mad64_nop, 1)
  RUN("v_mad_u64_u32+v_and_b32(per pair)", k_mad64_and, 1)
  RUN("v_mad_u64_u32(one dependent chain)", k_mad64_dep, 1)
  RUN("v_lshrrev_b64", k_shr64, 1)
  RUN("v_lshl_add_u64", k_add64, 1)
  return 0;
}
