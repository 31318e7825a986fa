// tools/microbench.hip -- gfx950 instruction-rate and field-arithmetic micro-benchmarks.
// Measures the integer-multiply roof the MSM kernels are bound by ("must be measured").
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I constantine_amd/csrc tools/microbench.hip -o tools/microbench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "generators.h"

using namespace ctt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 8;  // independent chains per lane

#define DEF_KERNEL(NAME, DECL, BODY, SINK)                                              \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                  \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                               \
    DECL;                                                                               \
    for (int it = 0; it < ITERS; it++) {                                                \
      BODY;                                                                             \
    }                                                                                   \
    out[tid] = SINK;                                                                    \
  }

// --- v_mad_u64_u32 ---
DEF_KERNEL(k_mad64,
  uint64_t a[UNROLL]; uint32_t x = tid * 2654435761u + seed; uint32_t y = x ^ 0x9e3779b9u;
  for (int i = 0; i < UNROLL; i++) a[i] = x + i,
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc"),
  (uint32_t)(a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]))

// --- mad + addc pair (the MAC of fp.h) ---
DEF_KERNEL(k_mac,
  uint64_t a[UNROLL]; uint32_t h[UNROLL]; uint32_t x = tid * 2654435761u + seed; uint32_t y = x ^ 0x9e3779b9u;
  for (int i = 0; i < UNROLL; i++) { a[i] = x + i; h[i] = i; },
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a[i]), "+v"(h[i]) : "v"(x), "v"(y) : "vcc"),
  (uint32_t)(a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]) ^ h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5] ^ h[6] ^ h[7])

// --- co-issue probes: does another pipe run underneath the 64-bit integer multiplier? -------------------------
// 8 independent v_mad_u64_u32 chains interleaved 1:1 with 8 independent chains of a second instruction
DEF_KERNEL(k_mad64_plus_fma64,
  uint64_t a[UNROLL]; double f[UNROLL]; uint32_t x = tid * 2654435761u + seed; uint32_t y = x ^ 0x9e3779b9u; double g = 1.0 + x * 1e-9;
  for (int i = 0; i < UNROLL; i++) { a[i] = x + i; f[i] = g + i; },
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_fma_f64 %1, %1, %4, %1" : "+v"(a[i]), "+v"(f[i]) : "v"(x), "v"(y), "v"(g) : "vcc"),
  (uint32_t)(a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]) ^ (uint32_t)(f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7]))
DEF_KERNEL(k_mad64_plus_add32,
  uint64_t a[UNROLL]; uint32_t b[UNROLL]; uint32_t x = tid * 2654435761u + seed; uint32_t y = x ^ 0x9e3779b9u;
  for (int i = 0; i < UNROLL; i++) { a[i] = x + i; b[i] = i; },
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2" : "+v"(a[i]), "+v"(b[i]) : "v"(x), "v"(y) : "vcc"),
  (uint32_t)(a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]) ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ b[4] ^ b[5] ^ b[6] ^ b[7])
DEF_KERNEL(k_mad64_plus_mad24,
  uint64_t a[UNROLL]; uint32_t b[UNROLL]; uint32_t x = tid * 2654435761u + seed; uint32_t y = x ^ 0x9e3779b9u;
  for (int i = 0; i < UNROLL; i++) { a[i] = x + i; b[i] = i; },
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u32_u24 %1, %1, %2, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(x), "v"(y) : "vcc"),
  (uint32_t)(a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]) ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ b[4] ^ b[5] ^ b[6] ^ b[7])

#define DEF32(NAME, ASM)                                                                \
  DEF_KERNEL(NAME,                                                                      \
    uint32_t a[UNROLL]; uint32_t x = tid * 2654435761u + seed;                          \
    for (int i = 0; i < UNROLL; i++) a[i] = x + i,                                      \
    _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile(ASM : "+v"(a[i]) : "v"(x) : "vcc"), \
    a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7])

DEF32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
DEF32(k_add_u32, "v_add_u32 %0, %0, %1")
DEF32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
DEF32(k_addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
DEF32(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEF32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %0")
DEF32(k_mov, "v_mov_b32 %0, %1")
DEF32(k_mul_i32_i24, "v_mul_i32_i24 %0, %0, %1")
DEF32(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %0")

// 64-bit ops
DEF_KERNEL(k_lshl_add_u64,
  uint64_t a[UNROLL]; uint64_t x = tid * 2654435761ull + seed;
  for (int i = 0; i < UNROLL; i++) a[i] = x + i,
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(x)),
  (uint32_t)(a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]))

DEF_KERNEL(k_fma_f64,
  double a[UNROLL]; double x = 1.0 + 1e-9 * tid; double y = 1e-12 * seed;
  for (int i = 0; i < UNROLL; i++) a[i] = x + i,
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y)),
  (uint32_t)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))

DEF_KERNEL(k_fma_f32,
  float a[UNROLL]; float x = 1.0f + 1e-6f * tid; float y = 1e-9f * seed;
  for (int i = 0; i < UNROLL; i++) a[i] = x + i,
  _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y)),
  (uint32_t)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))

// --- field arithmetic chains ---
template <class F, int OP>
__global__ void k_field_chain(uint32_t* out, uint32_t seed, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  F a = F::one(), b = F::one();
  a.l[0] ^= tid & 0xffff;
  b.l[1] ^= seed & 0xffff;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) a = F::mul(a, b);
    if (OP == 1) a = F::sqr(a);
    if (OP == 2) a = F::add(a, b);
    if (OP == 3) a = F::sub(a, b);
  }
  uint32_t s = 0;
  for (int i = 0; i < F::N; i++) s ^= a.l[i];
  out[tid] = s;
}

template <class K, class... A>
static double time_kernel(K kern, dim3 grid, dim3 block, int reps, A... args) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, grid, block, 0, 0, args...);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, grid, block, 0, 0, args...);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e-3;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;  // Hz
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f}\n", prop.name, cus, clk / 1e6);
  const int block = 256;
  const int blocks = cus * 8;  // 8 waves per SIMD
  uint32_t* out;
  CK(hipMalloc(&out, (size_t)blocks * block * 4));

#define RUN(NAME, K, OPS_PER_ITER)                                                                            \
  {                                                                                                           \
    double t = time_kernel(K, dim3(blocks), dim3(block), 5, out, 12345u);                                     \
    double ops = (double)blocks * block * ITERS * UNROLL * (OPS_PER_ITER);                                    \
    double rate = ops / t;                                                                                    \
    printf("{\"instr\": \"%s\", \"Gops_per_s\": %.1f, \"lanes_per_clk_per_cu\": %.2f}\n", NAME, rate / 1e9,   \
           rate / clk / cus);                                                                                 \
  }
  RUN("v_mad_u64_u32", k_mad64, 1)
  RUN("mac(v_mad_u64_u32+v_addc_co_u32)", k_mac, 1)
  RUN("pair(v_mad_u64_u32 + v_fma_f64), pairs/s", k_mad64_plus_fma64, 1)
  RUN("pair(v_mad_u64_u32 + v_add_u32), pairs/s", k_mad64_plus_add32, 1)
  RUN("pair(v_mad_u64_u32 + v_mad_u32_u24), pairs/s", k_mad64_plus_mad24, 1)
  RUN("v_mul_lo_u32", k_mul_lo, 1)
  RUN("v_mul_hi_u32", k_mul_hi, 1)
  RUN("v_add_u32", k_add_u32, 1)
  RUN("v_add_co_u32", k_add_co, 1)
  RUN("v_addc_co_u32", k_addc_co, 1)
  RUN("v_mad_u32_u24", k_mad_u32_u24, 1)
  RUN("v_mul_hi_u32_u24", k_mul_hi_u24, 1)
  RUN("v_mad_u32_u16", k_mad_u32_u16, 1)
  RUN("v_mul_i32_i24", k_mul_i32_i24, 1)
  RUN("v_mad_i32_i24", k_mad_i32_i24, 1)
  RUN("v_mov_b32", k_mov, 1)
  RUN("v_lshl_add_u64", k_lshl_add_u64, 1)
  RUN("v_fma_f64", k_fma_f64, 1)
  RUN("v_fma_f32", k_fma_f32, 1)

  // field chains at several occupancies (waves per SIMD = blocks*4/ (cus*4) with 256-thread blocks)
  const int fiters = 256;
#define RUNF(NAME, F, OP, WPS)                                                                                    \
  {                                                                                                               \
    int nb = cus * (WPS);                                                                                         \
    double t = time_kernel(k_field_chain<F, OP>, dim3(nb), dim3(block), 3, out, 7u, fiters);                      \
    double ops = (double)nb * block * fiters;                                                                     \
    printf("{\"field_op\": \"%s\", \"waves_per_simd\": %d, \"Gops_per_s\": %.2f, \"ns_per_wave_op\": %.1f}\n", NAME, \
           WPS, ops / t / 1e9, t / fiters * 1e9);                                                                 \
  }
  using F381 = Fp<BLS12_381_Fp>;
  using F254 = Fp<BN254_Fp>;
  for (int wps : {1, 2, 3, 4, 8}) {
    RUNF("bls12_381_fp_mul", F381, 0, wps)
    RUNF("bls12_381_fp_sqr", F381, 1, wps)
  }
  RUNF("bls12_381_fp_add", F381, 2, 8)
  RUNF("bls12_381_fp_sub", F381, 3, 8)
  for (int wps : {1, 2, 4, 8}) {
    RUNF("bn254_fp_mul", F254, 0, wps)
    RUNF("bn254_fp_sqr", F254, 1, wps)
  }

  return 0;
}
