// tools/microbench_fpu.hip -- throughput of the carry-free Montgomery multiplier variants at several occupancies.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I constantine_amd/csrc tools/microbench_fpu.hip -o tools/microbench_fpu.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "fpu.h"
using namespace ctt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using UP = BLS12_381_Fp_U;
using FU = FpU<UP>;
constexpr int NL = UP::NL, LB = UP::LB;
constexpr uint32_t MASK = UP::MASK;

// variant B: separate accumulators for the a*b and the m*p products of a column
__device__ __forceinline__ FU mul_two_acc(const FU& a, const FU& b) {
  uint64_t carry = 0;
  uint32_t m[NL];
  FU t;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    uint64_t s1 = carry, s2 = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j >= 0 && j < NL) s1 += (uint64_t)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j >= 1 && j < NL && i < (k < NL ? k : NL)) s2 += (uint64_t)m[i] * UP::P[j];
    }
    uint64_t acc = s1 + s2;
    if (k < NL) {
      m[k] = ((uint32_t)acc * UP::M0INV) & MASK;
      acc += (uint64_t)m[k] * UP::P[0];
    } else {
      t.l[k - NL] = (uint32_t)acc & MASK;
    }
    carry = acc >> LB;
  }
  t.l[NL - 1] = (uint32_t)carry;
  return t;
}

// variant C: even/odd split of both product families (4 chains)
__device__ __forceinline__ FU mul_four_acc(const FU& a, const FU& b) {
  uint64_t carry = 0;
  uint32_t m[NL];
  FU t;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    uint64_t s[4] = {carry, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j >= 0 && j < NL) s[i & 1] += (uint64_t)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j >= 1 && j < NL && i < (k < NL ? k : NL)) s[2 + (i & 1)] += (uint64_t)m[i] * UP::P[j];
    }
    uint64_t acc = (s[0] + s[1]) + (s[2] + s[3]);
    if (k < NL) {
      m[k] = ((uint32_t)acc * UP::M0INV) & MASK;
      acc += (uint64_t)m[k] * UP::P[0];
    } else {
      t.l[k - NL] = (uint32_t)acc & MASK;
    }
    carry = acc >> LB;
  }
  t.l[NL - 1] = (uint32_t)carry;
  return t;
}

// variant S: 13 signed 30-bit limbs (390-bit radix), v_mad_i64_i32 accumulation; 338 multiplies instead of 392.
// Throughput experiment only (round-2 candidate representation for BLS12-381).
constexpr int NS = 13, SB = 30;
struct FS { int32_t l[NS]; };
__device__ constexpr int32_t PS[NS] = {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811,
                                       316751073, -321428361, 517541167, -375082566, -91332614, 1704210};
constexpr uint32_t PS_M0INV = 1073545213u;
__device__ __forceinline__ int32_t sext30(uint32_t x) { return (int32_t)(x << 2) >> 2; }
__device__ __forceinline__ FS mul_signed13(const FS& a, const FS& b) {
  int64_t carry = 0;
  int32_t m[NS];
  FS t;
#pragma unroll
  for (int k = 0; k < 2 * NS - 1; k++) {
    int64_t acc = carry;
#pragma unroll
    for (int i = 0; i < NS; i++) {
      const int j = k - i;
      if (j >= 0 && j < NS) acc += (int64_t)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NS; i++) {
      const int j = k - i;
      if (j >= 1 && j < NS && i < (k < NS ? k : NS)) acc += (int64_t)m[i] * PS[j];
    }
    if (k < NS) {
      m[k] = sext30((uint32_t)acc * PS_M0INV);
      acc += (int64_t)m[k] * PS[0];
      carry = acc >> SB;
    } else {
      int32_t d = sext30((uint32_t)acc);
      t.l[k - NS] = d;
      carry = (acc - d) >> SB;
    }
  }
  t.l[NS - 1] = (int32_t)carry;
  return t;
}

template <int V>
__global__ void k_chain_s(uint32_t* out, uint32_t seed, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  FS a, b;
  for (int i = 0; i < NS; i++) {
    a.l[i] = sext30(tid * 2654435761u + i * 40503u + seed) >> 1;
    b.l[i] = sext30(tid * 2246822519u + i * 69069u + seed * 3u) >> 1;
  }
  const FS a0 = a, b0 = b;
  for (int i = 0; i < iters; i++) {
    if (V == 0) a = mul_signed13(a, b);
    if (V == 1) { FS c = mul_signed13(a, a0); b = mul_signed13(b, b0); a = c; }
  }
  uint32_t s = 0;
  for (int i = 0; i < NS; i++) s ^= (uint32_t)a.l[i] ^ (uint32_t)b.l[i];
  out[tid] = s;
}

template <int V>
__global__ void k_chain(uint32_t* out, uint32_t seed, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  // every limb lane-varying and the two operands unrelated, so that nothing is wave-uniform or shared
  FU a, b;
  for (int i = 0; i < NL; i++) {
    a.l[i] = (tid * 2654435761u + i * 40503u + seed) & MASK;
    b.l[i] = (tid * 2246822519u + i * 69069u + seed * 3u) & MASK;
  }
  const FU a0 = a, b0 = b;
  for (int i = 0; i < iters; i++) {
    if (V == 0) a = FU::mul(a, b);
    if (V == 1) a = mul_two_acc(a, b);
    if (V == 2) a = mul_four_acc(a, b);
    if (V == 3) a = FU::sqr(a);
    if (V == 4) { FU c = FU::mul(a, a0); b = FU::mul(b, b0); a = c; }   // two independent products per iteration
    if (V == 5) { FU c, d; FU::mul_pair(a, a0, b, b0, c, d); a = c; b = d; }
    if (V == 6) { FU c, d; FU::sqr_pair(a, b, c, d); a = c; b = d; }
    if (V == 7) { a = FU::mul2(a, a0, b, b0); }
  }
  uint32_t s = 0;
  for (int i = 0; i < NL; i++) s ^= a.l[i] ^ b.l[i];
  out[tid] = s;
}

template <class K>
static double time_kernel(K kern, int nb, int block, uint32_t* out, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(nb), dim3(block), 0, 0, out, 7u, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(nb), dim3(block), 0, 0, out, 7u, iters);
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
  uint32_t* out;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  const int iters = 256;
  const char* names[] = {"fpu_mul", "fpu_mul_two_acc", "fpu_mul_four_acc", "fpu_sqr", "fpu_mul_x2_independent",
                         "fpu_mul_pair(per product)", "fpu_sqr_pair(per square)", "fpu_mul2(per call)"};
  for (int wps : {1, 2, 4, 8}) {
    int nb = cus * wps;
    double t[8];
    t[0] = time_kernel(k_chain<0>, nb, 256, out, iters);
    t[1] = time_kernel(k_chain<1>, nb, 256, out, iters);
    t[2] = time_kernel(k_chain<2>, nb, 256, out, iters);
    t[3] = time_kernel(k_chain<3>, nb, 256, out, iters);
    t[4] = time_kernel(k_chain<4>, nb, 256, out, iters) / 2;
    t[5] = time_kernel(k_chain<5>, nb, 256, out, iters) / 2;
    t[6] = time_kernel(k_chain<6>, nb, 256, out, iters) / 2;
    t[7] = time_kernel(k_chain<7>, nb, 256, out, iters);
    printf("{\"field_op\": \"signed13x30_mul\", \"waves_per_simd\": %d, \"Gops_per_s\": %.2f}\n", wps,
           (double)nb * 256 * iters / time_kernel(k_chain_s<0>, nb, 256, out, iters) / 1e9);
    printf("{\"field_op\": \"signed13x30_mul_x2_independent\", \"waves_per_simd\": %d, \"Gops_per_s\": %.2f}\n", wps,
           (double)nb * 256 * iters * 2 / time_kernel(k_chain_s<1>, nb, 256, out, iters) / 1e9);
    for (int v = 0; v < 8; v++)
      printf("{\"field_op\": \"%s\", \"waves_per_simd\": %d, \"Gops_per_s\": %.2f}\n", names[v], wps,
             (double)nb * 256 * iters / t[v] / 1e9);
  }
  return 0;
}
