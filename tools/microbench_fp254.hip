// tools/microbench_fp254.hip -- limb layouts for a 254-bit base field (the BN254 prime), arithmetic only: ns per Montgomery product
// and ns per XYZZ mixed addition (ec.h xyzz_madd_flag, the accumulate kernel's formula) for
//   9 x 29   carry-free, what the engine uses (FpU<BN254_Fp_U>): 81 + 81 multiply-adds per product, lazy operands on one side only
//   10 x 26  carry-free with room for lazy operands on both sides of a product (LAZY_BOTH): 100 + 100 multiply-adds
//   8 x 32   saturated limbs with explicit carries (fp.h, the boundary field): 64 + 64 multiply-adds + their carry instructions
// Every lane runs a dependent chain on its own operands (nothing wave-uniform); two, three and four waves per SIMD (what the register budgets 256 / 168 / 128 allow).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I constantine_amd/csrc tools/microbench_fp254.hip -o tools/microbench_fp254.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ec.h"
using namespace ctt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// the BN254 base field in 10 limbs of 26 bits (R' = 2^260; generated like field_params.h)
struct BN254_Fp_U26 {
  using Sat = BN254_Fp;
  static constexpr int LB = 26;
  static constexpr int NL = 10;
  static constexpr uint32_t MASK = 0x03ffffffu;
  static constexpr uint32_t M0INV = 0x00866389u;
  static constexpr int RP_OVER_P_LOG2 = 6;
  static constexpr uint32_t P[NL] = {0x007cfd47u, 0x002305b6u, 0x00a8d3c2u, 0x0245a1c7u, 0x0197816au, 0x00605617u, 0x01045b68u, 0x0280a6e1u, 0x0272e131u, 0x000c1913u};
  static constexpr uint32_t ONE[NL] = {0x02fce4b4u, 0x0082203du, 0x009a8455u, 0x0126eaa6u, 0x02498908u, 0x0063c052u, 0x029201d8u, 0x01c93e16u, 0x024e1bb7u, 0x0007c590u};
  static constexpr uint32_t C_IN[NL] = {0x02ec667au, 0x02c3cabeu, 0x030fffbdu, 0x03b6589au, 0x00ad825au, 0x0278a83fu, 0x02f48b70u, 0x038d5c9du, 0x0064ef86u, 0x00035e45u};
};
using F29 = FpU<BN254_Fp_U>;
using F26 = FpU<BN254_Fp_U26>;
using F32 = Fp<BN254_Fp>;
static_assert(!F29::LAZY_BOTH && F26::LAZY_BOTH, "the point of the 26-bit layout");

template <class F> struct LimbInfo { static constexpr int N = F::N; static constexpr uint32_t MASK = 0xffffffffu; };
template <class UP> struct LimbInfo<FpU<UP>> { static constexpr int N = UP::NL; static constexpr uint32_t MASK = UP::MASK; };

template <class F>
__device__ F lane_value(uint32_t tid, uint32_t salt) {
  F r;
  for (int i = 0; i < LimbInfo<F>::N; i++) r.l[i] = (tid * 2654435761u + (uint32_t)i * 40503u + salt * 2246822519u) & LimbInfo<F>::MASK;
  r.l[LimbInfo<F>::N - 1] &= 0xfffu;   // below p
  return r;
}
template <class F>
__device__ uint32_t fold(const F& a) {
  uint32_t s = 0;
  for (int i = 0; i < LimbInfo<F>::N; i++) s ^= a.l[i];
  return s;
}

// V = 0: one dependent product chain; 1: two independent chains (per product); 2: mixed additions
template <class F, int V, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_chain(uint32_t* out, uint32_t seed, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (V == 2) {
    XYZZ<F> acc;
    acc.x = lane_value<F>(tid, seed);
    acc.y = lane_value<F>(tid, seed + 1);
    acc.zz = lane_value<F>(tid, seed + 2);
    acc.zzz = lane_value<F>(tid, seed + 3);
    const F qx = lane_value<F>(tid, seed + 4), qy = lane_value<F>(tid, seed + 5);
    bool empty = false;
    for (int i = 0; i < iters; i++) xyzz_madd_flag<F>(acc, empty, qx, qy, (i & 1) != 0);
    out[tid] = fold(acc.x) ^ fold(acc.y) ^ fold(acc.zz) ^ fold(acc.zzz) ^ (uint32_t)empty;
  } else {
    F a = lane_value<F>(tid, seed), b = lane_value<F>(tid, seed + 1);
    const F a0 = a, b0 = b;
    for (int i = 0; i < iters; i++) {
      if constexpr (V == 0) a = F::mul(a, b);
      if constexpr (V == 1) {
        F c, d;
        fmul_pair<F>(a, a0, b, b0, c, d);
        a = c;
        b = d;
      }
    }
    out[tid] = fold(a) ^ fold(b);
  }
}

template <class K>
static double time_kernel(K kern, int nb, uint32_t* out, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, 7u, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, 7u, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3 * 1e-3;
}

template <class F, int WAVES>
static void run(const char* name, int cus, uint32_t* out) {
  const int nb = cus * WAVES;   // 256 threads = one wave per SIMD per block
  const int iters = 128;
  const double lanes = (double)nb * 256;
  const double t0 = time_kernel(k_chain<F, 0, WAVES>, nb, out, iters);
  const double t1 = time_kernel(k_chain<F, 1, WAVES>, nb, out, iters);
  const double t2 = time_kernel(k_chain<F, 2, WAVES>, nb, out, iters);
  int regs = 0;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_chain<F, 2, WAVES>)));
  regs = fa.numRegs;
  printf("{\"layout\": \"%s\", \"waves_per_simd\": %d, \"G_products_per_s_one_chain\": %.1f, \"G_products_per_s_paired\": %.1f, "
         "\"G_mixed_additions_per_s\": %.2f, \"ns_per_mixed_addition_whole_chip\": %.4f, \"madd_kernel_vgprs\": %d, \"madd_kernel_scratch_bytes\": %d}\n",
         name, WAVES, lanes * iters / t0 / 1e9, lanes * iters * 2 / t1 / 1e9, lanes * iters / t2 / 1e9, t2 / (lanes * iters) * 1e9, regs,
         (int)fa.localSizeBytes);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* out;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  run<F29, 2>("9x29 carry-free", cus, out);
  run<F29, 4>("9x29 carry-free", cus, out);
  run<F29, 3>("9x29 carry-free", cus, out);
  run<F26, 2>("10x26 carry-free, lazy both sides", cus, out);
  run<F26, 4>("10x26 carry-free, lazy both sides", cus, out);
  run<F26, 3>("10x26 carry-free, lazy both sides", cus, out);
  run<F32, 2>("8x32 saturated, explicit carries", cus, out);
  run<F32, 4>("8x32 saturated, explicit carries", cus, out);
  run<F32, 3>("8x32 saturated, explicit carries", cus, out);
  return 0;
}
