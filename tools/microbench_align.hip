// tools/microbench_align.hip -- does the 8-byte PHASE of an 8-byte VOP3 instruction cost issue time on gfx950?
// (round 6: the BLS12-381 G2 accumulate kernel -- one wave per SIMD, 18 k v_mad_u64_u32 per iteration -- runs 10 % slower whenever a change
// shifts its instruction stream by an odd number of dwords: 73 % of its multiply-adds sit at addresses = 0 mod 8 in the fast builds, 27 % in
// the slow ones.)  Three loop bodies of 64 v_mad_u64_u32 over 8 independent accumulators, identical but for placement:
//   aligned      .p2align 3; s_nop; s_nop; 64 x v_mad          every multiply-add at 0 mod 8
//   misaligned   .p2align 3; s_nop; 64 x v_mad; s_nop           every multiply-add at 4 mod 8 (same two s_nop per body)
//   mixed        .p2align 3; (v_mad; v_mad; s_nop) x ...        alternating phase
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_align.hip -o tools/microbench_align.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2048;
#define S(x) #x
#define M(n) "v_mad_u64_u32 %" S(n) ", vcc, %8, %9, %" S(n) "\n\t"
#define M8 M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define M64 M8 M8 M8 M8 M8 M8 M8 M8
#define MN(n) "v_mad_u64_u32 %" S(n) ", vcc, %8, %9, %" S(n) "\n\ts_nop 0\n\t"
#define MN8 MN(0) MN(1) MN(2) MN(3) MN(4) MN(5) MN(6) MN(7)
#define KERNEL(NAME, BODY)                                                                                        \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                                     \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                                         \
    uint32_t x = tid * 2654435761u + seed, y = x ^ 0x9e3779b9u;                                                   \
    uint64_t a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;          \
    for (int it = 0; it < ITERS; it++)                                                                            \
      asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)          \
                   : "v"(x), "v"(y) : "vcc");                                                                     \
    out[tid] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);                                                 \
  }
KERNEL(k_aligned, ".p2align 3\n\ts_nop 0\n\ts_nop 0\n\t" M64)
KERNEL(k_misaligned, ".p2align 3\n\ts_nop 0\n\t" M64 "s_nop 0\n\t")
KERNEL(k_nonop_aligned, ".p2align 3\n\t" M64)
// 64 multiply-adds with an s_nop behind every one: phases alternate 0, 4, 0, 4 ...
KERNEL(k_alternating, ".p2align 3\n\t" MN8 MN8 MN8 MN8 MN8 MN8 MN8 MN8)

template <class K>
static double time_kernel(K kern, int nb, uint32_t* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, 7u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, 7u);
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
  const double clk = prop.clockRate * 1e3;
  uint32_t* out;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
#define RUN(NAME, K)                                                                                                     \
  for (int wps : {1, 2, 4, 8}) {                                                                                         \
    const int nb = cus * wps;                                                                                            \
    const double t = time_kernel(K, nb, out);                                                                            \
    const double bodies = (double)nb * 4 * ITERS;   /* wave-bodies */                                                    \
    printf("{\"body\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_body_of_64_mads_at_nominal_clock\": %.1f, \"per_mad\": %.3f}\n", NAME, wps, \
           clk * cus * 4 / (bodies / t), clk * cus * 4 / (bodies / t) / 64);                                             \
  }
  RUN("aligned (2 s_nop + 64 v_mad at 0 mod 8)", k_aligned)
  RUN("misaligned (s_nop + 64 v_mad at 4 mod 8 + s_nop)", k_misaligned)
  RUN("aligned, no s_nop", k_nonop_aligned)
  RUN("alternating (v_mad + s_nop) x 64", k_alternating)
  return 0;
}
