// tools/microbench_mfma.hip -- exploratory (round 3): can the CONSTANT-operand half of the carry-free field arithmetic -- the m*p
// products of the Montgomery reduction, 1764 of the 3542 v_mad_u64_u32 of a mixed addition -- run on the MFMA pipe as an int8
// Toeplitz product next to the VALU chains?
//
// A product of a per-lane 392-bit number q (14 limbs of 28 bits) with a constant is, in radix 2^7, U[k][lane] = sum_i P[k][i] *
// Q[i][lane]: a 112 x 56 constant Toeplitz matrix times a 56 x 64 matrix of digits (one column per lane) -- with 32x32x16 int8
// tiles that is 4 (rows) x 2 (lane halves) x 4 (K) = 32 v_mfma_i32_32x32x16_i8 per wave and product.  What has to be paid on
// the VALU besides: q into 56 seven-bit digits in the B-operand layout, and the 112 int32 digit sums back into 28-bit limbs.
// Three measurements:
//   (1) v_mad_u64_u32 alone, v_mfma_i32_32x32x16_i8 alone, and both interleaved in one instruction stream: do the pipes overlap?
//   (2) the repack on its own (digits out + digit sums in, per product, WITHOUT the cross-lane exchange the operand layouts
//       also need): VALU instructions and time per product, against the 196 + 14 multiply-adds it would replace;
//   (3) the arithmetic of it: 32 MFMAs per product at the measured rate against 196 multiply-adds at theirs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_mfma.hip -o tools/microbench_mfma.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ITERS = 512;

// (1a) 32 independent-ish multiply-adds per iteration (8 accumulators x 4)
__global__ void k_mad(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = tid * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  uint64_t a[8];
  for (int i = 0; i < 8; i++) a[i] = x + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
// (1b) NM MFMAs per iteration on two accumulator tiles
template <int NM>
__global__ void k_mfma(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  long av = (long)(tid * 2654435761u + seed) | 0x0101010101010101L, bv = av ^ 0x0202020202020202L;
  v16i c0 = {0}, c1 = {0};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < NM / 2; r++) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, bv, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(bv, av, c1, 0, 0, 0);
    }
  }
  int s = 0;
  for (int i = 0; i < 16; i++) s ^= c0[i] ^ c1[i];
  out[tid] = (uint32_t)s;
}
// (1c) both in one stream: 32 multiply-adds and NM MFMAs per iteration, interleaved
template <int NM>
__global__ void k_both(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = tid * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  long av = (long)x | 0x0101010101010101L, bv = av ^ 0x0202020202020202L;
  uint64_t a[8];
  for (int i = 0; i < 8; i++) a[i] = x + i;
  v16i c0 = {0}, c1 = {0};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int r = 0; r < NM / 2; r++) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, bv, c0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 32 / NM; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[(2 * r * (32 / NM) + i) & 7]) : "v"(x), "v"(y) : "vcc");
      c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(bv, av, c1, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 32 / NM; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[((2 * r + 1) * (32 / NM) + i) & 7]) : "v"(x), "v"(y) : "vcc");
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  int t = 0;
  for (int i = 0; i < 16; i++) t ^= c0[i] ^ c1[i];
  out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)t;
}

// (2) the repack of ONE constant product per lane and iteration:
//     14 limbs of 28 bits -> 14 registers of four 7-bit digits in byte lanes (the int8 operand), and
//     112 int32 digit sums (< 2^20: 56 terms of 127 * 127) -> 28 limbs of 28 bits (28 bits = four digits) with carries
__device__ __forceinline__ uint32_t spread7(uint32_t l) {
  // bits [0,7) [7,14) [14,21) [21,28) -> bytes 0..3
  return (l & 0x7fu) | ((l & 0x3f80u) << 1) | ((l & 0x1fc000u) << 2) | ((l & 0xfe00000u) << 3);
}
__global__ void k_repack(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t l[14];
  for (int i = 0; i < 14; i++) l[i] = (tid * 2654435761u + seed + 977u * i) & 0x0fffffffu;
  uint32_t acc = 0;
  for (int it = 0; it < ITERS; it++) {
    uint32_t b[14];
#pragma unroll
    for (int i = 0; i < 14; i++) b[i] = spread7(l[i]);          // digits out
    // stand-in for the 112 digit sums the MFMA tiles would deliver: any values < 2^20 that depend on the digits
    uint32_t s[112];
#pragma unroll
    for (int k = 0; k < 112; k++) s[k] = (b[k % 14] >> (k & 7)) & 0xfffffu;
    // digit sums in: limb j = s[4j] + s[4j+1] 2^7 + s[4j+2] 2^14 + s[4j+3] 2^21 + carry, 28 bits kept
    uint64_t carry = 0;
    uint32_t t[28];
#pragma unroll
    for (int j = 0; j < 28; j++) {
      uint64_t v = carry + s[4 * j] + ((uint64_t)s[4 * j + 1] << 7) + ((uint64_t)s[4 * j + 2] << 14) + ((uint64_t)s[4 * j + 3] << 21);
      t[j] = (uint32_t)v & 0x0fffffffu;
      carry = v >> 28;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) l[i] = t[i] ^ t[14 + i];       // (keeps every limb live)
    acc ^= (uint32_t)carry;
  }
  uint32_t r = acc;
  for (int i = 0; i < 14; i++) r ^= l[i];
  out[tid] = r;
}
// the 196 + 14 multiply-adds the repack + MFMA would replace (u = q * p with the quotient digits, as in fpu.h col_finish)
__global__ void k_mp_valu(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m[14], p[14];
  for (int i = 0; i < 14; i++) { m[i] = (tid * 2654435761u + seed + 977u * i) & 0x0fffffffu; p[i] = (0x9e3779b9u * (i + 1)) & 0x0fffffffu; }
  for (int it = 0; it < ITERS; it++) {
    uint64_t acc = 0;
    uint32_t t[14];
#pragma unroll
    for (int k = 0; k < 27; k++) {
#pragma unroll
      for (int i = 0; i < 14; i++) {
        const int j = k - i;
        if (j >= 0 && j < 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[i]), "v"(p[j]) : "vcc");
      }
      if (k >= 13) t[k - 13] = (uint32_t)acc & 0x0fffffffu;
      acc >>= 28;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) m[i] = t[i];
  }
  uint32_t r = 0;
  for (int i = 0; i < 14; i++) r ^= m[i];
  out[tid] = r;
}

template <class K>
static double run(K kernel, int waves_per_simd, uint32_t* d_out) {
  int dev = 0, ncu = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = ncu * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, d_out, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, d_out, 2u + rep);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e6 / ITERS;   // ns per loop iteration (per wave, with waves_per_simd waves sharing a SIMD)
}

int main() {
  uint32_t* d_out;
  CK(hipMalloc(&d_out, 1u << 24));
  for (int w : {1, 2}) {
    const double mad = run(k_mad, w, d_out), mf8 = run(k_mfma<8>, w, d_out), mf16 = run(k_mfma<16>, w, d_out);
    const double b8 = run(k_both<8>, w, d_out), b16 = run(k_both<16>, w, d_out);
    printf("{\"bench\": \"mad_vs_mfma_i8\", \"waves_per_simd\": %d, \"ns_per_iter\": {\"32_mad_u64_u32\": %.1f, \"8_mfma_i32_32x32x16_i8\": %.1f, "
           "\"16_mfma\": %.1f, \"32_mad_and_8_mfma_interleaved\": %.1f, \"32_mad_and_16_mfma_interleaved\": %.1f}, "
           "\"ns_per_mad\": %.2f, \"ns_per_mfma\": %.2f, \"overlap_8\": \"%.0f %% of the shorter stream hidden\", \"overlap_16\": \"%.0f %%\"}\n",
           w, mad, mf8, mf16, b8, b16, mad / 32, mf16 / 16, 100.0 * (mad + mf8 - b8) / (mad < mf8 ? mad : mf8),
           100.0 * (mad + mf16 - b16) / (mad < mf16 ? mad : mf16));
    const double rp = run(k_repack, w, d_out), mp = run(k_mp_valu, w, d_out);
    printf("{\"bench\": \"constant_product_392bit\", \"waves_per_simd\": %d, \"ns_per_product\": {\"valu_196_mad_u64_u32_plus_column_shifts\": %.1f, "
           "\"repack_only_digits_out_and_digit_sums_in\": %.1f, \"32_mfma_at_measured_rate\": %.1f}, "
           "\"note\": \"the repack alone (no cross-lane exchange for the MFMA operand layouts, no MFMA) against the VALU product it would replace\"}\n",
           w, mp, rp, 32.0 * mf16 / 16);
    fflush(stdout);
  }
  return 0;
}
