// tools/cu_mask_probe.hip -- what a CU-masked stream (hipExtStreamCreateWithCUMask) does on this GPU:
//   (1) which compute units a mask selects: every workgroup of a launch records HW_REG_XCC_ID and HW_REG_HW_ID (SE / SH / CU),
//       the histogram per mask is printed -- the engine reserves "r CUs per XCD" as the low 8 r bits of the mask and relies on
//       the user mask being dealt round-robin over the XCDs;
//   (2) what a long issue-bound kernel (a stand-in for the bucket accumulation: one-wave workgroups of dependent
//       v_mad_u64_u32, every wave slot of its CUs taken) does to a chain of short dependent launches on another stream
//       (a stand-in for the narrow reduction passes), unmasked against masked.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cu_mask_probe.hip -o tools/cu_mask_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_where(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // a little work so that the launch spreads over every CU it may use instead of draining through the first few
  uint64_t a = threadIdx.x + 1;
  for (int i = 0; i < spin; i++) a = a * 6364136223846793005ull + 1442695040888963407ull;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw ^ (uint32_t)(a & 0);   // (a & 0: keeps the loop alive)
  }
}

// one-wave workgroups, `iters` x 32 dependent-free multiply-adds per lane: issue-bound like k_accum
__global__ void __launch_bounds__(64) k_busy(uint32_t* out, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = tid * 2654435761u + 12345u, y = x ^ 0x9e3779b9u;
  uint64_t a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      a0 = (uint64_t)x * (uint32_t)y + a0; a1 = (uint64_t)x * (uint32_t)y + a1; a2 = (uint64_t)x * (uint32_t)y + a2; a3 = (uint64_t)x * (uint32_t)y + a3;
      a4 = (uint64_t)x * (uint32_t)y + a4; a5 = (uint64_t)x * (uint32_t)y + a5; a6 = (uint64_t)x * (uint32_t)y + a6; a7 = (uint64_t)x * (uint32_t)y + a7;
      x += (uint32_t)a0;
    }
  }
  out[tid] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
// a short latency-bound kernel: few waves, one dependent chain each (~ one four-lane EC addition: a few microseconds)
__global__ void __launch_bounds__(64) k_short(uint32_t* out, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t a = tid + 1;
  uint32_t x = tid * 2654435761u + 1u;
  for (int it = 0; it < iters; it++) a = (uint64_t)x * (uint32_t)a + (a >> 7);
  out[tid] = (uint32_t)a;
}

static hipStream_t masked_stream(const std::vector<uint32_t>& mask) {
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  return s;
}

// a tiny kernel of wide workgroups (the sort's scan kernels: 20 x 1024 lanes, a few microseconds)
__global__ void __launch_bounds__(1024) k_tiny_wide(uint32_t* out, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = tid;
  for (int it = 0; it < iters; it++) a = a * 1664525u + 1013904223u;
  out[tid] = a;
}

int main(int argc, char** argv) {
  if (argc > 1) setenv("GPU_MAX_HW_QUEUES", argv[1], 1);   // (before the first HIP call; default: the runtime's four)
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int words = (ncu + 31) / 32;
  printf("# device: %s, %d CUs, mask words %d\n", prop.name, ncu, words);
  uint32_t* d_out;
  CK(hipMalloc(&d_out, 64u << 20));
  std::vector<uint32_t> h(2 * 8192);

  auto low_bits = [&](int nbits) {
    std::vector<uint32_t> m(words, 0u);
    for (int b = 0; b < nbits && b < ncu; b++) m[b / 32] |= 1u << (b % 32);
    return m;
  };
  auto complement = [&](const std::vector<uint32_t>& m) {
    std::vector<uint32_t> c(words, 0u);
    for (int b = 0; b < ncu; b++) if (!(m[b / 32] >> (b % 32) & 1u)) c[b / 32] |= 1u << (b % 32);
    return c;
  };

  // ---- (1) placement ------------------------------------------------------------------------------------------------
  for (int r : {0, 1, 2, 4, 8}) {
    for (int comp = 0; comp < (r ? 2 : 1); comp++) {
      hipStream_t s;
      std::vector<uint32_t> m = low_bits(8 * r);
      if (comp) m = complement(m);
      if (r == 0) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      else s = masked_stream(m);
      const int nblk = 8192;
      hipLaunchKernelGGL(k_where, dim3(nblk), dim3(64), 0, s, d_out, 20000);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h.data(), d_out, nblk * 8, hipMemcpyDeviceToHost));
      std::map<uint32_t, int> per_xcc;
      std::map<uint64_t, int> per_cu;
      for (int b = 0; b < nblk; b++) {
        const uint32_t xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc]++;
        per_cu[((uint64_t)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
      }
      printf("mask %s%d CUs/XCD (bits %s): distinct CUs used %zu; per XCD:", comp ? "complement of low " : "low ", r,
             r == 0 ? "none" : (comp ? "[8r,256)" : "[0,8r)"), per_cu.size());
      for (auto& kv : per_xcc) {
        int cus = 0;
        for (auto& c : per_cu) if ((c.first >> 16) == kv.first) cus++;
        printf(" x%u:%dcu", kv.first, cus);
      }
      printf("\n");
      if (r == 1 && !comp) {
        printf("   (the 8 CUs of the low-8-bits mask: ");
        for (auto& c : per_cu) printf("xcc%llu/se%llu/sh%llu/cu%llu ", (unsigned long long)(c.first >> 16), (unsigned long long)((c.first >> 8) & 0xff),
                                      (unsigned long long)((c.first >> 4) & 0xf), (unsigned long long)(c.first & 0xf));
        printf(")\n");
      }
      CK(hipStreamDestroy(s));
    }
  }

  // ---- (2) a chain of short dependent launches beside a long issue-bound kernel ---------------------------------------
  // occupancy of k_busy: waves per SIMD it reaches (few registers: 8); cap the grid at 2 waves per SIMD like k_accum
  auto time_chain = [&](hipStream_t sbusy, hipStream_t schain, int busy_cus, bool with_busy, int chain_waves, const char* label) {
    hipEvent_t e0, e1, b0, b1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const int busy_blocks = busy_cus * 4 * 2;          // 2 waves per SIMD
    const int chain_len = 20;
    float chain_ms = 0, busy_ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      if (with_busy) {
        CK(hipEventRecord(b0, sbusy));
        hipLaunchKernelGGL(k_busy, dim3(busy_blocks), dim3(64), 0, sbusy, d_out + (8u << 20), 60000);
        CK(hipEventRecord(b1, sbusy));
      }
      // give the busy kernel a moment to occupy its slots
      { auto t = std::chrono::steady_clock::now(); while (std::chrono::steady_clock::now() - t < std::chrono::microseconds(200)) {} }
      CK(hipEventRecord(e0, schain));
      for (int i = 0; i < chain_len; i++) hipLaunchKernelGGL(k_short, dim3(chain_waves), dim3(64), 0, schain, d_out, 400);
      CK(hipEventRecord(e1, schain));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&chain_ms, e0, e1));
      if (with_busy) CK(hipEventElapsedTime(&busy_ms, b0, b1));
    }
    printf("%-58s chain of %d x %4d-wave launches: %8.1f us (%.1f us per launch)%s", label, chain_len, chain_waves, chain_ms * 1e3, chain_ms * 1e3 / chain_len,
           with_busy ? "" : "\n");
    if (with_busy) printf("   busy kernel %.3f ms\n", busy_ms);
  };
  {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
    for (int cw : {16, 64, 512}) {
      time_chain(a, b, ncu, false, cw, "unmasked, chain alone");
      time_chain(a, b, ncu, true, cw, "unmasked, chain (high priority) beside busy on all CUs");
      time_chain(a, b, ncu - ncu / 16, true, cw, "unmasked, busy grid leaves 1/16 of the wave slots free");
    }
    CK(hipStreamDestroy(a));
    CK(hipStreamDestroy(b));
  }
  for (int r : {1, 2, 4, 8}) {
    std::vector<uint32_t> tail = low_bits(8 * r), mainm = complement(tail);
    hipStream_t a = masked_stream(mainm), b = masked_stream(tail);
    char lab[128];
    for (int cw : {16, 64, 512}) {
      snprintf(lab, sizeof lab, "masked %d CUs/XCD for the chain, chain alone", r);
      time_chain(a, b, ncu - 8 * r, false, cw, lab);
      snprintf(lab, sizeof lab, "masked %d CUs/XCD for the chain, busy on the other %d CUs", r, ncu - 8 * r);
      time_chain(a, b, ncu - 8 * r, true, cw, lab);
    }
    CK(hipStreamDestroy(a));
    CK(hipStreamDestroy(b));
  }
  // ---- (3) two chains of dependent short launches on two streams at once (the main-stream chain of a small MSM beside the previous
  // MSM's tail chain): does a second active hardware queue delay dependent launches?  k dummy streams are created between the two
  // (hardware queues are handed out in creation order; queues that share a pipe of the command processor take turns)
  for (int dummies : {0, 1, 2, 3, 5}) {
    hipStream_t a, b, d[8];
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    for (int i = 0; i < dummies; i++) {
      CK(hipStreamCreateWithFlags(&d[i], hipStreamNonBlocking));
      hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, d[i], d_out + (4u << 20), 1);   // (a stream gets its queue with its first work)
    }
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const int len = 40;
    float ta = 0, tb = 0, alone = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a0, a));
      for (int i = 0; i < len; i++) hipLaunchKernelGGL(k_short, dim3(64), dim3(64), 0, a, d_out, 400);
      CK(hipEventRecord(a1, a));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&alone, a0, a1));
      CK(hipEventRecord(a0, a));
      CK(hipEventRecord(b0, b));
      for (int i = 0; i < len; i++) {
        hipLaunchKernelGGL(k_short, dim3(64), dim3(64), 0, a, d_out, 400);
        hipLaunchKernelGGL(k_short, dim3(64), dim3(64), 0, b, d_out + (2u << 20), 400);
      }
      CK(hipEventRecord(a1, a));
      CK(hipEventRecord(b1, b));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ta, a0, a1));
      CK(hipEventElapsedTime(&tb, b0, b1));
    }
    printf("two chains of %d dependent 64-wave launches, %d dummy streams between: alone %.1f us per launch; together %.1f / %.1f us per launch\n", len, dummies,
           alone * 1e3 / len, ta * 1e3 / len, tb * 1e3 / len);
    CK(hipStreamDestroy(a));
    CK(hipStreamDestroy(b));
    for (int i = 0; i < dummies; i++) CK(hipStreamDestroy(d[i]));
  }
  // ---- (4) the shape seen in the MSM timelines: the main stream's chain of SHORT kernels with wide workgroups (count / scan / scatter of the
  // sort) beside the tail stream's chain of ~9 us kernels with many one-wave workgroups (narrow reduction passes), the latter on a
  // high-priority stream; k dummy streams between the two
  for (int dummies : {0, 1, 2, 3, 4, 5, 6}) {
    hipStream_t a, b, d[8];
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    for (int i = 0; i < dummies; i++) {
      CK(hipStreamCreateWithFlags(&d[i], hipStreamNonBlocking));
      hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, d[i], d_out + (4u << 20), 1);
    }
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
    CK(hipDeviceSynchronize());
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const int len = 40;
    float ta = 0, tb = 0, alone = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a0, a));
      for (int i = 0; i < len; i++) hipLaunchKernelGGL(k_tiny_wide, dim3(20), dim3(1024), 0, a, d_out, 200);
      CK(hipEventRecord(a1, a));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&alone, a0, a1));
      CK(hipEventRecord(b0, b));
      for (int i = 0; i < len; i++) hipLaunchKernelGGL(k_short, dim3(1280), dim3(64), 0, b, d_out + (2u << 20), 500);
      CK(hipEventRecord(b1, b));
      CK(hipEventRecord(a0, a));
      for (int i = 0; i < len; i++) hipLaunchKernelGGL(k_tiny_wide, dim3(20), dim3(1024), 0, a, d_out, 200);
      CK(hipEventRecord(a1, a));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ta, a0, a1));
      CK(hipEventElapsedTime(&tb, b0, b1));
    }
    printf("short wide-workgroup chain beside a high-priority chain of 1280-wave launches, %d dummy streams between: alone %.1f us per launch; beside %.1f us per launch (the other chain %.1f)\n",
           dummies, alone * 1e3 / len, ta * 1e3 / len, tb * 1e3 / len);
    CK(hipStreamDestroy(a));
    CK(hipStreamDestroy(b));
    for (int i = 0; i < dummies; i++) CK(hipStreamDestroy(d[i]));
  }
  // the busy kernel alone, masked and unmasked (what the reserved CUs cost it): same work per wave, fewer waves
  {
    hipStream_t a;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    hipEvent_t b0, b1;
    CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    for (int r : {0, 1, 2, 4, 8}) {
      hipStream_t s = r ? masked_stream(complement(low_bits(8 * r))) : a;
      const int cus = ncu - 8 * r;
      // the same TOTAL work spread over the CUs the stream has: iterations scale with 256 / cus
      const int iters = (int)(60000.0 * ncu / cus);
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(b0, s));
        hipLaunchKernelGGL(k_busy, dim3(cus * 8), dim3(64), 0, s, d_out + (8u << 20), iters);
        CK(hipEventRecord(b1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, b0, b1));
      }
      printf("busy kernel, same total work on %3d CUs (%d reserved per XCD): %.3f ms\n", cus, r, ms);
      if (r) CK(hipStreamDestroy(s));
    }
  }
  return 0;
}
