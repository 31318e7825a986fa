// tools/h2d_overlap.hip -- how host-to-device copies behave on this box while a kernel holds every wave slot
// (the situation of a host-pointer MSM that uploads the next slice of points underneath the accumulate kernel):
// pageable vs pinned source, hipHostRegister cost, threaded staging through a pinned bounce buffer.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -pthread tools/h2d_overlap.hip -o tools/h2d_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one-wave workgroups, ~200 VGPRs' worth of dependent integer work: takes every wave slot like k_accum does
__global__ void __launch_bounds__(64, 2) k_busy(uint32_t* out, int iters) {
  uint64_t a[48];
  uint32_t x = blockIdx.x * 64 + threadIdx.x;
  for (int i = 0; i < 48; i++) a[i] = x + i;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 48; i++) a[i] = a[i] * (uint64_t)(x | 1) + a[(i + 1) % 48];
  uint64_t s = 0;
  for (int i = 0; i < 48; i++) s ^= a[i];
  out[x] = (uint32_t)s;
}

int main() {
  const size_t MB = 1 << 20, bytes = 128 * MB;
  char* pageable = (char*)malloc(bytes);
  memset(pageable, 1, bytes);
  char* pinned;
  CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
  memset(pinned, 2, bytes);
  char* dev;
  CK(hipMalloc((void**)&dev, bytes));
  uint32_t* out;
  const int nblk = 256 * 8;
  CK(hipMalloc((void**)&out, nblk * 64 * 4));
  hipStream_t sc, sk;
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  // calibrate the busy kernel to ~5 ms
  int iters = 2000;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0, sk));
    hipLaunchKernelGGL(k_busy, dim3(nblk), dim3(64), 0, sk, out, iters);
    CK(hipEventRecord(e1, sk));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    iters = (int)(iters * 5.0 / ms) + 1;
  }
  auto busy_ms = [&]() {
    CK(hipEventRecord(e0, sk));
    hipLaunchKernelGGL(k_busy, dim3(nblk), dim3(64), 0, sk, out, iters);
    CK(hipEventRecord(e1, sk));
  };
  auto timed_copy = [&](const char* src, size_t n, const char* label, bool with_kernel) {
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      if (with_kernel) busy_ms();
      double t0 = now();
      CK(hipMemcpyAsync(dev, src, n, hipMemcpyHostToDevice, sc));
      double t1 = now();
      CK(hipStreamSynchronize(sc));
      double t2 = now();
      float kms = 0;
      if (with_kernel) {
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&kms, e0, e1));
      }
      if (rep == 2)
        printf("{\"case\": \"%s\", \"MiB\": %zu, \"kernel_running\": %d, \"call_returns_ms\": %.3f, \"copy_done_ms\": %.3f, \"GBps\": %.1f, \"kernel_ms\": %.3f}\n",
               label, n / MB, (int)with_kernel, (t1 - t0) * 1e3, (t2 - t0) * 1e3, n / (t2 - t0) / 1e9, kms);
    }
  };
  for (size_t n : {32 * MB, 128 * MB}) {
    timed_copy(pageable, n, "pageable", false);
    timed_copy(pageable, n, "pageable", true);
    timed_copy(pinned, n, "pinned(hipHostMalloc)", false);
    timed_copy(pinned, n, "pinned(hipHostMalloc)", true);
  }
  // hipHostRegister of a fresh pageable buffer
  for (size_t n : {32 * MB, 128 * MB}) {
    char* p = (char*)malloc(n);
    memset(p, 3, n);
    double t0 = now();
    CK(hipHostRegister(p, n, hipHostRegisterDefault));
    double t1 = now();
    CK(hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, sc));
    CK(hipStreamSynchronize(sc));
    double t2 = now();
    CK(hipHostUnregister(p));
    double t3 = now();
    printf("{\"case\": \"hipHostRegister\", \"MiB\": %zu, \"register_ms\": %.3f, \"copy_ms\": %.3f, \"unregister_ms\": %.3f}\n", n / MB,
           (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    free(p);
  }
  // threaded staging: T host threads copy pageable -> pinned bounce buffer in 4 MiB pieces, each piece DMA'd as soon as it is staged
  for (int T : {1, 2, 4, 8}) {
    for (int with_kernel = 0; with_kernel < 2; with_kernel++) {
      CK(hipDeviceSynchronize());
      if (with_kernel) busy_ms();
      const size_t piece = 4 * MB, np = bytes / piece;
      double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
          for (size_t i = t; i < np; i += T) memcpy(pinned + i * piece, pageable + i * piece, piece);
        });
      for (auto& x : th) x.join();
      double t1 = now();
      CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, sc));
      CK(hipStreamSynchronize(sc));
      double t2 = now();
      float kms = 0;
      if (with_kernel) {
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&kms, e0, e1));
      }
      printf("{\"case\": \"stage through pinned, %d threads\", \"MiB\": 128, \"kernel_running\": %d, \"memcpy_ms\": %.3f, \"memcpy_GBps\": %.1f, \"dma_ms\": %.3f, \"kernel_ms\": %.3f}\n",
             T, with_kernel, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, kms);
    }
  }
  return 0;
}
