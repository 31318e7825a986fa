// tools/microbench_inv.hip -- (A) cost of one field inversion on a lane: division steps (modinv.h) against Fermat's a^(p-2),
// 381- and 254-bit moduli; (B) the rate of BATCHED AFFINE additions built on it -- the reference's bucket accumulation for
// c >= 9 (ec_multi_scalar_mul_scheduler.nim:414-553, adds ec_shortweierstrass_batch_ops.nim:424-455): every lane owns M
// independent pairs (P_i, Q_i), one Montgomery-trick inversion per lane and round, 5M + 1S per addition
// (lambda = dy * (1/dx), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1), in the carry-free device field.
// The figure to beat is the XYZZ mixed addition of k_accum: 16.7 M additions in 2.38 ms = 7.0 G additions/s (BLS12-381 G1).
// The pair round reads its operands from HBM and writes its sums back (a tree over the sorted entries has to: a lane cannot
// hold M points in registers), which is the other half of the question.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I constantine_amd/csrc tools/microbench_inv.hip -o tools/microbench_inv.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ec.h"
using namespace ctt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- (A) inversion chains -------------------------------------------------------------------------------------------
template <class PP, bool FERMAT>
__global__ void __launch_bounds__(64) k_inv_chain(const uint32_t* in, uint32_t* out, int reps) {
  using F = Fp<PP>;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  F x;
#pragma unroll
  for (int i = 0; i < F::N; i++) x.l[i] = in[(size_t)j * F::N + i];
  const F one = F::one();
  for (int r = 0; r < reps; r++) {
    F y;
    if constexpr (FERMAT) y = F::inv_fermat(x); else y = F::inv(x);
    x = F::add(y, one);   // dependent chain, a fresh operand every time
  }
#pragma unroll
  for (int i = 0; i < F::N; i++) out[(size_t)j * F::N + i] = x.l[i];
}
// division-step batches one inversion takes: max and mean over the lanes (host side, same code)
template <class PP>
static void batch_stats(const std::vector<uint32_t>& vals, size_t n, double* mean, int* mx) {
  using MI = ModInv<PP>;
  double sum = 0;
  int m = 0;
  for (size_t j = 0; j < n; j++) {
    typename MI::S30 f = MI::modulus(), g, d, e;
    for (int i = 0; i < MI::L; i++) { g.v[i] = MI::limb30(&vals[j * PP::N], i); d.v[i] = 0; e.v[i] = i == 0; }
    int32_t zeta = -1;
    int it = 0;
    for (;; ) {
      int32_t t[4];
      zeta = MI::divsteps30(zeta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), t);
      MI::update_de(d, e, t);
      MI::update_fg(f, g, t);
      it++;
      int32_t nz = 0;
      for (int i = 0; i < MI::L; i++) nz |= g.v[i];
      if (!nz || it >= MI::MAX_BATCHES) break;
    }
    sum += it;
    if (it > m) m = it;
  }
  *mean = sum / (double)n;
  *mx = m;
}

template <class PP, bool FERMAT>
static void bench_inv(const char* field, int waves_per_simd, int reps) {
  int dev = 0, ncu = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t lanes = (size_t)ncu * 4 * 64 * waves_per_simd;
  constexpr int N = PP::N;
  std::vector<uint32_t> h(lanes * N);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < h.size(); i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (uint32_t)(s >> 32); }
  for (size_t j = 0; j < lanes; j++) h[j * N + N - 1] &= (PP::P[N - 1] >> 1);   // below p
  uint32_t *d_in, *d_out;
  CK(hipMalloc(&d_in, h.size() * 4));
  CK(hipMalloc(&d_out, h.size() * 4));
  CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_inv_chain<PP, FERMAT>), dim3(lanes / 64), dim3(64), 0, 0, d_in, d_out, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_inv_chain<PP, FERMAT>), dim3(lanes / 64), dim3(64), 0, 0, d_in, d_out, reps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double mean = 0;
  int mx = 0;
  if (!FERMAT) batch_stats<PP>(h, lanes < 4096 ? lanes : 4096, &mean, &mx);
  printf("{\"bench\": \"inversion\", \"field\": \"%s\", \"algo\": \"%s\", \"waves_per_simd\": %d, \"lanes\": %zu, \"reps\": %d, \"ms\": %.3f, "
         "\"us_per_inversion_per_wave\": %.2f, \"G_inversions_per_s\": %.4f, \"divstep_batches_mean\": %.2f, \"divstep_batches_max\": %d}\n",
         field, FERMAT ? "fermat" : "divsteps30", waves_per_simd, lanes, reps, ms, ms * 1e3 / reps, (double)lanes * reps / ms / 1e6, mean, mx);
  fflush(stdout);
  CK(hipFree(d_in));
  CK(hipFree(d_out));
}

// ---- (B) one round of batched affine additions, BLS12-381 G1, carry-free field ----------------------------------------
using UP = BLS12_381_Fp_U;
using FU = FpU<UP>;
using PPs = BLS12_381_Fp;
constexpr int NL = UP::NL, LB = UP::LB;
struct Rec { FU x, y; uint32_t pad[4]; };   // 128 bytes, the engine's point record
static_assert(sizeof(Rec) == 128, "record");

// FpU value (x R', < 4p, limbs normalised) -> canonical 32-bit words of x R' mod p
__device__ __forceinline__ void fu_to_words(const FU& a, uint32_t* w) {
  constexpr int N = PPs::N;
  uint32_t v[N + 1];
#pragma unroll
  for (int i = 0; i <= N; i++) v[i] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int pos = LB * i, word = pos >> 5, sh = pos & 31;
    const uint64_t x = (uint64_t)a.l[i] << sh;
    uint64_t s = (uint64_t)v[word] + (uint32_t)x;
    v[word] = (uint32_t)s;
    s = (s >> 32) + (uint64_t)v[word + 1] + (uint32_t)(x >> 32);
    v[word + 1] = (uint32_t)s;
  }
  for (int k = 0; k < 3; k++) {   // < 4p: at most three subtractions
    uint32_t d[N + 1];
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i <= N; i++) {
      const uint64_t s = (uint64_t)v[i] - (i < N ? PPs::P[i] : 0u) - bw;
      d[i] = (uint32_t)s;
      bw = (s >> 32) & 1u;
    }
    const uint32_t keep = 0u - (uint32_t)bw;   // borrow: v < p, keep v
#pragma unroll
    for (int i = 0; i <= N; i++) v[i] = (v[i] & keep) | (d[i] & ~keep);
  }
#pragma unroll
  for (int i = 0; i < N; i++) w[i] = v[i];
}
__device__ __forceinline__ FU words_to_fu(const uint32_t* w) {
  FU r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = FU::bits_at(w, PPs::N, LB * i);
  return r;
}
// 1/a in the carry-free Montgomery domain: a = x R' -> x^-1 R' = (x R')^-1 R'^2 = mont'((x R')^-1, R'^3)
__device__ __forceinline__ FU fu_inv(const FU& a, const FU& rp3) {
  uint32_t w[PPs::N], o[PPs::N];
  fu_to_words(a, w);
  ModInv<PPs>::inv_words(w, o);
  return FU::mul(words_to_fu(o), rp3);
}

// lane l adds the pairs (in[2 (l M + i)], in[2 (l M + i) + 1]), i < M, into out[l M + i]; prefix: M running products per lane
template <int M>
__global__ void __launch_bounds__(64, 2) k_affine_round(const Rec* in, Rec* out, FU* prefix, FU rp3, uint32_t npairs, uint32_t* bad) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t p0 = (uint64_t)lane * M;
  if (p0 >= npairs) return;
  FU run = FU::one();
  const uint32_t nlanes = (npairs + M - 1) / M;
  FU* pre = prefix + lane;             // lane-interleaved: the wave's 64 running products of step i are contiguous
#pragma unroll 1
  for (int i = 0; i < M; i++) {
    const FU x1 = in[2 * (p0 + i)].x, x2 = in[2 * (p0 + i) + 1].x;
    const FU dx = FU::template sub_lazy<2>(x2, x1);
    pre[(uint64_t)i * nlanes] = run;   // product of the dx before this pair
    run = FU::mul(run, dx);
  }
  FU inv = fu_inv(run, rp3);
  {   // self-check of the inversion: run * inv == 1
    const FU chk = FU::mul(run, inv);
    uint32_t w[PPs::N];
    fu_to_words(chk, w);
    uint32_t w1[PPs::N];
    fu_to_words(FU::one(), w1);
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < PPs::N; i++) d |= w[i] ^ w1[i];
    if (d) atomicAdd(bad, 1u);
  }
#pragma unroll 1
  for (int i = M - 1; i >= 0; i--) {
    const Rec a = in[2 * (p0 + i)], b = in[2 * (p0 + i) + 1];
    const FU dx = FU::template sub_lazy<2>(b.x, a.x);
    const FU dy = FU::template sub_lazy<2>(b.y, a.y);
    const FU idx = FU::mul(inv, pre[(uint64_t)i * nlanes]);   // 1 / dx_i
    inv = FU::mul(inv, dx);
    const FU lam = FU::mul(dy, idx);
    const FU l2 = FU::sqr(lam);
    Rec r;
    r.x = FU::template sub<5>(FU::template sub<3>(l2, a.x), b.x);          // < 2 + 3 + 5
    const FU t = FU::template sub_lazy<10>(a.x, r.x);
    r.y = FU::template sub<3>(FU::mul(lam, t), a.y);
    r.pad[0] = r.pad[1] = r.pad[2] = r.pad[3] = 0;
    out[p0 + i] = r;
  }
}

template <int M>
static void bench_round(uint32_t npairs, const FU& rp3, const Rec* d_in, Rec* d_out, FU* d_prefix, uint32_t* d_bad) {
  const uint32_t lanes = (npairs + M - 1) / M;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_affine_round<M>, dim3((lanes + 63) / 64), dim3(64), 0, 0, d_in, d_out, d_prefix, rp3, npairs, d_bad);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  uint32_t bad = 0;
  CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("{\"bench\": \"batched_affine_round\", \"curve\": \"bls12_381_g1\", \"pairs_per_lane\": %d, \"pairs\": %u, \"lanes\": %u, \"ms\": %.3f, "
         "\"G_additions_per_s\": %.3f, \"inversion_self_check_failures\": %u, \"k_accum_G_additions_per_s\": 7.0}\n",
         M, npairs, lanes, best, npairs / best / 1e6, bad);
  fflush(stdout);
}

// R'^3 mod p in the carry-free limbs, by 3 * LB * NL modular doublings of 1 on the host
static FU host_rp3() {
  constexpr int N = PPs::N;
  uint32_t v[N + 1] = {1};
  for (int step = 0; step < 3 * LB * NL; step++) {
    uint32_t c = 0;
    for (int i = 0; i <= N; i++) { const uint32_t nc = v[i] >> 31; v[i] = (v[i] << 1) | c; c = nc; }
    uint32_t d[N + 1];
    uint64_t bw = 0;
    for (int i = 0; i <= N; i++) { const uint64_t s = (uint64_t)v[i] - (i < N ? PPs::P[i] : 0u) - bw; d[i] = (uint32_t)s; bw = (s >> 32) & 1u; }
    if (!bw) memcpy(v, d, sizeof(d));
  }
  FU r;
  for (int i = 0; i < NL; i++) r.l[i] = FU::bits_at(v, N, LB * i);
  return r;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 8;
  for (int w : {1, 2, 4}) {
    bench_inv<BLS12_381_Fp, false>("bls12_381_fp", w, reps);
    bench_inv<BN254_Fp, false>("bn254_fp", w, reps);
  }
  bench_inv<BLS12_381_Fp, true>("bls12_381_fp", 2, 2);
  bench_inv<BN254_Fp, true>("bn254_fp", 2, 2);

  const uint32_t npairs = 1u << 23;
  std::vector<uint32_t> h((size_t)npairs * 2 * 32);
  uint64_t s = 12345;
  for (size_t i = 0; i < h.size(); i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (uint32_t)(s >> 36) & UP::MASK; }
  for (size_t r = 0; r < (size_t)npairs * 2; r++) { h[r * 32 + NL - 1] &= 0xffffu; h[r * 32 + 2 * NL - 1] &= 0xffffu; }   // values below p (its top limb has 17 bits)
  Rec *d_in, *d_out;
  FU* d_prefix;
  uint32_t* d_bad;
  CK(hipMalloc(&d_in, (size_t)npairs * 2 * sizeof(Rec)));
  CK(hipMalloc(&d_out, (size_t)npairs * sizeof(Rec)));
  CK(hipMalloc(&d_prefix, (size_t)npairs * sizeof(FU)));
  CK(hipMalloc(&d_bad, 4));
  CK(hipMemset(d_bad, 0, 4));
  CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const FU rp3 = host_rp3();
  bench_round<8>(npairs, rp3, d_in, d_out, d_prefix, d_bad);
  bench_round<16>(npairs, rp3, d_in, d_out, d_prefix, d_bad);
  bench_round<32>(npairs, rp3, d_in, d_out, d_prefix, d_bad);
  bench_round<64>(npairs, rp3, d_in, d_out, d_prefix, d_bad);
  bench_round<128>(npairs, rp3, d_in, d_out, d_prefix, d_bad);
  return 0;
}
