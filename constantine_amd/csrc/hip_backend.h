// hip_backend.h -- HIP launch layer shared by the per-curve translation units (curve_*.hip) and the
// engine (msm_engine.hip): stream, stage events, EC kernel templates, per-curve operation table.
#pragma once
#include <hip/hip_runtime.h>

#include "generators.h"
#include "hip_errors.h"
#include "msm_pipeline.h"

namespace ctt {

// a failed HIP call is reported through the entry point's return value where there is one, and aborts where there is none (hip_errors.h)
#define HIP_CHECK(expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      ::ctt::hip_failed(#expr, hipGetErrorString(e_), e_ == hipErrorOutOfMemory, __FILE__, __LINE__); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Kernel templates (instantiated per curve in curve_*.hip)
// ---------------------------------------------------------------------------------------------
template <class Fr>
__global__ void k_fr_from_mont(const uint32_t* in, uint32_t* out, uint32_t n) {
  fr_from_mont_body<Fr>(in, out, n, blockIdx.x * blockDim.x + threadIdx.x);
}

// --- EC kernels: one lane = one body invocation ----------------------------------------------------
static constexpr int ACCUM_BLOCK = 64;

static constexpr int EC_BLOCK = 64;

// Input points -> records (msm_bodies.h convert_point_body defines the record).  A lane converts its point into a record
// held in LDS and the workgroup then streams the records out with consecutive 16-byte stores: written straight from the
// lanes, every store instruction of a wave touched 64 different lines, 16 bytes each (the record stride is a whole line),
// and the L2 had to merge 8 such partial writes per line.  The 16-byte chunks of a record are XOR-swizzled with the record
// number so that both the lane-private writes and the linear read-out spread over all LDS banks.
static constexpr int CONVERT_BLOCK = 128;
template <class F, class FD>
__global__ void __launch_bounds__(CONVERT_BLOCK) k_convert_points(const Affine<F>* in, void* out, uint32_t n) {
  constexpr uint32_t STRIDE = gather_stride<FD>(), CH = STRIDE / 16u;   // chunks per record (8, 16, ...: a power of two)
  extern __shared__ uint4 cv_lds[];
  const uint32_t t = threadIdx.x;
  const uint32_t j0 = blockIdx.x * CONVERT_BLOCK, j = j0 + t;
  union Rec {
    uint4 q[CH];
    struct { Affine<FD> a; } v;
    uint32_t w[STRIDE / 4u];
  };
  if (j < n) {
    Rec r;
#pragma unroll
    for (uint32_t i = 0; i < CH; i++) r.q[i] = make_uint4(0u, 0u, 0u, 0u);
    const Affine<F> p = in[j];
    r.v.a.x = FD::from_sat(p.x);
    r.v.a.y = FD::from_sat(p.y);
    r.w[gather_flag_offset<FD>() / 4u] = p.is_inf() ? 1u : 0u;
#pragma unroll
    for (uint32_t i = 0; i < CH; i++) cv_lds[t * CH + (i ^ (t & (CH - 1u)))] = r.q[i];
  }
  __syncthreads();
  const uint32_t cnt = (n - j0 < (uint32_t)CONVERT_BLOCK ? n - j0 : (uint32_t)CONVERT_BLOCK) * CH;
  uint4* dst = reinterpret_cast<uint4*>((char*)out + (uint64_t)j0 * STRIDE);
  for (uint32_t i = t; i < cnt; i += CONVERT_BLOCK) {
    const uint32_t rec = i / CH, ch = i & (CH - 1u);
    dst[i] = cv_lds[rec * CH + (ch ^ (rec & (CH - 1u)))];
  }
}

#ifndef CTT_ACCUM_WAVES
#define CTT_ACCUM_WAVES 2   // waves per SIMD the accumulate kernel is compiled for (register budget 512/waves)
#endif
// an XYZZ accumulator over Fp2 (448 bytes) plus a point and the temporaries of the addition does not fit two waves per SIMD
// (parking ZZ, ZZZ in LDS to get there was measured and rejected: DESIGN.md section 5)
// The gather of the accumulate loop, device form (msm_bodies.h GatherDirect for the contract): request() has the record of
// the NEXT entry written into LDS by global_load_lds_dwordx4 while the current addition runs -- the data never waits in
// registers (a register-staged request costs 29 / 57 VGPRs: 242 of 256 for BLS12-381 G1, spills for the 9-limb fields at four
// waves per SIMD and for G2 at 512).  One workgroup = one wave: chunk j of lane l lies at stage[j][l], which is where the
// instruction puts it (LDS address = M0 + lane * 16).  collect() waits for everything the lane has in flight (vmcnt counts the
// LDS-bound loads too) and reads the chunks back.
template <class F>
struct GatherLds {
  static constexpr uint32_t NCH = gather_chunks<F>();
  static_assert(F::UNSAT, "records in the converted layout only");
  uint4 (*stage)[ACCUM_BLOCK];
  __device__ void request(const char* rec) {
    // the previous record has been read out of the staging lines (collect's ds_reads have returned) before they are overwritten
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (uint32_t j = 0; j < NCH; j++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + 16u * j),
                                       (__attribute__((address_space(3))) void*)&stage[j][0], 16, 0, 0);
  }
  __device__ void collect(Affine<F>& pt, bool& qinf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    union {
      uint4 q[NCH];
      uint32_t w[NCH * 4u];
    } r;
#pragma unroll
    for (uint32_t j = 0; j < NCH; j++) r.q[j] = stage[j][threadIdx.x];
    __builtin_memcpy(&pt, r.w, sizeof(Affine<F>));
    qinf = r.w[gather_flag_offset<F>() / 4u] != 0u;
  }
};
template <class F, bool INTO = false>
__global__ void __launch_bounds__(ACCUM_BLOCK, (sizeof(XYZZ<F>) > 256 ? 1 : CTT_ACCUM_WAVES)) k_accum(AccumArgs<F> a) {
#if defined(CTT_G2_PHASE_NOPS) && CTT_G2_PHASE_NOPS > 0
  // Shifts the BLS12-381 G2 kernel's instruction stream by 4 bytes per s_nop: its speed depends on the 8-byte phase of its multiply-adds
  // (msm_bodies.h accum_body_z; tools/phase_stats.py says which phase a build has).  Not needed for the shipped sources (73 % at 0 mod 8).
  if constexpr (IsFp2<F>::value && F::UNSAT) {
#pragma unroll
    for (int i = 0; i < CTT_G2_PHASE_NOPS; i++) asm volatile("s_nop 0");
  }
#endif
  if constexpr (F::UNSAT && !IsFp2<F>::value) {   // (not the quadratic extension: msm_bodies.h accum_body_z has the measurement)
    __shared__ uint4 stage[GatherLds<F>::NCH][ACCUM_BLOCK];
    GatherLds<F> gq{stage};
    accum_body<F, GatherLds<F>, INTO>(a, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, gq);
  } else {
    accum_body<F, INTO>(a, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
  }
}
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_merge_tail(MergeArgs<F> a) {
  merge_tail_body<F>(a, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
}
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_merge_step(MergeArgs<F> a, uint32_t d) {
  merge_step_body<F>(a, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, d);
}
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_merge_final(MergeArgs<F> a) {
  if (merge_chain_bound<F>(a) <= 1) return;   // the tail merge wrote the buckets
  merge_final_body<F>(a, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
}
// queue form of the head merge (msm_bodies.h merge_tail_queue_body): the tail merge; the first heads of the chains with work left
// go to the queue with ONE atomic per wave (ballot + prefix count; a workgroup is one wave)
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_merge_tail_queue(MergeArgs<F> a) {
  uint32_t item = 0;
  const bool more = merge_tail_queue_body<F>(a, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, &item);
  const uint64_t mask = __ballot(more);
  if (mask == 0) return;
  uint32_t base = 0;
  const uint32_t lane = threadIdx.x & 63u;
  if (lane == (uint32_t)(__ffsll((long long)mask) - 1)) base = atomicAdd(a.qcount, (uint32_t)__popcll(mask));
  base = __shfl(base, __ffsll((long long)mask) - 1, 64);
  if (more) a.queue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = item;
}
// one lane per queued chain; the grid covers the queue's capacity, the lanes beyond its count leave at once
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_merge_queue(MergeArgs<F> a, uint32_t lmax) {
  merge_queue_body<F>(a, blockIdx.x * blockDim.x + threadIdx.x, lmax);
}
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_pyr(PyrArgs<F> a, uint32_t ntasks) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ntasks) pyr_body<F>(a, blockIdx.y, t);
}
static __global__ void k_iota(uint32_t* entries, uint32_t n, uint32_t* bucket_start, uint32_t* maxcount) {
  iota_body(entries, n, bucket_start, maxcount, blockIdx.x * blockDim.x + threadIdx.x);
}
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_batch_affine(BatchAffineArgs<F> a) {
  batch_affine_body<F>(a, blockIdx.x * blockDim.x + threadIdx.x);
}
// --- four lanes per addition -------------------------------------------------------------------------------------
// The late passes of the bucket reduction (and the head-merge steps) have only a few hundred additions per window
// left: one lane per addition leaves the chip idle while every lane walks 14 dependent field products (~18 us).
// Here a quad of lanes shares one addition: the 14 products of the XYZZ sum are arranged in 4 rounds of (up to) 4
// independent products, one per lane, with quad broadcasts in between -- 4 products deep instead of 14.
//   round 1   U1 = X1*ZZ2      U2 = X2*ZZ1      S1 = Y1*ZZZ2      S2 = Y2*ZZZ1
//   round 2   PP = P^2         RR = R^2         Z2 = ZZ1*ZZ2      Z3 = ZZZ1*ZZZ2        (P = U2-U1, R = S2-S1)
//   round 3   PPP = P*PP       Q = U1*PP        ZZ3 = Z2*PP       -
//   round 4   A = R*(Q-X3)     Bv = S1*PPP      -                 ZZZ3 = Z3*PPP         (X3 = RR-PPP-2Q, Y3 = A-Bv)
// Exceptional inputs (a neutral operand, P = +-Q) are rare here: lane 0 of the quad then runs the ordinary addition.
// value of lane ROLE of the quad, in every lane of the quad: one v_mov_b32 with a DPP quad_perm per limb (full rate; the
// first version went through ds_bpermute_b32, an LDS-crossbar instruction per limb)
template <int ROLE, class F>
__device__ __forceinline__ F quad_bcast(const F& v) {
  F r;
  if constexpr (IsFp2<F>::value) {
    r.c0 = quad_bcast<ROLE>(v.c0);
    r.c1 = quad_bcast<ROLE>(v.c1);
  } else {
    constexpr int ctrl = ROLE | (ROLE << 2) | (ROLE << 4) | (ROLE << 6);   // quad_perm:[ROLE,ROLE,ROLE,ROLE]
#pragma unroll
    for (int i = 0; i < (int)(sizeof(F) / 4); i++)
      // `old` = the value itself (every lane of a quad is active whenever this runs): with a constant `old` hipcc's DPP
      // combine folds the move into a neighbouring subtraction and gets the operand order of a - dpp(b) wrong when both
      // are the same register (y came out negated in the window-sum kernel: tools/quad_probe.hip)
      r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)v.l[i], (int)v.l[i], ctrl, 0xf, 0xf, false);
  }
  return r;
}

template <class F>
__device__ __forceinline__ void xyzz_add_quad(const XYZZ<F>* s1, const XYZZ<F>* s2, XYZZ<F>* d1, XYZZ<F>* d2, int role) {
  constexpr int M = F::MULB;
  const F* A = &s1->x;  // x, y, zz, zzz
  const F* Q = &s2->x;
  const bool ainf = A[2].is_zero(), qinf = Q[2].is_zero();
  bool plain = ainf | qinf;
  F U1, U2, S1, S2, P, R;
  if (!plain) {
    const int hi = role >> 1;
    const bool fromq = (role & 1) != 0;
    const F opA = fromq ? Q[hi] : A[hi];          // a.x | q.x | a.y | q.y
    const F opB = fromq ? A[2 + hi] : Q[2 + hi];  // q.zz | a.zz | q.zzz | a.zzz
    const F T1 = F::mul(opA, opB);
    U1 = quad_bcast<0>(T1);
    U2 = quad_bcast<1>(T1);
    S1 = quad_bcast<2>(T1);
    S2 = quad_bcast<3>(T1);
    P = fsub<F, M>(U2, U1);  // < 2M
    R = fsub<F, M>(S2, S1);  // < 2M
    plain = fis_zero_modp<F, 2 * M>(P);
  }
  if (plain) {  // uniform inside the quad
    if (role == 0) {
      const XYZZ<F> x = *s1, y = *s2;
      const XYZZ<F> r = xyzz_add_inl<F>(x, y);
      *d1 = r;
      if (d2) *d2 = r;
    }
    return;
  }
  const int zi = role >= 2 ? role : 2;
  const F za = A[zi], zq = Q[zi];
  const F PR = F::select(role == 0, P, R);
  const F T2 = F::mul(F::select(role < 2, PR, za), F::select(role < 2, PR, zq));  // PP | RR | Z2 | Z3
  const F PP = quad_bcast<0>(T2);
  const F RR = quad_bcast<1>(T2);
  const F T3 = F::mul(F::select(role == 0, P, F::select(role == 1, U1, T2)), PP);  // PPP | Q | ZZ3 | (unused)
  const F PPP = quad_bcast<0>(T3);
  const F Qv = quad_bcast<1>(T3);
  const F X3 = fsub<F, 2 * M>(fsub<F, M>(RR, PPP), F::dbl(Qv));  // < 4M
  const F T4 = F::mul(F::select(role == 0, R, F::select(role == 1, S1, T2)),
                      F::select(role == 0, fsub<F, 4 * M>(Qv, X3), PPP));        // A | Bv | (unused) | ZZZ3
  const F Bv = quad_bcast<1>(T4);
  if (role == 0) {
    const F Y3 = fsub<F, M>(T4, Bv);  // < 2M
    d1->x = X3;
    d1->y = Y3;
    if (d2) { d2->x = X3; d2->y = Y3; }
  } else if (role == 2) {
    d1->zz = T3;
    if (d2) d2->zz = T3;
  } else if (role == 3) {
    d1->zzz = T4;
    if (d2) d2->zzz = T4;
  }
}

// --- the accumulator of a quad resident in registers (every lane holds a full copy): the bit Horner of k_reduce_finish ---
template <class F>
__device__ __forceinline__ F quad_pick(int role, const F& a, const F& b, const F& c, const F& d) {
  return F::select(role < 2, F::select(role == 0, a, b), F::select(role == 2, c, d));
}
template <class F>
__device__ __forceinline__ void xyzz_dbl_quad_reg(XYZZ<F>& p, int role) {
  constexpr int M = F::MULB;
  if (p.is_inf()) return;  // uniform inside the quad
  const F U = F::dbl(p.y);                                            // < 4M
  const F T1 = F::sqr(F::select(role == 0, U, p.x));                  // V | XX | - | -
  const F V = quad_bcast<0>(T1), XX = quad_bcast<1>(T1);
  const F Mm = F::add(F::dbl(XX), XX);                                // < 3M
  const F T2 = F::mul(quad_pick<F>(role, U, p.x, Mm, Mm), quad_pick<F>(role, V, V, Mm, Mm));   // Wv | S | MM | -
  const F Wv = quad_bcast<0>(T2), S = quad_bcast<1>(T2), MM = quad_bcast<2>(T2);
  const F X3 = fsub<F, 2 * M>(MM, F::dbl(S));                         // < 3M
  const F T3 = F::mul(quad_pick<F>(role, Mm, Wv, V, Wv), quad_pick<F>(role, fsub<F, 3 * M>(S, X3), p.y, p.zz, p.zzz));
  const F Bv = quad_bcast<1>(T3);
  const F Yl = fsub<F, M>(T3, Bv);                                    // lane 0: Y3 = A - Bv < 2M (the other lanes' values are unused)
  p.x = X3;
  p.y = quad_bcast<0>(Yl);
  p.zz = quad_bcast<2>(T3);
  p.zzz = quad_bcast<3>(T3);
}
// acc += q (q from memory), the accumulator in registers; same rounds as xyzz_add_quad
template <class F>
__device__ __forceinline__ void xyzz_add_quad_reg(XYZZ<F>& acc, const XYZZ<F>& q, int role) {
  constexpr int M = F::MULB;
  if (q.is_inf()) return;
  if (acc.is_inf()) { acc = q; return; }
  const F T1 = F::mul(quad_pick<F>(role, acc.x, q.x, acc.y, q.y), quad_pick<F>(role, q.zz, acc.zz, q.zzz, acc.zzz));
  const F U1 = quad_bcast<0>(T1), U2 = quad_bcast<1>(T1), S1 = quad_bcast<2>(T1), S2 = quad_bcast<3>(T1);
  const F P = fsub<F, M>(U2, U1), R = fsub<F, M>(S2, S1);             // < 2M
  if (fis_zero_modp<F, 2 * M>(P)) {                                   // uniform: every lane holds the same values
    acc = xyzz_add_inl<F>(acc, q);
    return;
  }
  const F T2 = F::mul(quad_pick<F>(role, P, R, acc.zz, acc.zzz), quad_pick<F>(role, P, R, q.zz, q.zzz));   // PP | RR | Z2 | Z3
  const F PP = quad_bcast<0>(T2), RR = quad_bcast<1>(T2);
  const F T3 = F::mul(quad_pick<F>(role, P, U1, T2, T2), PP);          // PPP | Q | ZZ3 | (unused)
  const F PPP = quad_bcast<0>(T3), Qv = quad_bcast<1>(T3);
  const F X3 = fsub<F, 2 * M>(fsub<F, M>(RR, PPP), F::dbl(Qv));       // < 4M
  const F T4 = F::mul(quad_pick<F>(role, R, S1, T2, T2), quad_pick<F>(role, fsub<F, 4 * M>(Qv, X3), PPP, PPP, PPP));  // A | Bv | - | ZZZ3
  const F Bv = quad_bcast<1>(T4);
  const F Yl = fsub<F, M>(T4, Bv);                                    // lane 0: Y3 = A - Bv < 2M
  acc.x = X3;
  acc.y = quad_bcast<0>(Yl);
  acc.zz = quad_bcast<2>(T3);
  acc.zzz = quad_bcast<3>(T4);
}
// --- bit Horner of a window in groups of h bits: P_{w,g} = sum_{l in group g} 2^(l - g h) O_l (+ TOP in group 0) -----------
// One quad of lanes per (window, group) walks its chain with the accumulator resident in registers (every lane of the quad
// holds a full copy): a doubling is 3 rounds of independent products, an addition 4 --
//   doubling  round 1   V = U^2 (U = 2Y)   XX = X^2         -                -
//             round 2   Wv = U*V           S = X*V          MM = Mm^2        -              (Mm = 3 XX)
//             round 3   A = Mm*(S-X3)      Bv = Wv*Y        ZZ3 = V*ZZ       ZZZ3 = Wv*ZZZ  (X3 = MM - 2S, Y3 = A - Bv)
// a round costs ~1.8 us.  Round 2 ran ONE quad per window over all c - 1 bits (k_window_sums: 15 x 7 + 4 rounds = 200 us for
// BLS12-381 at c = 16) or left the whole bit Horner to the host; with groups of h = 4 bits the chain is 3 x 7 + 4 rounds (45 us)
// and the host's Horner over the windows -- W c doublings whatever h is -- takes ngrp - 1 more additions per window.
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_window_groups(const XYZZ<F>* out, XYZZ<F>* wsum, int c, int h, int ngrp) {
  // EC_BLOCK / 4 groups per workgroup; more groups than that (horner_bits = 1 with c >= 18: 17 .. 19 groups) go to blockIdx.y
  const uint32_t tid = threadIdx.x;
  const int role = (int)(tid & 3u), g = (int)(blockIdx.y * (EC_BLOCK / 4u) + (tid >> 2));
  if (g >= ngrp) return;   // (whole quads leave together)
  const XYZZ<F>* o = out + (size_t)blockIdx.x * c;
  const int lo = g * h;
  int hi = lo + h;
  if (hi > c - 1) hi = c - 1;
  XYZZ<F> r = XYZZ<F>::inf();
  for (int l = hi - 1; l >= lo; l--) {
    xyzz_dbl_quad_reg<F>(r, role);
    const XYZZ<F> y = o[l];
    xyzz_add_quad_reg<F>(r, y, role);
  }
  if (g == 0) {
    const XYZZ<F> top = o[c - 1];
    xyzz_add_quad_reg<F>(r, top, role);
  }
  if (role == 0) wsum[(size_t)blockIdx.x * ngrp + g] = r;
}

// a narrow pass of the bucket reduction: four lanes per addition (xyzz_add_quad)
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_pyr_quad(PyrArgs<F> a, uint32_t ntasks, uint32_t prio) {
  // The narrow passes are a chain of short dependent launches of few waves which, when MSMs are pipelined, share their SIMDs with the next MSM's
  // sort and accumulate kernels; at equal wave priority the chain -- the critical path of a small MSM -- stretches.  s_setprio 3 lets these waves
  // issue first: BLS12-381 G1 -5 ... -6 % per MSM at 2^14 ... 2^18 pairs with two in flight, BN254 -5 ... -9 % at 2^16 ... 2^19; level from 2^20 on,
  // and the host switches it off there and for the curves it does not help (Curve::NARROW_PRIO_LOG2N, profiles/wave_priority_r06.txt).
  if (prio) __builtin_amdgcn_s_setprio(3);
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = lane >> 2;
  const int role = (int)(lane & 3u);
  const XYZZ<F>* s1;
  const XYZZ<F>* s2;
  XYZZ<F>* d1;
  XYZZ<F>* d2;
  const bool live = t < ntasks && pyr_decode<F>(a, blockIdx.y, t, s1, s2, d1, d2);
  if (!live) return;  // whole quads leave together (t is the same for the four lanes)
  if (!s2) {
    if (role == 0) {
      const XYZZ<F> x = *s1;
      *d1 = x;
      if (d2) *d2 = x;
    }
    return;
  }
  xyzz_add_quad<F>(s1, s2, d1, d2, role);
}
static constexpr int RED_BLOCK = 256;
// end of the head merge: one workgroup per window (msm_bodies.h merge_finish_body)
template <class F>
__global__ void __launch_bounds__(RED_BLOCK) k_merge_finish(MergeArgs<F> a, uint32_t first_d) {
  merge_finish_body<F>(a, blockIdx.x, first_d, threadIdx.x, blockDim.x, []() { __syncthreads(); });
}
// the queue kernel with FOUR lanes per chain (xyzz_add_quad_reg: the running sum resident in the registers of the quad, the next head
// from memory): the queue holds a few ten thousand chains of one or two additions -- latency, not throughput -- and a four-lane addition is
// 4 products deep instead of 14.  Same box, merge stage of a blocking 2^16-pair call, one lane / four lanes per chain: see DESIGN.md 4.4.
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_merge_queue_quad(MergeArgs<F> a, uint32_t lmax) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t qi = lane >> 2;
  const int role = (int)(lane & 3u);
  if (qi >= *a.qcount) return;                  // (whole quads leave together)
  const uint32_t slot = a.queue[qi];
  const uint32_t w = slot / a.G, s = slot - w * a.G;
  const uint32_t b = a.hkey[slot];
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  const uint32_t len = chain_last(bs, a.K, b) - s + 1;
  if (len > lmax) return;                       // merge_long_body's
  XYZZ<F> acc = a.heads[slot];
  for (uint32_t i = 1; i < len; i++) {
    const XYZZ<F> h = a.heads[slot + i];
    xyzz_add_quad_reg<F>(acc, h, role);
  }
  if (role == 0) a.buckets[(uint64_t)w * a.B + b] = acc;
}
// the chains the chain form left (more than lmax heads: unusual inputs), one workgroup per window (msm_bodies.h merge_long_body)
template <class F>
__global__ void __launch_bounds__(RED_BLOCK) k_merge_long(MergeArgs<F> a, uint32_t lmax) {
  merge_long_body<F>(a, blockIdx.x, lmax, threadIdx.x, blockDim.x, []() { __syncthreads(); });
}
// head-merge tree step with four lanes per addition (a step has at most G / 2d additions per window)
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_merge_step_quad(MergeArgs<F> a, uint32_t d) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = lane >> 2;
  if (!merge_step_active<F>(a, blockIdx.y, g, d)) return;
  XYZZ<F>* h = a.heads + (uint64_t)blockIdx.y * a.G + g;
  xyzz_add_quad<F>(h, h + d, h, (XYZZ<F>*)nullptr, (int)(lane & 3u));
}
template <class C>
__global__ void __launch_bounds__(EC_BLOCK) k_gen_points(uint64_t seed, uint64_t first, uint32_t n, Affine<typename C::F>* out) {
  Affine<typename C::F> G = generator<C>();
  gen_point_body<typename C::F>(G, seed, first, n, out, blockIdx.x * blockDim.x + threadIdx.x);
}

template <class Fr>
__global__ void __launch_bounds__(EC_BLOCK) k_fr_quotient_inv(FrQuotientArgs<Fr> a) {
  fr_quotient_inv_body<Fr>(a, blockIdx.x * blockDim.x + threadIdx.x);
}
template <class Fr>
__global__ void __launch_bounds__(FR_QUOTIENT_SUM_LANES) k_fr_quotient_sum(FrQuotientArgs<Fr> a) {
  fr_quotient_sum_body<Fr>(a, threadIdx.x, FR_QUOTIENT_SUM_LANES);
  __syncthreads();   // (orders the workgroup's global writes: one workgroup, the sums are read by its lane 0)
  if (threadIdx.x == 0) fr_quotient_y_body<Fr>(a, FR_QUOTIENT_SUM_LANES);
}
template <class Fr>
__global__ void __launch_bounds__(256) k_fr_quotient_out(FrQuotientArgs<Fr> a) {
  fr_quotient_out_body<Fr>(a, blockIdx.x * blockDim.x + threadIdx.x);
}

template <class C>
__global__ void __launch_bounds__(EC_BLOCK) k_subgroup_check(const Affine<typename C::F>* pts, uint32_t n, uint8_t* ok) {
  subgroup_check_body<C>(pts, n, ok, blockIdx.x * blockDim.x + threadIdx.x);
}

// one level of a window table: next = 2^c * prev (msm_bodies.h table_next_body; run once per table)
template <class F>
__global__ void __launch_bounds__(EC_BLOCK) k_table_next(const Affine<F>* prev, Affine<F>* next, uint32_t n, int c) {
  table_next_body<F>(prev, next, n, c, blockIdx.x * blockDim.x + threadIdx.x);
}

// probe of the device field FD (msm_bodies.h dev_field_probe): inputs in the reference representation, output = raw FD limbs
template <class F, class FD>
__global__ void k_field_op_dev(int op, const F* a, const F* b, FD* r, uint32_t n) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if constexpr (FD::UNSAT) r[j] = dev_field_probe<FD>(op, FD::from_sat(a[j]), FD::from_sat(b[j]));
}

// field-op probe for the GPU unit tests: op 0 mul, 1 sqr, 2 add, 3 sub, 4 neg
template <class F>
__global__ void k_field_op(int op, const F* a, const F* b, F* r, uint32_t n) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  F x = a[j], y = b[j], o;
  switch (op) {
    case 0: o = F::mul(x, y); break;
    case 1: o = F::sqr(x); break;
    case 2: o = F::add(x, y); break;
    case 3: o = F::sub(x, y); break;
    default: o = F::neg(x); break;
  }
  r[j] = o;
}

// ---------------------------------------------------------------------------------------------
// HIP backend
// ---------------------------------------------------------------------------------------------
// k_accum<F, true> lives in a translation unit of its own per curve (the curve's .hip compiled again with -DCTT_TU_ACCUM_INTO, Makefile
// into_%.o): k_accum is by far the slowest kernel to compile (BLS12-381 G2: two minutes), and its second form then builds beside the
// first one instead of after it.  Declared here, defined and explicitly instantiated there.
template <class F>
void launch_accum_into(hipStream_t stream, const AccumArgs<F>& a, uint32_t W);
#ifdef CTT_TU_ACCUM_INTO
template <class F>
void launch_accum_into(hipStream_t stream, const AccumArgs<F>& a, uint32_t W) {
  hipLaunchKernelGGL((k_accum<F, true>), dim3((a.G + ACCUM_BLOCK - 1) / ACCUM_BLOCK, W), dim3(ACCUM_BLOCK), 0, stream, a);
}
#endif

struct HipBackend {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // Tail stream: the narrow (latency-bound) passes at the end of the bucket reduction and the result copy of MSM i run
  // here, so that the first kernels of MSM i+1 (point conversion, sort: memory- and LDS-bound) run underneath them.
  hipStream_t aux = nullptr;
  hipEvent_t ev_tail_fork = nullptr, ev_tail_done = nullptr;
  bool on_aux = false, tail_pending = false;
  hipStream_t cur() const { return on_aux ? aux : stream; }
  bool no_tail = false;   // experiment knob ($CTT_HIP_MSM_NO_TAIL): everything on the main stream
  void tail_begin() {
    if (no_tail) return;
    HIP_CHECK(hipEventRecord(ev_tail_fork, stream));
    HIP_CHECK(hipStreamWaitEvent(aux, ev_tail_fork, 0));
    on_aux = true;
  }
  bool tail_forked() const { return on_aux; }
  void tail_end() {
    if (no_tail) return;
    HIP_CHECK(hipEventRecord(ev_tail_done, aux));
    on_aux = false;
    tail_pending = true;
  }
  // Front stream (round 5): the conversion and the sort of a SMALL MSM that a caller keeps in flight run here, beside the previous MSM's
  // head merge and first reduction pass, instead of behind them on the main stream.  They need the coefficients and points only, write
  // nothing the previous MSM still reads once its accumulation has ended (front_begin waits for that), and are memory- / LDS-bound where
  // the merge is a latency chain of additions: the 60-70 us of a 2^16-2^17-pair MSM's sort drop out of the step.  front_end joins:
  // the accumulation on the main stream waits for the sort.  (The large sizes get the same overlap from the early tail, msm_pipeline.h.)
  hipStream_t srt = nullptr;
  hipEvent_t ev_accum_done = nullptr, ev_front_done = nullptr;
  bool on_front = false, accum_marked = false;
  hipStream_t front() const { return on_front ? srt : stream; }
  void front_begin() {
    // the previous accumulation has read the shared entry list / records: its own event when it left one (accum_mark), else -- the first
    // MSM of a pipeline -- everything the main stream holds now (no overlap for that one, same order as without the front stream)
    if (!accum_marked) HIP_CHECK(hipEventRecord(ev_accum_done, stream));
    HIP_CHECK(hipStreamWaitEvent(srt, ev_accum_done, 0));
    on_front = true;
  }
  void front_abort() { on_front = false; accum_marked = false; }
  void front_end() {
    HIP_CHECK(hipEventRecord(ev_front_done, srt));
    HIP_CHECK(hipStreamWaitEvent(stream, ev_front_done, 0));
    on_front = false;
  }
  // behind every accumulate launch: with an event for the next MSM's front stage when one is expected (an event record is a barrier
  // packet in the queue: not for callers that do not pipeline small MSMs)
  void accum_mark(bool wanted) {
    if (wanted) HIP_CHECK(hipEventRecord(ev_accum_done, stream));
    accum_marked = wanted;
  }
  // "the wide reduction passes of the tail are done": what the next MSM's accumulation waits for when the tail starts early
  // (MsmEngine::reduce_buckets) -- the narrow rest of the tail may run beside that accumulation, the wide passes may not
  hipEvent_t ev_wide_done = nullptr;
  bool wide_pending = false;
  void wide_mark() {
    if (!on_aux) return;
    HIP_CHECK(hipEventRecord(ev_wide_done, aux));
    wide_pending = true;
  }
  // partitioned chip: the mark goes behind the head merge instead (MsmEngine::submit)
  void merge_mark() { wide_mark(); }
  void wide_wait() {
    if (!wide_pending) return;
    HIP_CHECK(hipStreamWaitEvent(stream, ev_wide_done, 0));
    wide_pending = false;
  }
  // before anything on the main stream touches what a tail still reads or writes (pyramid, trees, per-window output)
  void tail_wait() {
    if (!tail_pending) return;
    HIP_CHECK(hipStreamWaitEvent(stream, ev_tail_done, 0));
    tail_pending = false;
    wide_pending = false;   // (the end of the tail is behind its wide passes)
  }
  // four lanes per addition below this many additions per pass (all windows).
  // rounds 1-4: 24576 (measured at 2^20: 18 us vs 19.6 us at 18432 additions, 24 us vs 21 us at 32768).  Round 5 swept it again with the
  // small sizes in view (profiles/pyr_quad_threshold_r05.txt, same box, ms per MSM with two in flight at 24576 / 49152 / 131072):
  // 2^16 0.443-0.454 / 0.439-0.450 / 0.436-0.448, 2^17 0.631 / 0.619 / 0.614, 2^18 0.978-0.993 / 0.974-0.986 / 0.965-0.989, 2^20 2.79-2.88 /
  // 2.83-2.89 / 2.89-2.90 (blocking 3.13-3.17 / 3.14-3.16 / 3.23-3.25): 49152 takes the small sizes' 2 % and leaves 2^20 where it was
  uint32_t quad_adds = 49152u;       // $CTT_HIP_MSM_QUAD (read when the context is created)
  // passes with at most this many additions (all windows) go to the tail stream: about what fits under the next MSM's conversion + sort
  // (measured 2^20: 3.34 ms at 24576, 3.30 at 131072; above that the tail queues behind the next accumulation)
  uint32_t tail_adds = 131072u;      // $CTT_HIP_MSM_TAIL
  bool pyr_is_narrow(uint32_t ntasks, uint32_t W) const { return (uint64_t)ntasks * W <= quad_adds; }
  bool pyr_goes_to_tail(uint32_t ntasks, uint32_t W) const { return (uint64_t)ntasks * W <= tail_adds; }
  // ---- spatial partition of the chip (round 6): CU-masked streams ------------------------------------------------------------
  // Every overlap of rounds 1-5 was temporal: the tail of MSM i (head merge, reduction passes, bit Horner, result copy) and the
  // accumulation of MSM i+1 shared the SIMDs and instruction caches of every CU.  $CTT_HIP_CU_TAIL = r > 0 creates the tail stream with
  // hipExtStreamCreateWithCUMask over r compute units of every XCD -- the low 8 r bits of the mask: the runtime deals the user mask
  // round-robin over the XCDs (tools/cu_mask_probe.hip prints the placement) -- and, unless $CTT_HIP_CU_MAIN = 0, the main stream over
  // the complement: the accumulate grid is then sized for the CUs it really has (resident_lanes), never waits for a tail and leaves no
  // wave slots free on purpose.  Measured: profiles/cu_mask_r06.txt, DESIGN.md.
  int cu_tail = 0;        // CUs per XCD in the tail stream's mask (0: an ordinary high-priority stream over the whole chip)
  bool cu_main = false;   // the main stream is masked to the complement
  static constexpr int XCDS = 8;
  bool partitioned() const { return cu_tail > 0 && cu_main; }
  int main_cus() const { return partitioned() ? num_cu - XCDS * cu_tail : num_cu; }
  int num_cu = 256;
  // stage events per in-flight slot; a host-pointer MSM runs the first stages once per upload slice (MsmEngine::submit_host):
  // every slice records its own pair (stage_chunk) and collect_timings adds them up
  static constexpr int MAX_CHUNKS = 8;
  static constexpr int NSLOT = 3;   // (= MsmEngine::NSLOT)
  hipEvent_t ev_begin[NSLOT][ST_COUNT][MAX_CHUNKS] = {}, ev_end[NSLOT][ST_COUNT][MAX_CHUNKS] = {};
  uint32_t ev_used[NSLOT][ST_COUNT];   // bit i: chunk i recorded
  int chunk = 0;
  void stage_chunk(int i) { chunk = i < MAX_CHUNKS ? i : MAX_CHUNKS - 1; }
  hipEvent_t ev_done[NSLOT] = {};
  float stage_ms[ST_COUNT];

  void init(int dev);  // msm_engine.hip

  void* alloc(size_t b) {
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, b);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {   // recoverable: the call fails, the process lives
      (void)hipGetLastError();
      fprintf(stderr, "[ctt_msm_hip] out of device memory: hipMalloc of %zu bytes failed\n", b);
      throw OutOfDeviceMemory{b};
    }
    HIP_CHECK(e);
    return p;
  }
  void free(void* p) { HIP_CHECK(hipFree(p)); }
  // teardown paths (destructors, a context lost to a HIP failure): nothing to report to and nothing that may throw -- a failing hipFree
  // on a context with a sticky error must not turn the documented recovery ("destroy it and create a new one") into std::terminate
  void free_quiet(void* p) noexcept {
    if (p && hipFree(p) != hipSuccess) (void)hipGetLastError();
  }
  void free_host_quiet(void* p) noexcept {
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
  }
  void shutdown() noexcept;   // msm_engine.hip: streams and events of init()
  void memset0(void* p, size_t b) { HIP_CHECK(hipMemsetAsync(p, 0, b, stream)); }
  void d2d_async(void* dst, const void* src, size_t b) {
    HIP_CHECK(hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToDevice, stream));
  }
  void* alloc_host(size_t b) {
    void* p = nullptr;
    HIP_CHECK(hipHostMalloc(&p, b, hipHostMallocDefault));
    return p;
  }
  void free_host(void* p) { HIP_CHECK(hipHostFree(p)); }
  // Host -> device on the copy stream: a pageable source keeps the calling thread busy for the duration of the copy but
  // not the compute queues (profiles/h2d_overlap_r02.jsonl: 56 GB/s with or without a kernel holding every wave slot).
  hipStream_t cpy = nullptr;
  hipEvent_t ev_copy = nullptr;
  void h2d(void* dst, const void* src, size_t b) {
    if (b) HIP_CHECK(hipMemcpyAsync(dst, src, b, hipMemcpyHostToDevice, cpy));
  }
  // what is enqueued on the main stream from now on sees the copies made so far
  void h2d_done() {
    HIP_CHECK(hipEventRecord(ev_copy, cpy));
    HIP_CHECK(hipStreamWaitEvent(stream, ev_copy, 0));
  }
  // The slices of a host-pointer MSM are copied by a thread of their own (MsmEngine::submit_host): a pageable hipMemcpyAsync
  // returns when its last byte has left, and a caller that enqueues the ~15 launches of slice i between the copies of slices i and
  // i+1 leaves the link idle for ~0.1 ms per slice (profiles/hostptr_timeline_r04.txt).  The uploader records ev_slice[i] behind the
  // copies of slice i and raises a host flag; the submitting thread waits for the flag (an event that has not been recorded yet
  // would not hold the stream) and lets the main stream wait for the event.
  static constexpr bool THREADED_UPLOAD = true;
  hipEvent_t ev_slice[MAX_CHUNKS] = {};
  void (*uploader_hook)(int device) = nullptr;   // (msm_engine.hip: the NUMA pinning of the uploader thread)
  void uploader_begin() {
    HIP_CHECK(hipSetDevice(device));
    if (uploader_hook) uploader_hook(device);
  }
  using GuardScope = ErrorGuard;   // "this helper thread's HIP failures are thrown, not fatal" (MsmEngine::submit_host's uploader)
  [[noreturn]] void uploader_failed() {
    hip_failed("upload of a host-pointer MSM's slices (helper thread)", "a copy or event record failed on the uploader thread", 0, __FILE__, __LINE__);
  }
  void h2d_slice_done(uint32_t i) { HIP_CHECK(hipEventRecord(ev_slice[i], cpy)); }
  void h2d_slice_wait(uint32_t i) { HIP_CHECK(hipStreamWaitEvent(stream, ev_slice[i], 0)); }
  // (the coefficients of slice i alone: the digits and the sort can start on them while the slice's points are still crossing)
  hipEvent_t ev_coefs[MAX_CHUNKS] = {};
  void h2d_coefs_done(uint32_t i) { HIP_CHECK(hipEventRecord(ev_coefs[i], cpy)); }
  void h2d_coefs_wait(uint32_t i) { HIP_CHECK(hipStreamWaitEvent(stream, ev_coefs[i], 0)); }
  void d2h_async(int slot, void* dst_pinned, const void* src, size_t b) {
    HIP_CHECK(hipMemcpyAsync(dst_pinned, src, b, hipMemcpyDeviceToHost, cur()));
    HIP_CHECK(hipEventRecord(ev_done[slot], cur()));
  }
  void d2h_wait(int slot) { HIP_CHECK(hipEventSynchronize(ev_done[slot])); }
  void d2h_sync(void* dst, const void* src, size_t b) {
    HIP_CHECK(hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  void launch_iota(uint32_t* entries, uint32_t n, uint32_t* bucket_start, uint32_t* maxcount) {
    hipLaunchKernelGGL(k_iota, grid1(n, 256), dim3(256), 0, stream, entries, n, bucket_start, maxcount);
    HIP_CHECK(hipGetLastError());
  }
  // Stage timing is opt-in (ctt_hip_msm_set_option "timings"): twelve event records per MSM are host time a caller of a
  // small MSM should not pay for a number it does not read.
  int timing = 0;   // 0 off, 1 every stage, 2 the accumulate stage only (what a roofline needs; an event record is a barrier
                    // packet in the queue: twelve per MSM cost a small pipelined MSM 15 % -- 2^16: 0.63 ms per step against 0.55)
  // "timings_every" k: only every k-th MSM records its events (the others report zeros) -- a 0.5 ms pipelined MSM pays 12 % for
  // four records, and an average over a quarter of the launches of a timed loop is as good an average
  int timing_every = 1;
  uint32_t timing_tick = 0;
  bool timing_live = false;   // this MSM records (decided when its ST_TOTAL stage begins)
  bool stage_on(int s) const { return timing_live && (timing == 1 || (timing == 2 && (s == ST_ACCUM || s == ST_TOTAL))); }
  void stage_begin(int slot, int s) {
    if (s == ST_TOTAL) {
      timing_live = timing != 0 && (timing_tick++ % (uint32_t)(timing_every < 1 ? 1 : timing_every)) == 0u;
      for (int i = 0; i < ST_COUNT; i++) ev_used[slot][i] = 0;
      chunk = 0;
    }
    if (!stage_on(s)) return;
    const int ch = (s == ST_TOTAL || s == ST_REDUCE) ? 0 : chunk;
    HIP_CHECK(hipEventRecord(ev_begin[slot][s][ch], (s == ST_DIGITS || s == ST_SORT) ? front() : cur()));
    ev_used[slot][s] |= 1u << ch;
  }
  void stage_end(int slot, int s) {
    if (!stage_on(s)) return;
    const int ch = (s == ST_TOTAL || s == ST_REDUCE) ? 0 : chunk;
    HIP_CHECK(hipEventRecord(ev_end[slot][s][ch], (s == ST_DIGITS || s == ST_SORT) ? front() : cur()));
  }
  // stage times of the MSM that used `slot` (call after its finish())
  void collect_timings(int slot) {
    for (int i = 0; i < ST_COUNT; i++) {
      stage_ms[i] = 0.f;
      if (!timing) continue;
      for (int ch = 0; ch < MAX_CHUNKS; ch++) {
        if (!(ev_used[slot][i] >> ch & 1u)) continue;
        float ms = 0.f;
        HIP_CHECK(hipEventSynchronize(ev_end[slot][i][ch]));
        HIP_CHECK(hipEventElapsedTime(&ms, ev_begin[slot][i][ch], ev_end[slot][i][ch]));
        stage_ms[i] += ms;
      }
    }
  }

  template <class F>
  uint32_t resident_lanes() {
    int nb = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_accum<F, false>, ACCUM_BLOCK, 0));
    if (nb < 1) nb = 1;
    return (uint32_t)nb * ACCUM_BLOCK * (uint32_t)main_cus();
  }

  static dim3 grid1(uint32_t n, int block) { return dim3((n + block - 1) / block); }
  static dim3 grid2(uint32_t n, int block, uint32_t W) { return dim3((n + block - 1) / block, W); }

  template <class Fr>
  void launch_fr_from_mont(const uint32_t* in, uint32_t* out, uint32_t n) {
    hipLaunchKernelGGL(k_fr_from_mont<Fr>, grid1(n, 256), dim3(256), 0, front(), in, out, n);
    HIP_CHECK(hipGetLastError());
  }
  template <class F, class FD>
  void launch_convert(const Affine<F>* in, void* out, uint32_t n) {
    hipLaunchKernelGGL((k_convert_points<F, FD>), grid1(n, CONVERT_BLOCK), dim3(CONVERT_BLOCK),
                       (size_t)CONVERT_BLOCK * gather_stride<FD>(), front(), in, out, n);
    HIP_CHECK(hipGetLastError());
  }
  template <class F>
  void launch_table_next(const Affine<F>* prev, Affine<F>* next, uint32_t n, int c) {
    hipLaunchKernelGGL(k_table_next<F>, grid1(n, EC_BLOCK), dim3(EC_BLOCK), 0, stream, prev, next, n, c);
    HIP_CHECK(hipGetLastError());
  }
  void sync() { HIP_CHECK(hipStreamSynchronize(stream)); }
  void launch_digits_sort(const SortArgs& a);  // msm_engine.hip
  // into: the bucket set holds earlier sums that the runs continue (a later slice of a host-pointer MSM, accum_body_xyzz)
  template <class F>
  void launch_accum(const AccumArgs<F>& a, uint32_t W, bool into = false) {
    if (into) launch_accum_into<F>(stream, a, W);
    else hipLaunchKernelGGL((k_accum<F, false>), grid2(a.G, ACCUM_BLOCK, W), dim3(ACCUM_BLOCK), 0, stream, a);
    HIP_CHECK(hipGetLastError());
  }
  template <class F>
  void launch_merge_tail(const MergeArgs<F>& a, uint32_t W) {
    hipLaunchKernelGGL(k_merge_tail<F>, grid2(a.G, EC_BLOCK, W), dim3(EC_BLOCK), 0, cur(), a);
    HIP_CHECK(hipGetLastError());
  }
  template <class F>
  void launch_merge_step(const MergeArgs<F>& a, uint32_t W, uint32_t d) {
    if ((uint64_t)a.G * W * 4u <= 1048576u) {  // the additions of a step are sparse: four lanes each
      hipLaunchKernelGGL(k_merge_step_quad<F>, grid2(a.G * 4u, EC_BLOCK, W), dim3(EC_BLOCK), 0, cur(), a, d);
      HIP_CHECK(hipGetLastError());
      return;
    }
    hipLaunchKernelGGL(k_merge_step<F>, grid2(a.G, EC_BLOCK, W), dim3(EC_BLOCK), 0, cur(), a, d);
    HIP_CHECK(hipGetLastError());
  }
  // the steps beyond those the plan enqueued (unusual inputs only: returns at once otherwise), then the chain heads -> buckets
  template <class F>
  void launch_merge_finish(const MergeArgs<F>& a, uint32_t W, uint32_t first_d) {
    hipLaunchKernelGGL(k_merge_finish<F>, dim3(W), dim3(RED_BLOCK), 0, cur(), a, first_d);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_merge_final<F>, grid2(a.G, EC_BLOCK, W), dim3(EC_BLOCK), 0, cur(), a);
    HIP_CHECK(hipGetLastError());
  }
  template <class F>
  void launch_merge_tail_queue(const MergeArgs<F>& a, uint32_t W) {
    hipLaunchKernelGGL(k_merge_tail_queue<F>, grid2(a.G, EC_BLOCK, W), dim3(EC_BLOCK), 0, cur(), a);
    HIP_CHECK(hipGetLastError());
  }
  template <class F>
  void launch_merge_queue(const MergeArgs<F>& a, uint32_t W, uint32_t lmax, bool quad) {
    // at most every second lane starts a chain of two or more heads
    if (quad) hipLaunchKernelGGL(k_merge_queue_quad<F>, grid1(merge_queue_capacity(W, a.G) * 4u, EC_BLOCK), dim3(EC_BLOCK), 0, cur(), a, lmax);
    else hipLaunchKernelGGL(k_merge_queue<F>, grid1(merge_queue_capacity(W, a.G), EC_BLOCK), dim3(EC_BLOCK), 0, cur(), a, lmax);
    HIP_CHECK(hipGetLastError());
  }
  template <class F>
  void launch_merge_long(const MergeArgs<F>& a, uint32_t W, uint32_t lmax) {
    hipLaunchKernelGGL(k_merge_long<F>, dim3(W), dim3(RED_BLOCK), 0, cur(), a, lmax);
    HIP_CHECK(hipGetLastError());
  }
  template <class F>
  void launch_window_groups(const XYZZ<F>* out, XYZZ<F>* wsum, uint32_t W, int c, int h, int ngrp) {
    // the kernel is compiled for EC_BLOCK lanes (16 quads): the groups beyond 16 of a window take further blocks in y
    const uint32_t gy = ((uint32_t)ngrp + EC_BLOCK / 4u - 1u) / (EC_BLOCK / 4u);
    hipLaunchKernelGGL(k_window_groups<F>, dim3(W, gy), dim3(EC_BLOCK), 0, cur(), out, wsum, c, h, ngrp);
    HIP_CHECK(hipGetLastError());
  }
  // raised wave priority for the narrow reduction passes of the MSM being submitted (k_pyr_quad): the pipeline decides per curve and size
  bool narrow_prio = false;
  void narrow_priority(bool on) { narrow_prio = on; }
  template <class F>
  void launch_pyr(const PyrArgs<F>& a, uint32_t W, uint32_t ntasks) {
    // few tasks left: four lanes per addition (the chip is mostly idle, the addition is 3.5x shallower)
    if (pyr_is_narrow(ntasks, W)) {
      hipLaunchKernelGGL(k_pyr_quad<F>, grid2(ntasks * 4u, EC_BLOCK, W), dim3(EC_BLOCK), 0, cur(), a, ntasks, narrow_prio ? 1u : 0u);
      HIP_CHECK(hipGetLastError());
      return;
    }
    hipLaunchKernelGGL(k_pyr<F>, grid2(ntasks, EC_BLOCK, W), dim3(EC_BLOCK), 0, cur(), a, ntasks);
    HIP_CHECK(hipGetLastError());
  }
};


// ---------------------------------------------------------------------------------------------
// Per-curve operation table: each curve_*.hip instantiates the templates once and exports one of these
// ---------------------------------------------------------------------------------------------
struct CurveOps {
  int curve_id;
  size_t aff_bytes;
  void* (*engine_create)(HipBackend* bk);
  void (*engine_destroy)(void* eng);
  // split form: at most MsmEngine::NSLOT (3) MSMs in flight per engine; submit returns the slot, or -1 when all are taken;
  // finish returns 0, or -1 when the slot is not in flight
  int (*submit)(void* eng, const MsmOptions* opt, const void* d_coefs, int coef_is_fr, const void* d_points, uint32_t n,
                int* plan);
  int (*finish)(void* eng, int slot, void* r_host, int out_kind);
  // host-resident inputs, uploaded in slices underneath the accumulation (MsmEngine::submit_host); chunks 0 = automatic
  int (*submit_host)(void* eng, const MsmOptions* opt, const void* h_coefs, int coef_is_fr, const void* h_points, uint32_t n,
                     void* d_stage_coefs, void* d_stage_points, int chunks, int* plan);
  // cached bases: device records for `n` points (d_points in the C-API layout, device memory); submit against them
  void* (*bases_prepare)(void* eng, const void* d_points, uint32_t n);
  // (table_c > 0: d_prepared is a window table over table_n bases made by table_prepare)
  int (*submit_bases)(void* eng, const MsmOptions* opt, const void* d_coefs, int coef_is_fr, const void* d_prepared,
                      uint32_t n, int table_c, uint32_t table_n, int* plan);
  void (*gen_points)(HipBackend* bk, uint64_t seed, uint64_t first, uint32_t n, void* d_out);
  void (*field_op)(HipBackend* bk, int op, const void* d_a, const void* d_b, void* d_r, uint32_t n);
  // host-only: r_aff = sum of n affine points (combining the per-GPU partial results of a sharded MSM,
  // the `r ~+= partial` of ec_multi_scalar_mul_parallel.nim:427-429)
  void (*ec_sum_affine)(const void* pts_aff, size_t n, void* r_host, int out_kind);
  // r = sum of n affine points resident on the device (sum_reduce_vartime); returns the K used
  int (*sum_reduce)(void* eng, const MsmOptions* opt, const void* d_points, uint32_t n, void* r_host, int out_kind);
  // dst[i] = affine(src[i]) for n Jacobian (src_kind 1) or projective (2) points, device memory, K points per lane
  void (*batch_affine)(HipBackend* bk, int src_kind, void* d_dst, const void* d_src, uint32_t n, uint32_t K);
  size_t fe_bytes;  // one coordinate
  // ok[j] = [r]P_j is the neutral element (r = the curve order), n points and n flags in device memory
  void (*subgroup_check)(HipBackend* bk, const void* d_points, uint32_t n, void* d_ok);
  // window table over n bases (MsmEngine::prepare_table): records of 2^(c*w) * P_j for every digit window; c = 0 chooses
  void* (*table_prepare)(void* eng, const void* d_points, uint32_t n, int c, int* c_out);
  // KZG quotient polynomial over the curve's scalar field (msm_bodies.h FrQuotientArgs): d_poly canonical, d_dom Montgomery,
  // z canonical (host, fr_bytes), d_work >= (n + ceil(n/8)) * fr_bytes; q (device, canonical) and y (host, canonical) out.
  // Returns 0, or -2 when z is one of the n-th roots of unity (the caller's other formula applies).
  int (*fr_quotient)(HipBackend* bk, const void* d_poly, const void* d_dom, const void* z_host, uint32_t n, void* d_work,
                     void* d_q, void* y_host);
  size_t fr_bytes;
  // the same check on the host, for a handful of host-resident points (a precompile call, one commitment): a single GPU lane
  // walks the 255 dependent doublings of [r]P in ~6.5 ms whatever n is, a CPU core needs ~0.2 ms per G1 point
  void (*subgroup_check_host)(const void* pts_aff, size_t first, size_t step, size_t n, uint8_t* ok);
  // cached bases (records, or a window table when table_c > 0) with HOST-resident coefficients: the coefficients go up in slices
  // underneath the accumulation like the pairs of submit_host (MsmEngine::submit_host with d_prepared)
  int (*submit_host_bases)(void* eng, const MsmOptions* opt, const void* h_coefs, int coef_is_fr, const void* d_prepared, uint32_t n,
                           int table_c, uint32_t table_n, void* d_stage_coefs, int chunks, int* plan);
};

template <class C>
struct CurveImpl {
  using F = typename C::F;
  using FD = typename C::FD;
  using Engine = MsmEngine<C, HipBackend>;
  static void* create(HipBackend* bk) {
    Engine* e = new Engine(*bk);
    e->opt.lanes = bk->template resident_lanes<FD>();
    return e;
  }
  static void destroy(void* e) { delete (Engine*)e; }
  static int submit(void* eng, const MsmOptions* opt, const void* d_coefs, int coef_is_fr, const void* d_points, uint32_t n,
                    int* plan) {
    Engine& e = *(Engine*)eng;
    uint32_t lanes = e.opt.lanes;
    e.opt = *opt;
    e.opt.lanes = lanes;
    int sl = e.submit((const uint32_t*)d_coefs, coef_is_fr != 0, (const Affine<F>*)d_points, n);
    if (sl < 0) return sl;
    const MsmPlan& p = e.last_plan;
    plan[0] = p.c; plan[1] = p.W; plan[2] = (int)p.K; plan[3] = (int)p.G; plan[4] = (int)p.S; plan[5] = (int)lanes;
    return sl;
  }
  static int submit_host(void* eng, const MsmOptions* opt, const void* h_coefs, int coef_is_fr, const void* h_points, uint32_t n,
                         void* d_stage_coefs, void* d_stage_points, int chunks, int* plan) {
    Engine& e = *(Engine*)eng;
    uint32_t lanes = e.opt.lanes;
    e.opt = *opt;
    e.opt.lanes = lanes;
    int sl = e.submit_host(h_coefs, coef_is_fr != 0, h_points, n, d_stage_coefs, d_stage_points, chunks);
    if (sl < 0) return sl;
    const MsmPlan& p = e.last_plan;
    plan[0] = p.c; plan[1] = p.W; plan[2] = (int)p.K; plan[3] = (int)p.G; plan[4] = (int)p.S; plan[5] = (int)lanes;
    plan[7] = (int)e.last_chunks;
    return sl;
  }
  static int submit_host_bases(void* eng, const MsmOptions* opt, const void* h_coefs, int coef_is_fr, const void* d_prepared, uint32_t n,
                               int table_c, uint32_t table_n, void* d_stage_coefs, int chunks, int* plan) {
    Engine& e = *(Engine*)eng;
    uint32_t lanes = e.opt.lanes;
    e.opt = *opt;
    e.opt.lanes = lanes;
    int sl = e.submit_host(h_coefs, coef_is_fr != 0, nullptr, n, d_stage_coefs, nullptr, chunks, d_prepared, table_c, table_n);
    if (sl < 0) return sl;
    const MsmPlan& p = e.last_plan;
    plan[0] = p.c; plan[1] = p.Wd; plan[2] = (int)p.K; plan[3] = (int)p.G; plan[4] = (int)p.S; plan[5] = (int)lanes;
    plan[7] = (int)e.last_chunks;
    return sl;
  }
  static void* bases_prepare(void* eng, const void* d_points, uint32_t n) {
    return ((Engine*)eng)->prepare_bases((const Affine<F>*)d_points, n);
  }
  // table_c > 0: d_prepared is a window table over table_n bases (table_prepare)
  static int submit_bases(void* eng, const MsmOptions* opt, const void* d_coefs, int coef_is_fr, const void* d_prepared,
                          uint32_t n, int table_c, uint32_t table_n, int* plan) {
    Engine& e = *(Engine*)eng;
    uint32_t lanes = e.opt.lanes;
    e.opt = *opt;
    e.opt.lanes = lanes;
    int sl = e.submit((const uint32_t*)d_coefs, coef_is_fr != 0, nullptr, n, d_prepared, table_c, table_n);
    if (sl < 0) return sl;
    const MsmPlan& p = e.last_plan;
    plan[0] = p.c; plan[1] = p.Wd; plan[2] = (int)p.K; plan[3] = (int)p.G; plan[4] = (int)p.S; plan[5] = (int)lanes;
    return sl;
  }
  static void* table_prepare(void* eng, const void* d_points, uint32_t n, int c, int* c_out) {
    return ((Engine*)eng)->prepare_table((const Affine<F>*)d_points, n, c, c_out);
  }
  static int finish(void* eng, int slot, void* r_host, int out_kind) {
    Engine& e = *(Engine*)eng;
    if (!e.in_flight(slot)) return -1;
    auto res = e.finish(slot);
    write_result<typename Engine::HF>(r_host, res, out_kind);
    return 0;
  }
  static void gen_points(HipBackend* bk, uint64_t seed, uint64_t first, uint32_t n, void* d_out) {
    hipLaunchKernelGGL(k_gen_points<C>, dim3((n + EC_BLOCK - 1) / EC_BLOCK), dim3(EC_BLOCK), 0, bk->stream, seed, first, n,
                       (Affine<F>*)d_out);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(bk->stream));
  }
  static void field_op(HipBackend* bk, int op, const void* d_a, const void* d_b, void* d_r, uint32_t n) {
    if (op >= 16) {  // device-field probe (raw FD limbs out); only meaningful when FD != F
      hipLaunchKernelGGL((k_field_op_dev<F, FD>), dim3((n + 255) / 256), dim3(256), 0, bk->stream, op - 16, (const F*)d_a,
                         (const F*)d_b, (FD*)d_r, n);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(bk->stream));
      return;
    }
    hipLaunchKernelGGL(k_field_op<F>, dim3((n + 255) / 256), dim3(256), 0, bk->stream, op, (const F*)d_a, (const F*)d_b,
                       (F*)d_r, n);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(bk->stream));
  }
  static void ec_sum_affine(const void* pts_aff, size_t n, void* r_host, int out_kind) {
    using HF = typename Engine::HF;
    const Affine<HF>* p = (const Affine<HF>*)pts_aff;
    XYZZ<HF> acc = XYZZ<HF>::inf();
    for (size_t i = 0; i < n; i++) xyzz_madd<HF>(acc, p[i], false);
    write_result<HF>(r_host, acc, out_kind);
  }
  static int sum_reduce(void* eng, const MsmOptions* opt, const void* d_points, uint32_t n, void* r_host, int out_kind) {
    Engine& e = *(Engine*)eng;
    uint32_t lanes = e.opt.lanes;
    e.opt = *opt;
    e.opt.lanes = lanes;
    auto res = e.sum_reduce((const Affine<F>*)d_points, n);
    write_result<typename Engine::HF>(r_host, res, out_kind);
    return (int)e.last_sum_K;
  }
  static void batch_affine(HipBackend* bk, int src_kind, void* d_dst, const void* d_src, uint32_t n, uint32_t K) {
    if (n == 0) return;
    BatchAffineArgs<F> a{(const F*)d_src, (Affine<F>*)d_dst, n, src_kind, K};
    const uint32_t lanes = (n + K - 1) / K;
    hipLaunchKernelGGL(k_batch_affine<F>, dim3((lanes + EC_BLOCK - 1) / EC_BLOCK), dim3(EC_BLOCK), 0, bk->stream, a);
    HIP_CHECK(hipGetLastError());
  }
  static void subgroup_check(HipBackend* bk, const void* d_points, uint32_t n, void* d_ok) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_subgroup_check<C>, dim3((n + EC_BLOCK - 1) / EC_BLOCK), dim3(EC_BLOCK), 0, bk->stream,
                       (const Affine<F>*)d_points, n, (uint8_t*)d_ok);
    HIP_CHECK(hipGetLastError());
  }
  // points first, first + step, ... < n (one host thread's share); the same templates as the kernel, over the host's 64-bit field
  static void subgroup_check_host(const void* pts_aff, size_t first, size_t step, size_t n, uint8_t* ok) {
    using HF = typename Engine::HF;
    const Affine<HF>* p = (const Affine<HF>*)pts_aff;
    for (size_t j = first; j < n; j += step) ok[j] = point_in_subgroup<C, HF>(p[j]) ? 1 : 0;
  }
  static int fr_quotient(HipBackend* bk, const void* d_poly, const void* d_dom, const void* z_host, uint32_t n, void* d_work,
                         void* d_q, void* y_host) {
    using Fr = typename C::Fr;
    using HFr = Fp64<typename Fr::Params>;
    // host side: z -> Montgomery, z^n, (z^n - 1) / n
    HFr zc, r2, nn = HFr::zero();
    memcpy(zc.l, z_host, sizeof(zc.l));
    for (int i = 0; i < HFr::N; i++) r2.l[i] = (uint64_t)Fr::Params::R2[2 * i] | ((uint64_t)Fr::Params::R2[2 * i + 1] << 32);
    const HFr zm = HFr::mul(zc, r2);
    HFr zn = HFr::one();
    for (int b = 31; b >= 0; b--) {
      zn = HFr::sqr(zn);
      if ((n >> b) & 1u) zn = HFr::mul(zn, zm);
    }
    if (HFr::eq(zn, HFr::one())) return -2;
    nn.l[0] = n;
    const HFr scale = HFr::mul(HFr::sub(zn, HFr::one()), HFr::inv(HFr::mul(nn, r2)));
    FrQuotientArgs<Fr> a;
    a.poly = (const uint32_t*)d_poly;
    a.dom = (const uint32_t*)d_dom;
    memcpy(a.z.l, zm.l, sizeof(a.z.l));
    memcpy(a.scale.l, scale.l, sizeof(a.scale.l));
    a.n = n;
    a.K = 8;
    a.inv = (uint32_t*)d_work;
    a.partial = a.inv + (size_t)n * Fr::N;
    a.q = (uint32_t*)d_q;
    a.y = a.partial + (size_t)((n + a.K - 1) / a.K) * Fr::N;
    a.tsum = a.y + Fr::N;
    const uint32_t lanes = (n + a.K - 1) / a.K;
    hipLaunchKernelGGL(k_fr_quotient_inv<Fr>, dim3((lanes + EC_BLOCK - 1) / EC_BLOCK), dim3(EC_BLOCK), 0, bk->stream, a);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_fr_quotient_sum<Fr>, dim3(1), dim3(FR_QUOTIENT_SUM_LANES), 0, bk->stream, a);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_fr_quotient_out<Fr>, dim3((n + 255) / 256), dim3(256), 0, bk->stream, a);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(y_host, a.y, sizeof(Fr), hipMemcpyDeviceToHost, bk->stream));
    HIP_CHECK(hipStreamSynchronize(bk->stream));
    return 0;
  }
  static const CurveOps* ops() {
    static const CurveOps o = {C::ID, sizeof(Affine<F>), create, destroy, submit, finish, submit_host, bases_prepare, submit_bases, gen_points, field_op, ec_sum_affine, sum_reduce, batch_affine, sizeof(F), subgroup_check, table_prepare, fr_quotient, sizeof(typename C::Fr), subgroup_check_host, submit_host_bases};
    return &o;
  }
};

}  // namespace ctt
