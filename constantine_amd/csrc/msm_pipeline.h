// msm_pipeline.h -- host orchestration of one MSM over a Backend (HIP on the GPU; a CPU emulator in
// tests/emu that executes the same per-thread bodies).  See msm_bodies.h for the stage list.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "msm_bodies.h"

namespace ctt {

struct MsmPlan {
  uint32_t n;
  int c, W;
  uint32_t B;      // buckets per window = 2^(c-1)
  uint32_t S;      // sort slices per window
  uint32_t slice;  // scalars per slice
  uint32_t K;      // sorted entries per accumulate lane
  uint32_t G;      // accumulate lanes per window
  uint32_t rs;     // bucket-reduce chunk (power of two)
  uint32_t rlog;
};

struct MsmOptions {
  int c = 0;          // window bits (0 = choose)
  int K = 0;          // entries per lane (0 = choose from resident lanes)
  int rs_log = 3;     // reduce chunk = 2^rs_log
  int S = 0;          // sort slices (0 = choose)
  uint32_t lanes = 196608;  // resident lanes of the accumulate kernel (set by the backend)
};

// Window size for the GPU pipeline.  The reference's bestBucketBitSize
// (ec_multi_scalar_mul_scheduler.nim:172-223) models a CPU; any c yields the same group element,
// so the device uses its own cost model: W * (N mixed adds (10 mul) + 2^(c-1) buckets * ~2.3 full adds (14 mul)).
static inline int choose_window_bits(uint32_t n, int bits) {
  double best = 1e300;
  int bc = 2;
  for (int c = 2; c <= 16; c++) {
    double W = bits / c + 1;
    double cost = W * (10.0 * n + 32.0 * (double)(1u << (c - 1)) + 2000.0);
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

static inline MsmPlan make_plan(uint32_t n, int bits, const MsmOptions& o) {
  MsmPlan p;
  p.n = n;
  p.c = o.c > 0 ? o.c : choose_window_bits(n, bits);
  if (p.c < 2) p.c = 2;
  if (p.c > 16) p.c = 16;
  p.W = bits / p.c + 1;  // ec_multi_scalar_mul_parallel.nim:157-158: one more window when c | bits
  p.B = 1u << (p.c - 1);
  // sort slices: aim at >= 256 workgroups, slices of at least 4096 scalars
  uint32_t S = o.S > 0 ? (uint32_t)o.S : (256u + p.W - 1) / p.W;
  uint32_t maxS = (n + 4095u) / 4096u;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  p.S = S;
  p.slice = (n + S - 1) / S;
  // entries per lane: fill the resident lanes once
  uint64_t total = (uint64_t)p.W * n;
  uint32_t K = o.K > 0 ? (uint32_t)o.K : (uint32_t)((total + o.lanes - 1) / o.lanes);
  K = (K + 3u) & ~3u;
  if (K < 4) K = 4;
  p.K = K;
  p.G = (n + K - 1) / K;
  p.rlog = (uint32_t)o.rs_log;
  p.rs = 1u << p.rlog;
  return p;
}

// Horner over the window sums (ec_multi_scalar_mul.nim:250-254; _parallel.nim:199-203)
template <class F>
static inline XYZZ<F> combine_windows(const XYZZ<F>* sums, int W, int c) {
  XYZZ<F> r = sums[W - 1];
  for (int w = W - 2; w >= 0; w--) {
    for (int i = 0; i < c; i++) r = xyzz_dbl<F>(r);
    xyzz_add<F>(r, sums[w]);
  }
  return r;
}

// Stage indices for timings
enum { ST_DIGITS = 0, ST_SORT, ST_ACCUM, ST_MERGE, ST_REDUCE, ST_TOTAL, ST_COUNT };

template <class C, class BK>
struct MsmEngine {
  using F = typename C::F;
  BK& bk;
  MsmOptions opt;
  MsmPlan last_plan;

  // grow-only workspace
  struct Buf { void* p = nullptr; size_t cap = 0; };
  Buf digits, counts, bstart, entries, buckets, heads, tails, hkey, tkey, rA[2], rP[2], scal, maxcount;

  explicit MsmEngine(BK& b) : bk(b) {}
  ~MsmEngine() {
    Buf* all[] = {&digits, &counts, &bstart, &entries, &buckets, &heads, &tails, &hkey, &tkey, &rA[0], &rA[1], &rP[0], &rP[1], &scal, &maxcount};
    for (Buf* b : all) if (b->p) bk.free(b->p);
  }
  void* need(Buf& b, size_t bytes) {
    if (bytes > b.cap) {
      if (b.p) bk.free(b.p);
      size_t cap = bytes + bytes / 8 + 256;
      b.p = bk.alloc(cap);
      b.cap = cap;
    }
    return b.p;
  }

  // d_coefs: canonical scalars [n][8] (coef_is_fr = false) or Montgomery Fr elements (true), device memory.
  // d_points: affine Montgomery points, device memory.  Result: the MSM as an XYZZ point on the host.
  XYZZ<F> run(const uint32_t* d_coefs, bool coef_is_fr, const Affine<F>* d_points, uint32_t n) {
    if (n == 0) return XYZZ<F>::inf();  // len == 0 is UB upstream; we return the neutral
    MsmPlan p = make_plan(n, C::BITS, opt);
    last_plan = p;
    const uint32_t W = p.W, B = p.B;

    bk.stage_begin(ST_TOTAL);
    bk.stage_begin(ST_DIGITS);
    const uint32_t* d_scalars = d_coefs;
    if (coef_is_fr) {
      uint32_t* t = (uint32_t*)need(scal, (size_t)n * 32);
      bk.template launch_fr_from_mont<typename C::Fr>(d_coefs, t, n);
      d_scalars = t;
    }
    uint32_t* d_digits = (uint32_t*)need(digits, (size_t)W * n * 4);
    DigitsArgs da{d_scalars, d_digits, n, p.c, (int)W};
    bk.launch_digits(da);
    bk.stage_end(ST_DIGITS);

    bk.stage_begin(ST_SORT);
    uint32_t* d_counts = (uint32_t*)need(counts, (size_t)W * p.S * B * 4);
    uint32_t* d_bstart = (uint32_t*)need(bstart, (size_t)W * (B + 1) * 4);
    uint32_t* d_entries = (uint32_t*)need(entries, (size_t)W * n * 4);
    uint32_t* d_maxcount = (uint32_t*)need(maxcount, 256);
    bk.memset0(d_maxcount, 4);
    bk.launch_sort(d_digits, d_counts, d_bstart, d_entries, d_maxcount, n, B, p.S, p.slice, W);
    bk.stage_end(ST_SORT);

    bk.stage_begin(ST_ACCUM);
    XYZZ<F>* d_buckets = (XYZZ<F>*)need(buckets, (size_t)W * B * sizeof(XYZZ<F>));
    bk.memset0(d_buckets, (size_t)W * B * sizeof(XYZZ<F>));
    XYZZ<F>* d_heads = (XYZZ<F>*)need(heads, (size_t)W * p.G * sizeof(XYZZ<F>));
    XYZZ<F>* d_tails = (XYZZ<F>*)need(tails, (size_t)W * p.G * sizeof(XYZZ<F>));
    uint32_t* d_hkey = (uint32_t*)need(hkey, (size_t)W * p.G * 4);
    uint32_t* d_tkey = (uint32_t*)need(tkey, (size_t)W * p.G * 4);
    AccumArgs<F> aa{d_entries, d_bstart, d_points, d_buckets, d_heads, d_tails, d_hkey, d_tkey, n, B, p.K, p.G};
    bk.template launch_accum<F>(aa, W);
    bk.stage_end(ST_ACCUM);

    bk.stage_begin(ST_MERGE);
    MergeArgs<F> ma{d_bstart, d_buckets, d_heads, d_tails, d_hkey, d_tkey, d_maxcount, B, p.K, p.G};
    bk.template launch_merge_tail<F>(ma, W);
    // tree steps: host bound on the chain length is ceil(n/K)+1; kernels exit early on the device bound
    for (uint32_t d = 1; d < p.G + 1; d <<= 1) bk.template launch_merge_step<F>(ma, W, d);
    bk.template launch_merge_final<F>(ma, W);
    bk.stage_end(ST_MERGE);

    bk.stage_begin(ST_REDUCE);
    uint32_t n_in = B;
    const XYZZ<F>* A_in = d_buckets;
    const XYZZ<F>* P_in = nullptr;
    uint32_t wbase = 1;
    int pp = 0;
    size_t lvl_cap = (size_t)W * ((B + p.rs - 1) / p.rs) * sizeof(XYZZ<F>);
    for (int i = 0; i < 2; i++) { need(rA[i], lvl_cap); need(rP[i], lvl_cap); }
    const XYZZ<F>* d_sums = nullptr;
    for (;;) {
      uint32_t n_out = (n_in + p.rs - 1) / p.rs;
      ReduceArgs<F> ra{A_in, P_in, (XYZZ<F>*)rA[pp].p, (XYZZ<F>*)rP[pp].p, n_in, n_out, p.rs, p.rlog, wbase};
      bk.template launch_reduce<F>(ra, W);
      A_in = (const XYZZ<F>*)rA[pp].p;
      P_in = (const XYZZ<F>*)rP[pp].p;
      n_in = n_out;
      wbase = 0;
      pp ^= 1;
      if (n_out == 1) { d_sums = P_in; break; }
    }
    bk.stage_end(ST_REDUCE);

    std::vector<XYZZ<F>> sums(W);
    bk.d2h(sums.data(), d_sums, (size_t)W * sizeof(XYZZ<F>));  // synchronises
    bk.stage_end(ST_TOTAL);
    return combine_windows<F>(sums.data(), (int)W, p.c);
  }
};

// ---------------------------------------------------------------------------------------------
// Output conversion to the caller's coordinate system (C API structs, include/constantine/curves/*.h)
// ---------------------------------------------------------------------------------------------
enum { OUT_AFF = 0, OUT_JAC = 1, OUT_PRJ = 2 };

// EC_ShortW_Jac neutral (1,1,0) jacobian.nim:58-64; EC_ShortW_Prj neutral (0,1,0) projective.nim:56-62.
// Any representative of the same group element is a valid result (the reference's own raw coordinates
// differ between its serial and parallel paths); we emit the canonical one with Z = 1.
template <class F>
static inline void write_result(void* r, const XYZZ<F>& res, int out_kind) {
  Affine<F> a = xyzz_to_affine<F>(res);
  F* o = (F*)r;
  if (out_kind == OUT_AFF) {
    o[0] = a.x;
    o[1] = a.y;
    return;
  }
  if (a.is_inf()) {
    o[0] = out_kind == OUT_JAC ? F::one() : F::zero();
    o[1] = F::one();
    o[2] = F::zero();
  } else {
    o[0] = a.x;
    o[1] = a.y;
    o[2] = F::one();
  }
}

}  // namespace ctt
