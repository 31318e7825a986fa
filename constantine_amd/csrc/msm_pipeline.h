// msm_pipeline.h -- host orchestration of one MSM over a Backend (HIP on the GPU; a CPU emulator in
// tests/emu that executes the same per-thread bodies).  See msm_bodies.h for the stage list.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "host_fp64.h"
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

// Final Horner (ec_multi_scalar_mul.nim:250-254; _parallel.nim:199-203), fused with the last step of the
// bucket reduction: per window the device returns O_0..O_{c-2} (sum of the buckets whose index has bit l set)
// and TOP (sum of all buckets); window sum = sum_l 2^l O_l + TOP, result = sum_w 2^(c*w) * window sum.
template <class F>
static inline XYZZ<F> combine_windows_bits(const XYZZ<F>* o, int W, int c) {
  XYZZ<F> r = XYZZ<F>::inf();
  for (int w = W - 1; w >= 0; w--) {
    const XYZZ<F>* ow = o + (size_t)w * c;
    for (int l = c - 1; l >= 0; l--) {
      r = xyzz_dbl<F>(r);
      if (l <= c - 2) xyzz_add<F>(r, ow[l]);
    }
    xyzz_add<F>(r, ow[c - 1]);
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
  // d_points: affine Montgomery points, device memory.  Result: the MSM as an XYZZ point on the host
  // (host arithmetic on 64-bit limbs, same memory layout as the device field).
  using HF = typename HostField<F>::type;
  XYZZ<HF> run(const uint32_t* d_coefs, bool coef_is_fr, const Affine<F>* d_points, uint32_t n) {
    static_assert(sizeof(XYZZ<HF>) == sizeof(XYZZ<F>), "host/device layouts must match");
    if (n == 0) return XYZZ<HF>::inf();  // len == 0 is UB upstream; we return the neutral
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
    bk.fetch_u32_async(d_maxcount);  // largest bucket: read back while the accumulation runs
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
    // tree steps over the head chain of a bucket: a bucket of m entries spans at most floor((m-1)/K)+1 heads
    const uint32_t mc = bk.fetch_u32_wait();
    const uint32_t chain = mc ? (mc - 1) / p.K + 1 : 0;
    for (uint32_t d = 1; d < chain; d <<= 1) bk.template launch_merge_step<F>(ma, W, d);
    bk.template launch_merge_final<F>(ma, W);
    bk.stage_end(ST_MERGE);

    bk.stage_begin(ST_REDUCE);
    XYZZ<F>* d_pyr = (XYZZ<F>*)need(rA[0], (size_t)W * B * sizeof(XYZZ<F>));
    XYZZ<F>* d_q = (XYZZ<F>*)need(rA[1], (size_t)W * (B / 2 + 1) * sizeof(XYZZ<F>));
    XYZZ<F>* d_out = (XYZZ<F>*)need(rP[0], (size_t)W * p.c * sizeof(XYZZ<F>));
    for (int pass = 0; pass <= p.c - 2; pass++) {
      PyrArgs<F> pa{d_buckets, d_pyr, d_q, d_out, B, p.c, pass};
      bk.template launch_pyr<F>(pa, W, pyr_pass_tasks(B, p.c, pass));
    }
    bk.stage_end(ST_REDUCE);

    std::vector<XYZZ<HF>> sums((size_t)W * p.c);
    bk.d2h(sums.data(), d_out, (size_t)W * p.c * sizeof(XYZZ<F>));  // synchronises
    bk.stage_end(ST_TOTAL);
    return combine_windows_bits<HF>(sums.data(), (int)W, p.c);
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
