// tests/emu/msm_emu.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never loaded by constantine_amd).
//
// CPU emulator of the HIP pipeline: runs the *same* per-thread bodies (constantine_amd/csrc/msm_bodies.h)
// and the same host orchestration (msm_pipeline.h) with every "kernel launch" replaced by a loop over
// (window, thread).  It exists so that the kernel logic (segmented accumulation, partial merging,
// recursive bucket reduction, Booth digits, plan selection) can be checked against the oracle in the
// CPU-only container; the LDS counting sort is emulated slice by slice with the same offset scheme.
// Build: g++ -O2 -std=c++17 -shared -fPIC -I constantine_amd/csrc tests/emu/msm_emu.cpp -o tests/emu/libmsm_emu.so
#include <algorithm>

#include "msm_pipeline.h"

using namespace ctt;

struct EmuBackend {
  // EMU_ALLOC_LIMIT (bytes): larger requests fail the way hipMalloc does on a full device (tests of the fallbacks)
  void* alloc(size_t b) {
    const char* lim = getenv("EMU_ALLOC_LIMIT");
    if (lim && b > (size_t)strtoull(lim, nullptr, 10)) throw OutOfDeviceMemory{b};
    return malloc(b);
  }
  void free(void* p) { ::free(p); }
  void free_quiet(void* p) noexcept { ::free(p); }
  void free_host_quiet(void* p) noexcept { ::free(p); }
  static bool partitioned() { return false; }
  void merge_mark() {}
  void memset0(void* p, size_t b) { memset(p, 0, b); }
  void d2d_async(void* dst, const void* src, size_t b) { memcpy(dst, src, b); }
  void* alloc_host(size_t b) { return malloc(b); }
  void free_host(void* p) { ::free(p); }
  void d2h_async(int, void* dst, const void* src, size_t b) { memcpy(dst, src, b); }
  void d2h_wait(int) {}
  void tail_begin() {}
  void tail_end() {}
  void tail_wait() {}
  bool tail_forked() const { return false; }
  void wide_mark() {}
  void wide_wait() {}
  void front_begin() {}
  void front_end() {}
  void front_abort() {}
  void accum_mark(bool) {}
  void stage_chunk(int) {}
  void d2h_sync(void* dst, const void* src, size_t b) { memcpy(dst, src, b); }
  void h2d(void* dst, const void* src, size_t b) { memcpy(dst, src, b); }
  void h2d_done() {}
  static constexpr bool THREADED_UPLOAD = true;   // (the slices of submit_host are copied by a thread here too: same code path)
  void uploader_begin() {}
  struct GuardScope {};
  [[noreturn]] void uploader_failed() { abort(); }
  void h2d_slice_done(uint32_t) {}
  void h2d_slice_wait(uint32_t) {}
  void h2d_coefs_done(uint32_t) {}
  void h2d_coefs_wait(uint32_t) {}
  void launch_iota(uint32_t* entries, uint32_t n, uint32_t* bstart, uint32_t* maxcount) {
    for (uint32_t j = 0; j < (n ? n : 1); j++) iota_body(entries, n, bstart, maxcount, j);
  }
  void stage_begin(int, int) {}
  void stage_end(int, int) {}

  template <class Fr>
  void launch_fr_from_mont(const uint32_t* in, uint32_t* out, uint32_t n) {
    for (uint32_t j = 0; j < n; j++) fr_from_mont_body<Fr>(in, out, n, j);
  }
  template <class F>
  void launch_table_next(const Affine<F>* prev, Affine<F>* next, uint32_t n, int c) {
    for (uint32_t j = 0; j < n; j++) table_next_body<F>(prev, next, n, c, j);
  }
  void sync() {}
  template <class F, class FD>
  void launch_convert(const Affine<F>* in, void* out, uint32_t n) {
    for (uint32_t j = 0; j < n; j++) convert_point_body<F, FD>(in, out, n, j);
  }
  // digits + sort: only the output contract of SortArgs is emulated (the GPU's two-pass LDS sort is exercised by
  // the GPU parity tests); plain counting sort per window
  void launch_digits_sort(const SortArgs& a) {
    const uint32_t n = a.n, B = a.B;
    // the sort clears the empty buckets (SortArgs::zero_base); the others are poisoned here, so that a bucket the accumulation
    // and the merge fail to write cannot pass for the neutral element
    if (a.zero_bytes) memset(a.zero_base, 0xA5, (size_t)a.W * B * a.zero_bytes);
    struct ZeroEmpty {
      const SortArgs& a;
      ~ZeroEmpty() {
        if (!a.zero_bytes) return;
        for (uint32_t w = 0; w < a.W; w++)
          for (uint32_t b = 0; b < a.B; b++) {
            const uint32_t* bs = a.bstart + (size_t)w * (a.B + 1);
            if (bs[b + 1] == bs[b]) memset((char*)a.zero_base + ((size_t)w * a.B + b) * a.zero_bytes, 0, a.zero_bytes);
          }
      }
    } zero_empty{a};
    std::vector<uint32_t> dg(n), cnt(B);
    for (int q = 0; q < 4; q++) a.maxcount[q] = 0;   // (the sort's first kernel zeroes the MSM's counters: SortArgs::maxcount)
    if (a.merged) {
      // window table: every digit window goes to the one bucket set, entry = table row w*id_stride + j
      std::vector<uint32_t> all((size_t)a.Wd * n);
      std::fill(cnt.begin(), cnt.end(), 0u);
      // digits through the register walker the GPU's partition kernels use (for_each_digit), checked against the plain form
      std::fill(all.begin(), all.end(), DIGIT_NONE);
      for (uint32_t j = 0; j < n; j++) {
        uint32_t k[1][8];
        for (int q = 0; q < 8; q++) k[0][q] = a.scalars[8ull * j + q];
        for_each_digit<1>(k, 0, a.Wd, a.lay, [&](uint32_t w, const uint32_t (&dd)[1]) {
          const uint32_t d = dd[0];
          if (d != booth_digit_packed(a.scalars + 8ull * j, (int)w, a.lay)) abort();
          all[(size_t)w * n + j] = d;
          if (d != DIGIT_NONE) cnt[d >> 1]++;
        });
      }
      uint32_t run = 0;
      for (uint32_t b = 0; b < B; b++) {
        a.bstart[b] = run;
        run += cnt[b];
        *a.maxcount = std::max(*a.maxcount, cnt[b]);
      }
      a.bstart[B] = run;
      std::vector<uint32_t> cur(a.bstart, a.bstart + B);
      for (uint32_t w = 0; w < a.Wd; w++)
        for (uint32_t j = 0; j < n; j++) {
          const uint32_t d = all[(size_t)w * n + j];
          if (d != DIGIT_NONE) a.entries[cur[d >> 1]++] = (w * a.id_stride + j) | ((d & 1u) << 31);
        }
      return;
    }
    for (uint32_t w = 0; w < a.W; w++) {
      std::fill(cnt.begin(), cnt.end(), 0u);
      for (uint32_t j = 0; j < n; j++) {
        dg[j] = DIGIT_NONE;
        uint32_t k[1][8];
        for (int q = 0; q < 8; q++) k[0][q] = a.scalars[8ull * j + q];
        for_each_digit<1>(k, w, 1, a.lay, [&](uint32_t, const uint32_t (&dd)[1]) { dg[j] = dd[0]; });   // the GPU kernels' digit walker
        if (dg[j] != booth_digit_packed(a.scalars + 8ull * j, (int)w, a.lay)) abort();
        if (dg[j] != DIGIT_NONE) cnt[dg[j] >> 1]++;
      }
      uint32_t* bs = a.bstart + (size_t)w * (B + 1);
      uint32_t run = 0;
      for (uint32_t b = 0; b < B; b++) {
        bs[b] = run;
        run += cnt[b];
        *a.maxcount = std::max(*a.maxcount, cnt[b]);
      }
      bs[B] = run;
      std::vector<uint32_t> cur(bs, bs + B);
      for (uint32_t j = 0; j < n; j++)
        if (dg[j] != DIGIT_NONE) a.entries[(size_t)w * n + cur[dg[j] >> 1]++] = j | ((dg[j] & 1u) << 31);
    }
  }
  template <class F>
  void launch_accum(const AccumArgs<F>& a, uint32_t W, bool into = false) {
    for (uint32_t w = 0; w < W; w++)
      for (uint32_t g = 0; g < a.G; g++) {
        if (into) accum_body<F, true>(a, w, g); else accum_body<F, false>(a, w, g);
      }
  }
  template <class F>
  void launch_merge_tail(const MergeArgs<F>& a, uint32_t W) {
    for (uint32_t w = 0; w < W; w++) for (uint32_t g = 0; g < a.G; g++) merge_tail_body<F>(a, w, g);
  }
  template <class F>
  void launch_merge_step(const MergeArgs<F>& a, uint32_t W, uint32_t d) {
    for (uint32_t w = 0; w < W; w++) for (uint32_t g = 0; g < a.G; g++) merge_step_body<F>(a, w, g, d);
  }
  template <class F>
  void launch_merge_finish(const MergeArgs<F>& a, uint32_t W, uint32_t first_d) {
    // one "workgroup" of one lane per window: the lane strides over all g, the barrier is a no-op; then the wide final
    for (uint32_t w = 0; w < W; w++) merge_finish_body<F>(a, w, first_d, 0, 1, []() {});
    if (merge_chain_bound<F>(a) > 1)
      for (uint32_t w = 0; w < W; w++) for (uint32_t g = 0; g < a.G; g++) merge_final_body<F>(a, w, g);
  }
  template <class F>
  void launch_merge_tail_queue(const MergeArgs<F>& a, uint32_t W) {
    for (uint32_t w = 0; w < W; w++)
      for (uint32_t g = 0; g < a.G; g++) {
        uint32_t item = 0;
        if (merge_tail_queue_body<F>(a, w, g, &item)) {
          if (*a.qcount >= merge_queue_capacity(W, a.G)) abort();   // the queue's capacity (launch_merge_queue's grid)
          a.queue[(*a.qcount)++] = item;
        }
      }
  }
  template <class F>
  void launch_merge_queue(const MergeArgs<F>& a, uint32_t W, uint32_t lmax, bool /*quad: a GPU launch shape*/) {
    for (uint32_t qi = 0; qi < merge_queue_capacity(W, a.G); qi++) merge_queue_body<F>(a, qi, lmax);
  }
  template <class F>
  void launch_merge_long(const MergeArgs<F>& a, uint32_t W, uint32_t lmax) {
    // one "workgroup" of one lane per window (the barrier is a no-op), as for merge_finish_body
    for (uint32_t w = 0; w < W; w++) merge_long_body<F>(a, w, lmax, 0, 1, []() {});
  }
  static bool pyr_goes_to_tail(uint32_t, uint32_t) { return false; }
  template <class F>
  void launch_window_groups(const XYZZ<F>* out, XYZZ<F>* wsum, uint32_t W, int c, int h, int ngrp) {
    for (uint32_t w = 0; w < W; w++)
      for (int g = 0; g < ngrp; g++) wsum[(size_t)w * ngrp + g] = window_group_sum_body<F>(out + (size_t)w * c, c, h, g);
  }
  void narrow_priority(bool) {}   // (wave priority of the narrow passes: nothing to emulate)
  template <class F>
  void launch_pyr(const PyrArgs<F>& a, uint32_t W, uint32_t ntasks) {
    // a pass reads only what earlier passes wrote, except the in-place halving q[t] += q[t+n] (disjoint t)
    for (uint32_t w = 0; w < W; w++) for (uint32_t t = 0; t < ntasks; t++) pyr_body<F>(a, w, t);
  }
};

// ---------------------------------------------------------------------------------------------
// Per-curve operation table.  This file is compiled once per curve (-DEMU_CURVE=<id>) and once without
// (the dispatcher), so the build parallelises like the HIP library's.
// ---------------------------------------------------------------------------------------------
struct EmuOps {
  int (*msm)(int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t n, int c, int K,
             int S, int* plan_out);
  // host-pointer form: the inputs are uploaded in `chunks` slices, one bucket set per slice (MsmEngine::submit_host)
  int (*msm_host)(int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t n, int c, int chunks);
  // cached bases with a window table over `ntab` points (MsmEngine::prepare_table), MSM over the first n; returns the c used
  // (chunks > 0: the coefficients are host-resident and go up in that many slices -- MsmEngine::submit_host with cached bases)
  int (*msm_table)(int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t ntab, size_t n, int c,
                   int K, int chunks);
  void (*gen)(uint64_t seed, uint64_t first, uint32_t n, void* out);
  void (*fop)(int op, const void* a, const void* b, void* r);
  int (*fop_dev)(int op, const void* a, const void* b, void* r);
  int (*dev_info)(int* lb, int* nl);
  int (*sum_reduce)(int out_kind, void* r, const void* points, size_t n, int K);
  void (*batch_affine)(int src_kind, void* dst, const void* src, size_t n, int K);
  // ticket order: submit A, then two complete MSMs B and C while A is outstanding, then finish A; r3 = 3 affine results;
  // returns the number of submits that were refused (0 expected)
  int (*msm_slots)(void* r3, const void* coefs, const void* points, size_t n);
  // KZG quotient over the curve's scalar field (msm_bodies.h FrQuotientArgs): poly canonical, dom / z / scale Montgomery
  void (*fr_quotient)(const void* poly, const void* dom, const void* z_mont, const void* scale_mont, uint32_t n, uint32_t K, void* q,
                      void* y);
};

#ifdef EMU_CURVE
#include "generators.h"

// test knobs of the reduction shape (tests/test_emu_pipeline.py sets them per case)
static void emu_env_options(MsmOptions& o) {
  const char* s;
  if ((s = getenv("EMU_HORNER_BITS"))) o.horner_bits = atoi(s);
  if ((s = getenv("EMU_HOST_WINDOW_SUMS"))) o.host_window_sums = atoi(s);
  if ((s = getenv("EMU_MERGE_CHAIN"))) o.merge_chain = atoi(s);
  if ((s = getenv("EMU_MERGE_LMAX"))) o.merge_lmax = atoi(s);
}

template <class C>
struct EmuCurve {
  using F = typename C::F;
  using FD = typename C::FD;
  static int msm(int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t n, int c, int K,
                 int S, int* plan_out) {
    EmuBackend bk;
    MsmEngine<C, EmuBackend> eng(bk);
    eng.opt.c = c;
    eng.opt.K = K;
    eng.opt.S = S;
    eng.opt.lanes = 4096;
    emu_env_options(eng.opt);
    // exercise both in-flight slots: submit twice, finish in order
    // K < 0 in the test harness means: go through the cached-base path (prepare_bases + submit against it)
    void* prepared = nullptr;
    if (K < 0) {
      eng.opt.K = -K;
      prepared = eng.prepare_bases((const Affine<F>*)points, (uint32_t)n);
    }
    int s0 = eng.submit((const uint32_t*)coefs, coef_is_fr != 0, prepared ? nullptr : (const Affine<F>*)points, (uint32_t)n,
                        prepared);
    auto res = eng.finish(s0);
    if (prepared) bk.free(prepared);
    write_result<typename MsmEngine<C, EmuBackend>::HF>(r, res, out_kind);
    if (plan_out && n) {
      plan_out[0] = eng.last_plan.c; plan_out[1] = eng.last_plan.W; plan_out[2] = (int)eng.last_plan.K;
      plan_out[3] = (int)eng.last_plan.G; plan_out[4] = (int)eng.last_plan.S;
    }
    return 0;
  }
  static int msm_host(int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t n, int c, int chunks) {
    EmuBackend bk;
    MsmEngine<C, EmuBackend> eng(bk);
    eng.opt.c = c;
    eng.opt.lanes = 4096;
    emu_env_options(eng.opt);
    std::vector<unsigned char> sc(n * 32 + 64), sp(n * sizeof(Affine<F>) + 64);
    int s0 = eng.submit_host(coefs, coef_is_fr != 0, points, (uint32_t)n, sc.data(), sp.data(), chunks);
    auto res = eng.finish(s0);
    write_result<typename MsmEngine<C, EmuBackend>::HF>(r, res, out_kind);
    return (int)eng.last_chunks;
  }
  static int msm_table(int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t ntab, size_t n,
                       int c, int K, int chunks) {
    EmuBackend bk;
    MsmEngine<C, EmuBackend> eng(bk);
    eng.opt.K = K;
    eng.opt.lanes = 4096;
    emu_env_options(eng.opt);
    int cu = 0;
    void* tab = c < 0 ? nullptr : eng.prepare_table((const Affine<F>*)points, (uint32_t)ntab, c, &cu);   // (c < 0: plain cached records)
    if (!tab && ntab) tab = eng.prepare_bases((const Affine<F>*)points, (uint32_t)ntab);   // (what bases_create does: plain records)
    std::vector<unsigned char> sc(n * 32 + 64);
    int s0 = chunks > 0 ? eng.submit_host(coefs, coef_is_fr != 0, nullptr, (uint32_t)n, sc.data(), nullptr, chunks, tab, cu, (uint32_t)ntab)
                        : eng.submit((const uint32_t*)coefs, coef_is_fr != 0, nullptr, (uint32_t)n, tab, cu, (uint32_t)ntab);
    auto res = eng.finish(s0);
    if (tab) bk.free(tab);
    write_result<typename MsmEngine<C, EmuBackend>::HF>(r, res, out_kind);
    return cu;
  }
  static void gen(uint64_t seed, uint64_t first, uint32_t n, void* out) {
    Affine<F> G = generator<C>();
    for (uint32_t j = 0; j < n; j++) gen_point_body<F>(G, seed, first, n, (Affine<F>*)out, j);
  }
  // field-level probes: op 0 mul, 1 sqr, 2 add, 3 sub, 4 neg, 5 inv  (coordinate field, reference representation)
  static void fop(int op, const void* a, const void* b, void* r) {
    const F& x = *(const F*)a;
    const F& y = *(const F*)b;
    F& o = *(F*)r;
    switch (op) {
      case 0: o = F::mul(x, y); break;
      case 1: o = F::sqr(x); break;
      case 2: o = F::add(x, y); break;
      case 3: o = F::sub(x, y); break;
      case 4: o = F::neg(x); break;
      case 5: o = F::inv(x); break;
      case 6: if constexpr (!IsFp2<F>::value) o = F::inv_fermat(x); else o = F::inv(x); break;
    }
  }
  // device-field probe: inputs in the reference representation, output raw FD limbs (uint32[NL]); returns NL
  static int fop_dev(int op, const void* a, const void* b, void* r) {
    if constexpr (FD::UNSAT) {
      FD o = dev_field_probe<FD>(op, FD::from_sat(*(const F*)a), FD::from_sat(*(const F*)b));
      *(FD*)r = o;
      return FD::NL;
    }
    return 0;
  }
  static int dev_info(int* lb, int* nl) {
    if constexpr (FD::UNSAT) {
      *lb = FD::LB;
      *nl = FD::NL;
      return 1;
    }
    return 0;
  }
  static int sum_reduce(int out_kind, void* r, const void* points, size_t n, int K) {
    EmuBackend bk;
    MsmEngine<C, EmuBackend> eng(bk);
    eng.opt.K = K;
    eng.opt.lanes = 64;
    auto res = eng.sum_reduce((const Affine<F>*)points, (uint32_t)n);
    write_result<typename MsmEngine<C, EmuBackend>::HF>(r, res, out_kind);
    return (int)eng.last_sum_K;
  }
  static void batch_affine(int src_kind, void* dst, const void* src, size_t n, int K) {
    BatchAffineArgs<F> a{(const F*)src, (Affine<F>*)dst, (uint32_t)n, src_kind, (uint32_t)K};
    for (uint32_t lane = 0; (uint64_t)lane * K < n; lane++) batch_affine_body<F>(a, lane);
  }
  static int msm_slots(void* r3, const void* coefs, const void* points, size_t n) {
    EmuBackend bk;
    MsmEngine<C, EmuBackend> eng(bk);
    eng.opt.lanes = 4096;
    using HF = typename MsmEngine<C, EmuBackend>::HF;
    const size_t ab = sizeof(Affine<F>);
    int refused = 0;
    const int a = eng.submit((const uint32_t*)coefs, false, (const Affine<F>*)points, (uint32_t)n);
    const int b = eng.submit((const uint32_t*)coefs, false, (const Affine<F>*)points, (uint32_t)n);
    if (a < 0 || b < 0) return 100;
    write_result<HF>((char*)r3 + ab, eng.finish(b), OUT_AFF);
    const int c = eng.submit((const uint32_t*)coefs, false, (const Affine<F>*)points, (uint32_t)n);   // slot of B is free, A's is not
    if (c < 0) refused++; else write_result<HF>((char*)r3 + 2 * ab, eng.finish(c), OUT_AFF);
    const int d = eng.submit((const uint32_t*)coefs, false, (const Affine<F>*)points, (uint32_t)n);
    const int e = d >= 0 ? eng.submit((const uint32_t*)coefs, false, (const Affine<F>*)points, (uint32_t)n) : -1;
    const int f = e >= 0 ? eng.submit((const uint32_t*)coefs, false, (const Affine<F>*)points, (uint32_t)n) : -1;
    if (d < 0) refused++;
    if (e < 0) refused++;        // three slots (MsmEngine::NSLOT): A, D and E may be outstanding together
    if (f >= 0) refused += 10;   // ... and a fourth ticket must be refused
    if (e >= 0) eng.finish(e);   // (out of submission order: tickets may be finished in any order)
    if (d >= 0) eng.finish(d);
    write_result<HF>(r3, eng.finish(a), OUT_AFF);
    return refused;
  }
  static void fr_quotient(const void* poly, const void* dom, const void* z_mont, const void* scale_mont, uint32_t n, uint32_t K,
                          void* q, void* y) {
    using Fr = typename C::Fr;
    const uint32_t lanes = (n + K - 1) / K;
    std::vector<uint32_t> inv((size_t)n * Fr::N), partial((size_t)lanes * Fr::N), tsum((size_t)FR_QUOTIENT_SUM_LANES * Fr::N);
    FrQuotientArgs<Fr> a;
    a.poly = (const uint32_t*)poly;
    a.dom = (const uint32_t*)dom;
    memcpy(a.z.l, z_mont, sizeof(a.z.l));
    memcpy(a.scale.l, scale_mont, sizeof(a.scale.l));
    a.n = n;
    a.K = K;
    a.inv = inv.data();
    a.partial = partial.data();
    a.tsum = tsum.data();
    a.q = (uint32_t*)q;
    a.y = (uint32_t*)y;
    for (uint32_t l = 0; l < lanes; l++) fr_quotient_inv_body<Fr>(a, l);
    for (uint32_t t = 0; t < FR_QUOTIENT_SUM_LANES; t++) fr_quotient_sum_body<Fr>(a, t, FR_QUOTIENT_SUM_LANES);
    fr_quotient_y_body<Fr>(a, FR_QUOTIENT_SUM_LANES);
    for (uint32_t i = 0; i < n; i++) fr_quotient_out_body<Fr>(a, i);
  }
  static const EmuOps* ops() {
    static const EmuOps o = {msm, msm_host, msm_table, gen, fop, fop_dev, dev_info, sum_reduce, batch_affine, msm_slots, fr_quotient};
    return &o;
  }
};

#if EMU_CURVE == 0
extern "C" const EmuOps* emu_ops_0() { return EmuCurve<Bls12381G1>::ops(); }
#elif EMU_CURVE == 1
extern "C" const EmuOps* emu_ops_1() { return EmuCurve<Bls12381G2>::ops(); }
#elif EMU_CURVE == 2
extern "C" const EmuOps* emu_ops_2() { return EmuCurve<Bn254G1>::ops(); }
#elif EMU_CURVE == 3
extern "C" const EmuOps* emu_ops_3() { return EmuCurve<Bn254G2>::ops(); }
#elif EMU_CURVE == 4
extern "C" const EmuOps* emu_ops_4() { return EmuCurve<PallasEc>::ops(); }
#elif EMU_CURVE == 5
extern "C" const EmuOps* emu_ops_5() { return EmuCurve<VestaEc>::ops(); }
#endif

#else  // dispatcher

extern "C" {
const EmuOps* emu_ops_0();
const EmuOps* emu_ops_1();
const EmuOps* emu_ops_2();
const EmuOps* emu_ops_3();
const EmuOps* emu_ops_4();
const EmuOps* emu_ops_5();

static const EmuOps* ops_of(int curve) {
  switch (curve) {
    case 0: return emu_ops_0();
    case 1: return emu_ops_1();
    case 2: return emu_ops_2();
    case 3: return emu_ops_3();
    case 4: return emu_ops_4();
    case 5: return emu_ops_5();
  }
  return nullptr;
}

int emu_msm(int curve, int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t n, int c,
            int K, int S, int* plan_out) {
  const EmuOps* o = ops_of(curve);
  return o ? o->msm(coef_is_fr, out_kind, r, coefs, points, n, c, K, S, plan_out) : -1;
}
int emu_msm_host(int curve, int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t n, int c,
                 int chunks) {
  const EmuOps* o = ops_of(curve);
  return o ? o->msm_host(coef_is_fr, out_kind, r, coefs, points, n, c, chunks) : -1;
}
// the plan the engine would make (msm_pipeline.h make_plan / make_table_plan): c, W, Wd, B, K, G, S, slice, NG, gshift, nent, cb, r, merge steps
int emu_plan(uint32_t n, int bits, uint32_t lanes, int table_c, uint32_t ntab, uint32_t* out) {
  MsmOptions o;
  o.lanes = lanes;
  const MsmPlan p = table_c > 0 ? make_table_plan(n, bits, table_c, ntab, o) : make_plan(n, bits, o);
  out[0] = (uint32_t)p.c; out[1] = (uint32_t)p.W; out[2] = (uint32_t)p.Wd; out[3] = p.B; out[4] = p.K; out[5] = p.G;
  out[6] = p.S; out[7] = p.slice; out[8] = p.NG; out[9] = p.gshift; out[10] = p.nent;
  out[11] = (uint32_t)p.lay.cb; out[12] = (uint32_t)p.lay.r; out[13] = (uint32_t)p.merge_steps;
  return 0;
}
int emu_table_window_bits(uint32_t ntab, int bits) { return choose_table_window_bits(ntab, bits); }
int emu_msm_table(int curve, int coef_is_fr, int out_kind, void* r, const void* coefs, const void* points, size_t ntab, size_t n,
                  int c, int K, int chunks) {
  const EmuOps* o = ops_of(curve);
  return o ? o->msm_table(coef_is_fr, out_kind, r, coefs, points, ntab, n, c, K, chunks) : -1;
}
int emu_gen_points(int curve, uint64_t seed, uint64_t first, uint32_t n, void* out) {
  const EmuOps* o = ops_of(curve);
  if (!o) return -1;
  o->gen(seed, first, n, out);
  return 0;
}
int emu_field_op(int curve, int op, const void* a, const void* b, void* r) {
  const EmuOps* o = ops_of(curve);
  if (!o) return -1;
  o->fop(op, a, b, r);
  return 0;
}
int emu_field_op_dev(int curve, int op, const void* a, const void* b, void* r) {
  const EmuOps* o = ops_of(curve);
  return o ? o->fop_dev(op, a, b, r) : 0;
}
int emu_sum_reduce(int curve, int out_kind, void* r, const void* points, size_t n, int K) {
  const EmuOps* o = ops_of(curve);
  return o ? o->sum_reduce(out_kind, r, points, n, K) : -1;
}
int emu_batch_affine(int curve, int src_kind, void* dst, const void* src, size_t n, int K) {
  const EmuOps* o = ops_of(curve);
  if (!o) return -1;
  o->batch_affine(src_kind, dst, src, n, K);
  return 0;
}
int emu_msm_slots(int curve, void* r3, const void* coefs, const void* points, size_t n) {
  const EmuOps* o = ops_of(curve);
  return o ? o->msm_slots(r3, coefs, points, n) : -1;
}
int emu_fr_quotient(int curve, const void* poly, const void* dom, const void* z_mont, const void* scale_mont, uint32_t n, uint32_t K,
                    void* q, void* y) {
  const EmuOps* o = ops_of(curve);
  if (!o) return -1;
  o->fr_quotient(poly, dom, z_mont, scale_mont, n, K, q, y);
  return 0;
}
int emu_dev_field_info(int curve, int* lb, int* nl) {
  const EmuOps* o = ops_of(curve);
  return o ? o->dev_info(lb, nl) : 0;
}
}
#endif
