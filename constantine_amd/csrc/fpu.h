// fpu.h -- carry-free ("unsaturated") prime field for the hot kernels: NL limbs of LB < 32 bits.
//
// Why: on gfx950 the carry instruction of a saturated 32-bit-limb multiplier (v_addc_co_u32) costs almost
// as much issue time as the multiplier itself (v_mad_u64_u32), see profiles/microbench_r01.jsonl.  With
// LB-bit limbs a whole product-scanning column fits a 64-bit accumulator, so the Montgomery multiplication
// (same algorithm as the reference's, limbs_montgomery.nim:268-310, radix 2^LB instead of 2^64) is nothing but
// v_mad_u64_u32, plus one mask/shift per column: 392 multiplies and no carries for BLS12-381 (LB = 28,
// NL = 14) against 288 multiplies + 288 carries.
//
// Representation: value v = sum l[i] * 2^(LB*i) with v == x * R' (mod p), R' = 2^(LB*NL).  Elements are NOT
// kept in [0,p): every operation returns limbs normalised to < 2^LB (top limb takes the excess) and a value
// bounded by a small multiple of p that the caller tracks statically:
//     mul, sqr   : result < 2p   provided  a*b < R'*p   (R'/p >= 2^RP_OVER_P_LOG2: 2^11 for BLS12-381, 2^7 for
//                                                        the 254/255-bit fields; operands up to ~11p are fine)
//     add        : bound(a) + bound(b)
//     sub<B>     : a - b + B*p, needs b < B*p;  result < bound(a) + B
// Conversions to/from the reference's representation (Montgomery R = 2^(64L), saturated limbs) happen once per
// point on the way in (from_sat) and once per returned point on the host (HostField, host_fp64.h).
#pragma once
#include "fp.h"

namespace ctt {

template <class UP>
struct FpU {
  static constexpr int HEADROOM_LOG2 = UP::RP_OVER_P_LOG2;  // log2(R'/p): a product of operands < k1*p, k2*p needs k1*k2 < R'/p
  using Params = UP;
  using Sat = Fp<typename UP::Sat>;
  static constexpr int NL = UP::NL;
  static constexpr int LB = UP::LB;
  static constexpr uint32_t MASK = UP::MASK;
  static constexpr int N = NL;        // limb count (for generic helpers)
  static constexpr int MULB = 2;      // mul/sqr outputs are < MULB * p
  static constexpr bool UNSAT = true;
  uint32_t l[NL];

  CTT_HD static FpU zero() {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
    return r;
  }
  CTT_HD static FpU one() {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = UP::ONE[i];
    return r;
  }
  // raw test: every limb zero.  Exact for values we set to zero ourselves (neutral flags, (0,0) inputs);
  // a value that is merely == 0 (mod p) needs is_zero_modp.
  CTT_HD bool is_zero() const {
    uint32_t a = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) a |= l[i];
    return a == 0;
  }
  CTT_HD static FpU select(bool c, const FpU& a, const FpU& b) {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
  }

  // limbs of k*p in canonical radix-2^LB form
  struct KP { uint32_t l[NL]; };
  static constexpr KP kp(int k) {
    KP r{};
    uint64_t c = 0;
    for (int i = 0; i < NL; i++) {
      uint64_t v = (uint64_t)UP::P[i] * (uint64_t)k + c;
      r.l[i] = (i == NL - 1) ? (uint32_t)v : (uint32_t)(v & MASK);
      c = v >> LB;
    }
    return r;
  }

  // value == 0 (mod p), for a normalised element known to be < B*p
  template <int B>
  CTT_HD bool is_zero_modp() const {
    // quick reject on the low limb: v = k*p  =>  l[0] == (k*p) mod 2^LB for some k < B
    bool maybe = false;
#pragma unroll
    for (int k = 0; k < B; k++) maybe |= (l[0] == kp(k).l[0]);
    if (!maybe) return false;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");  // keep the full comparison behind a real branch (hipcc otherwise speculates it)
#endif
    return is_multiple_of_p(*this, B);
  }
  // rare path, kept out of line: compare against k*p limb by limb
  static CTT_HD bool is_multiple_of_p(const FpU& a, int B) {
    bool hit = false;
    for (int k = 0; k < B; k++) {
      uint32_t d = 0;
      uint64_t c = 0;
      for (int i = 0; i < NL; i++) {
        uint64_t v = (uint64_t)UP::P[i] * (uint64_t)k + c;
        uint32_t li = (i == NL - 1) ? (uint32_t)v : (uint32_t)(v & MASK);
        c = v >> LB;
        d |= a.l[i] ^ li;
      }
      hit |= (d == 0);
    }
    return hit;
  }

  // carry propagation: limbs < 2^LB afterwards (the top limb keeps the excess)
  CTT_HD void normalise() {
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
      uint32_t c = l[i] >> LB;
      l[i] &= MASK;
      l[i + 1] += c;
    }
  }

  CTT_HD static FpU add(const FpU& a, const FpU& b) {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    r.normalise();
    return r;
  }
  CTT_HD static FpU dbl(const FpU& a) {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] << 1;
    r.normalise();
    return r;
  }

  // a - b + B*p  (b normalised and < B*p).  B*p is spread over the limbs as
  // (c0 + 2^LB, c1 + 2^LB - 1, ..., c_top - 1) so that no limb goes negative before the carries are propagated.
  template <int B>
  CTT_HD static FpU sub(const FpU& a, const FpU& b) {
    constexpr KP c = kp(B);
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t bias = (i == 0) ? c.l[0] + (1u << LB) : (i == NL - 1) ? c.l[i] - 1u : c.l[i] + (1u << LB) - 1u;
      r.l[i] = a.l[i] + bias - b.l[i];
    }
    r.normalise();
    return r;
  }
  template <int B>
  CTT_HD static FpU neg(const FpU& a) { return sub<B>(zero(), a); }
  // conditional negation; the negated branch is B*p - a (a < B*p).  A raw zero stays a raw zero.
  template <int B>
  CTT_HD static FpU cneg(const FpU& a, bool c) {
    FpU n = sub<B>(zero(), a);
    return select(c, n, a);
  }

  // Montgomery product a*b/R' (mod p), radix 2^LB, product scanning; limbs of a, b < 2^30, a*b < R'*p.
  CTT_HD static FpU mul(const FpU& a, const FpU& b) {
    uint64_t acc = 0;
    uint32_t m[NL];
    FpU t;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i < k; i++)
        if (UP::P[k - i] != 0u) acc += (uint64_t)m[i] * UP::P[k - i];
      m[k] = ((uint32_t)acc * UP::M0INV) & MASK;
      acc += (uint64_t)m[k] * UP::P[0];
      acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
      for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - NL + 1; i < NL; i++)
        if (UP::P[k - i] != 0u) acc += (uint64_t)m[i] * UP::P[k - i];
      t.l[k - NL] = (uint32_t)acc & MASK;
      acc >>= LB;
    }
    t.l[NL - 1] = (uint32_t)acc;
    return t;
  }

  // (a*b + c*d)/R' (mod p): two products, one Montgomery reduction.  Needs a*b + c*d < R'*p.
  // The two products run in separate column accumulators (two dependency chains, see mul_pair).
  CTT_HD static FpU mul2(const FpU& a, const FpU& b, const FpU& c, const FpU& d) {
    uint64_t acc = 0;
    uint32_t m[NL];
    FpU t;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      uint64_t s2 = 0;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (j >= 0 && j < NL) {
          acc += (uint64_t)a.l[i] * b.l[j];
          s2 += (uint64_t)c.l[i] * d.l[j];
        }
      }
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (j >= 1 && j < NL && i < (k < NL ? k : NL) && UP::P[j] != 0u) s2 += (uint64_t)m[i] * UP::P[j];
      }
      acc += s2;
      if (k < NL) {
        m[k] = ((uint32_t)acc * UP::M0INV) & MASK;
        acc += (uint64_t)m[k] * UP::P[0];
      } else {
        t.l[k - NL] = (uint32_t)acc & MASK;
      }
      acc >>= LB;
    }
    t.l[NL - 1] = (uint32_t)acc;
    return t;
  }

  // Two independent products written column by column side by side: a single product is one long dependency
  // chain through its column accumulator; a second, independent chain gives the in-order issue logic something to
  // do while a v_mad_u64_u32 result is in flight.  hipcc is left to schedule the two chains (pinning the order
  // with sched_barrier or two-instruction asm statements measured slower: 2.66 vs 2.52 ms on the accumulate kernel).
  CTT_HD static void mul_pair(const FpU& a, const FpU& b, const FpU& c, const FpU& d, FpU& r1, FpU& r2) {
    uint64_t acc1 = 0, acc2 = 0;
    uint32_t m1[NL], m2[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc1 += (uint64_t)a.l[i] * b.l[k - i];
        acc2 += (uint64_t)c.l[i] * d.l[k - i];
      }
#pragma unroll
      for (int i = 0; i < k; i++)
        if (UP::P[k - i] != 0u) {
          acc1 += (uint64_t)m1[i] * UP::P[k - i];
          acc2 += (uint64_t)m2[i] * UP::P[k - i];
        }
      m1[k] = ((uint32_t)acc1 * UP::M0INV) & MASK;
      m2[k] = ((uint32_t)acc2 * UP::M0INV) & MASK;
      acc1 += (uint64_t)m1[k] * UP::P[0];
      acc2 += (uint64_t)m2[k] * UP::P[0];
      acc1 >>= LB;
      acc2 >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
      for (int i = k - NL + 1; i < NL; i++) {
        acc1 += (uint64_t)a.l[i] * b.l[k - i];
        acc2 += (uint64_t)c.l[i] * d.l[k - i];
      }
#pragma unroll
      for (int i = k - NL + 1; i < NL; i++)
        if (UP::P[k - i] != 0u) {
          acc1 += (uint64_t)m1[i] * UP::P[k - i];
          acc2 += (uint64_t)m2[i] * UP::P[k - i];
        }
      r1.l[k - NL] = (uint32_t)acc1 & MASK;
      r2.l[k - NL] = (uint32_t)acc2 & MASK;
      acc1 >>= LB;
      acc2 >>= LB;
    }
    r1.l[NL - 1] = (uint32_t)acc1;
    r2.l[NL - 1] = (uint32_t)acc2;
  }

  // two independent squares, interleaved
  CTT_HD static void sqr_pair(const FpU& a, const FpU& c, FpU& r1, FpU& r2) {
    uint64_t acc1 = 0, acc2 = 0;
    uint32_t m1[NL], m2[NL], a2[NL], c2[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
      a2[i] = a.l[i] << 1;
      c2[i] = c.l[i] << 1;
    }
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      {
        const int lo = k - NL + 1 > 0 ? k - NL + 1 : 0;
        const int hi = (k + 1) >> 1;  // i < j  <=>  i < (k+1)/2
#pragma unroll
        for (int i = lo; i < hi; i++) {
          acc1 += (uint64_t)a2[i] * a.l[k - i];
          acc2 += (uint64_t)c2[i] * c.l[k - i];
        }
      }
      if ((k & 1) == 0) {
        acc1 += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
        acc2 += (uint64_t)c.l[k >> 1] * c.l[k >> 1];
      }
      if (k < NL) {
#pragma unroll
        for (int i = 0; i < k; i++)
          if (UP::P[k - i] != 0u) {
            acc1 += (uint64_t)m1[i] * UP::P[k - i];
            acc2 += (uint64_t)m2[i] * UP::P[k - i];
          }
        m1[k] = ((uint32_t)acc1 * UP::M0INV) & MASK;
        m2[k] = ((uint32_t)acc2 * UP::M0INV) & MASK;
        acc1 += (uint64_t)m1[k] * UP::P[0];
        acc2 += (uint64_t)m2[k] * UP::P[0];
      } else {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++)
          if (UP::P[k - i] != 0u) {
            acc1 += (uint64_t)m1[i] * UP::P[k - i];
            acc2 += (uint64_t)m2[i] * UP::P[k - i];
          }
        r1.l[k - NL] = (uint32_t)acc1 & MASK;
        r2.l[k - NL] = (uint32_t)acc2 & MASK;
      }
      acc1 >>= LB;
      acc2 >>= LB;
    }
    r1.l[NL - 1] = (uint32_t)acc1;
    r2.l[NL - 1] = (uint32_t)acc2;
  }

  // square: cross products once with a doubled operand (2*a_i < 2^31 fits)
  CTT_HD static FpU sqr(const FpU& a) {
    uint64_t acc = 0;
    uint32_t m[NL];
    uint32_t a2[NL];
    FpU t;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      {
        const int lo = k - NL + 1 > 0 ? k - NL + 1 : 0;
        const int hi = (k + 1) >> 1;  // i < j  <=>  i < (k+1)/2
#pragma unroll
        for (int i = lo; i < hi; i++) acc += (uint64_t)a2[i] * a.l[k - i];
      }
      if ((k & 1) == 0) acc += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
      if (k < NL) {
#pragma unroll
        for (int i = 0; i < k; i++)
          if (UP::P[k - i] != 0u) acc += (uint64_t)m[i] * UP::P[k - i];
        m[k] = ((uint32_t)acc * UP::M0INV) & MASK;
        acc += (uint64_t)m[k] * UP::P[0];
      } else {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++)
          if (UP::P[k - i] != 0u) acc += (uint64_t)m[i] * UP::P[k - i];
        t.l[k - NL] = (uint32_t)acc & MASK;
      }
      acc >>= LB;
    }
    t.l[NL - 1] = (uint32_t)acc;
    return t;
  }

  // ---- conversions --------------------------------------------------------------------------------
  // bits [pos, pos+LB) of a little-endian 32-bit limb array of n words
  CTT_HD static uint32_t bits_at(const uint32_t* w, int n, int pos) {
    const int word = pos >> 5, sh = pos & 31;
    uint32_t v = word < n ? w[word] >> sh : 0u;
    if (sh + LB > 32 && word + 1 < n) v |= w[word + 1] << (32 - sh);
    return v & MASK;
  }
  // reference representation (Montgomery R, saturated, < p)  ->  this one (Montgomery R'); (raw) zero stays zero
  CTT_HD static FpU from_sat(const Sat& s) {
    FpU r, c;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      r.l[i] = bits_at(s.l, Sat::N, LB * i);
      c.l[i] = UP::C_IN[i];
    }
    return mul(r, c);
  }
};

// ---------------------------------------------------------------------------------------------
// Fp2 = Fp[i]/(i^2+1) over the carry-free base field.  Every component is normalised and bounded like a base
// element (MULB = 2: products come out of ONE Montgomery reduction each, < 2p):
//   mul:  c0 = a0*b0 + (Kp - a1)*b1,  c1 = a0*b1 + a1*b0        (two sum-of-products, towers.nim:852-878 prod2x)
//   sqr:  c0 = (a0 + a1)*(a0 - a1 + Kp),  c1 = a0*a1 + a0*a1    (square_complex, towers.nim:758-796)
// K = 10 covers every operand the EC formulas produce (components < 5*MULB*p); needs 2*K^2 < R'/p, true for
// BLS12-381 (R'/p = 2^11), not for the 29x9 fields.
// ---------------------------------------------------------------------------------------------
template <class UP>
struct Fp2<FpU<UP>> {
  using F = FpU<UP>;
  using Base = F;
  static constexpr int MULB = 2;
  static constexpr bool UNSAT = true;
  static constexpr int KNEG = 10;
  static constexpr int HEADROOM_LOG2 = UP::RP_OVER_P_LOG2;  // log2(R'/p): operand bounds k1*k2 must stay below it
  static constexpr int LB = F::LB;
  static constexpr int NL = 2 * F::NL;   // limbs of the whole element (c0 then c1)
  static_assert(2 * KNEG * KNEG < (1 << UP::RP_OVER_P_LOG2), "Fp2 over this base field needs more Montgomery headroom");
  F c0, c1;

  CTT_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
  CTT_HD static Fp2 one() { return {F::one(), F::zero()}; }
  CTT_HD bool is_zero() const { return c0.is_zero() & c1.is_zero(); }
  template <int B>
  CTT_HD bool is_zero_modp() const {
    return c0.template is_zero_modp<B>() && c1.template is_zero_modp<B>();
  }
  CTT_HD static Fp2 select(bool c, const Fp2& a, const Fp2& b) {
    return {F::select(c, a.c0, b.c0), F::select(c, a.c1, b.c1)};
  }
  CTT_HD static Fp2 add(const Fp2& a, const Fp2& b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
  CTT_HD static Fp2 dbl(const Fp2& a) { return {F::dbl(a.c0), F::dbl(a.c1)}; }
  template <int B>
  CTT_HD static Fp2 sub(const Fp2& a, const Fp2& b) {
    return {F::template sub<B>(a.c0, b.c0), F::template sub<B>(a.c1, b.c1)};
  }
  template <int B>
  CTT_HD static Fp2 cneg(const Fp2& a, bool c) {
    return {F::template cneg<B>(a.c0, c), F::template cneg<B>(a.c1, c)};
  }
  // Karatsuba on unreduced columns (3 limb products, 2 reductions) was tried: the six live operands plus two sets
  // of Montgomery quotients spill the accumulate kernel to scratch (G2 accumulate 8.7 -> 126 ms); two sums of
  // products it is.
  CTT_HD static Fp2 mul(const Fp2& a, const Fp2& b) {
    F na1 = F::template sub<KNEG>(F::zero(), a.c1);
    return {F::mul2(a.c0, b.c0, na1, b.c1), F::mul2(a.c0, b.c1, a.c1, b.c0)};
  }
  CTT_HD static Fp2 sqr(const Fp2& a) {
    F s = F::add(a.c0, a.c1);
    F d = F::template sub<KNEG>(a.c0, a.c1);
    return {F::mul(s, d), F::mul(F::dbl(a.c0), a.c1)};   // (a0+a1)(a0-a1), 2 a0 a1: two products, two reductions
  }
  // a*b + c*d: two products (not fused further: four base sum-of-products)
  CTT_HD static Fp2 mul2(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) { return add(mul(a, b), mul(c, d)); }
  CTT_HD static void mul_pair(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d, Fp2& r1, Fp2& r2) {
    r1 = mul(a, b);
    r2 = mul(c, d);
  }
  CTT_HD static void sqr_pair(const Fp2& a, const Fp2& c, Fp2& r1, Fp2& r2) {
    r1 = sqr(a);
    r2 = sqr(c);
  }
  CTT_HD static Fp2 from_sat(const Fp2<typename F::Sat>& s) { return {F::from_sat(s.c0), F::from_sat(s.c1)}; }
};

// Field-generic spellings used by ec.h: saturated fields ignore the bias/bound parameters.
template <class F, int B> CTT_HD F fsub(const F& a, const F& b) {
  if constexpr (F::UNSAT) return F::template sub<B>(a, b); else return F::sub(a, b);
}
template <class F, int B> CTT_HD F fcneg(const F& a, bool c) {
  if constexpr (F::UNSAT) return F::template cneg<B>(a, c); else return F::cneg(a, c);
}
// a*b - c*d with one reduction where the field supports it; B bounds d's partner c (c < B*p)
template <class F> struct IsFp2 { static constexpr bool value = false; };
template <class B> struct IsFp2<Fp2<B>> { static constexpr bool value = true; };
template <class F, int B> CTT_HD F fmul_sub(const F& a, const F& b, const F& c, const F& d) {
  if constexpr (F::UNSAT && IsFp2<F>::value) {
    return F::template sub<F::MULB>(F::mul(a, b), F::mul(c, d));   // < 2*MULB*p, the bound the callers assume
  } else if constexpr (F::UNSAT) {
    F nc = F::template sub<B>(F::zero(), c);   // B*p - c  (< B*p), so a*b + nc*d == a*b - c*d (mod p)
    return F::mul2(a, b, nc, d);
  } else {
    return F::sub(F::mul(a, b), F::mul(c, d));
  }
}
// two independent products / squares: interleaved where the field supports it (fpu.h mul_pair)
template <class F> CTT_HD void fmul_pair(const F& a, const F& b, const F& c, const F& d, F& r1, F& r2) {
  if constexpr (F::UNSAT) {
    F::mul_pair(a, b, c, d, r1, r2);
  } else {
    r1 = F::mul(a, b);
    r2 = F::mul(c, d);
  }
}
template <class F> CTT_HD void fsqr_pair(const F& a, const F& c, F& r1, F& r2) {
  if constexpr (F::UNSAT) {
    F::sqr_pair(a, c, r1, r2);
  } else {
    r1 = F::sqr(a);
    r2 = F::sqr(c);
  }
}
template <class F, int B> CTT_HD bool fis_zero_modp(const F& a) {
  if constexpr (F::UNSAT) return a.template is_zero_modp<B>(); else return a.is_zero();
}

}  // namespace ctt
