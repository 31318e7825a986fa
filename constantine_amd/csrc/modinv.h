// modinv.h -- modular inversion by Bernstein-Yang division steps ("safegcd"), 32-bit lanes.
//
// The reference inverts with the same family of algorithm on 64-bit words (constantine/math/arithmetic/limbs_exgcd.nim:844-876
// invmod_vartime, batches of 62 divsteps); rounds 1-2 of this engine used Fermat's a^(p-2) instead -- ~570 field products per
// inversion, every one of them paid by all 64 lanes of a wave (batch_affine at 2^16: 1.03 ms of pure inversion latency).
// Here: batches of 30 divsteps on the low words of (f, g) with 32-bit operations, each batch summarised as a 2x2 transition
// matrix with entries of at most 31 bits that is then applied to the full-width (f, g) and, modulo p, to (d, e); numbers are
// signed, 30 bits per limb.  One batch is ~720 32-bit VALU operations + 10 multiply-adds per limb, 590 divsteps bound the
// 256-bit case and the loop simply runs until g = 0 (a wave runs as long as its slowest lane): ~15 k instructions per
// inversion for a 381-bit modulus against ~280 k for Fermat.
//
// Every lane executes the same instruction stream inside a batch (the divstep is branch-free), so the routine is fit for
// SIMD execution; the number of batches depends on the input (variable time, like every routine on this path).
#pragma once
#include <stdint.h>

#ifndef CTT_HD
#error "include fp.h (which ends by including this file), not modinv.h"
#endif

namespace ctt {

template <class PP>
struct ModInv {
  static constexpr int N = PP::N;                        // 32-bit words of the modulus
  static constexpr int L = (PP::BITS + 2 + 29) / 30;     // signed 30-bit limbs: values in (-2p, 2p)
  static constexpr int32_t M30 = (int32_t)((1u << 30) - 1u);
  // divsteps needed for any input of this width: floor((49 d + 57) / 17) (Bernstein-Yang, theorem 11.2); the loop stops at
  // g = 0 long before (about 2.1 divsteps per bit), this only bounds it
  static constexpr int MAX_BATCHES = ((49 * PP::BITS + 57) / 17 + 29) / 30 + 1;

  struct S30 { int32_t v[L]; };

  // bits [30 i, 30 i + 30) of a little-endian 32-bit word array
  static constexpr int32_t limb30(const uint32_t* w, int i) {
    const int pos = 30 * i, word = pos >> 5, sh = pos & 31;
    uint64_t x = word < N ? (uint64_t)w[word] : 0u;
    if (word + 1 < N) x |= (uint64_t)w[word + 1] << 32;
    return (int32_t)((x >> sh) & (uint64_t)M30);
  }
  static constexpr S30 modulus() {
    S30 m{};
    for (int i = 0; i < L; i++) m.v[i] = limb30(PP::P, i);
    return m;
  }
  // p^-1 mod 2^30 (Newton from the low word; p is odd)
  static constexpr uint32_t pinv30() {
    const uint32_t p0 = PP::P[0];
    uint32_t x = p0;                     // correct to 3 bits
    for (int i = 0; i < 5; i++) x *= 2u - p0 * x;
    return x & (uint32_t)M30;
  }

  // 30 division steps on the low 30 bits of (f, g): the transition matrix t = (u, v; q, r) with
  //   2^30 (f', g') = t (f, g)   exactly,  |u| + |v| <= 2^30, |q| + |r| <= 2^30.
  // zeta = -(delta + 1/2) of the half-delta variant, starting at -1.  Branch-free: masks c1 = (zeta < 0), c2 = (g odd).
  CTT_HD static int32_t divsteps30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t* t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
    for (int i = 0; i < 30; i++) {
      uint32_t c1 = (uint32_t)(zeta >> 31);
      const uint32_t c2 = 0u - (g & 1u);
      // (x, y, z) = +-(f, u, v)
      const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
      g += x & c2;
      q += y & c2;
      r += z & c2;
      c1 &= c2;                                   // swap case: zeta < 0 and g odd
      zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;  // -zeta - 2, or zeta - 1
      f += g & c1;
      u += q & c1;
      v += r & c1;
      g >>= 1;
      u <<= 1;
      v <<= 1;
    }
    t[0] = (int32_t)u;
    t[1] = (int32_t)v;
    t[2] = (int32_t)q;
    t[3] = (int32_t)r;
    return zeta;
  }

  // (f, g) = t (f, g) / 2^30   (exact: the low 30 bits of both combinations vanish)
  CTT_HD static void update_fg(S30& f, S30& g, const int32_t* t) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = u * f.v[0] + v * g.v[0];
    int64_t cg = q * f.v[0] + r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < L; i++) {
      const int64_t fi = f.v[i], gi = g.v[i];
      cf += u * fi + v * gi;
      cg += q * fi + r * gi;
      f.v[i - 1] = (int32_t)cf & M30;
      g.v[i - 1] = (int32_t)cg & M30;
      cf >>= 30;
      cg >>= 30;
    }
    f.v[L - 1] = (int32_t)cf;
    g.v[L - 1] = (int32_t)cg;
  }

  // (d, e) = t (d, e) / 2^30 mod p, both kept in (-2p, p): a multiple of p (md, me) is added so that the division is exact
  CTT_HD static void update_de(S30& d, S30& e, const int32_t* t) {
    constexpr S30 P = modulus();
    constexpr uint32_t PINV = pinv30();
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[L - 1] >> 31, se = e.v[L - 1] >> 31;      // sign masks
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);      // start from d, e >= 0 (add p once per negative input)
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    md -= (int32_t)((PINV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((PINV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)P.v[0] * md;
    ce += (int64_t)P.v[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < L; i++) {
      const int64_t di = d.v[i], ei = e.v[i];
      cd += (int64_t)u * di + (int64_t)v * ei + (int64_t)P.v[i] * md;
      ce += (int64_t)q * di + (int64_t)r * ei + (int64_t)P.v[i] * me;
      d.v[i - 1] = (int32_t)cd & M30;
      e.v[i - 1] = (int32_t)ce & M30;
      cd >>= 30;
      ce >>= 30;
    }
    d.v[L - 1] = (int32_t)cd;
    e.v[L - 1] = (int32_t)ce;
  }

  // x (N words, 0 <= x < p)  ->  out = x^-1 mod p (N words, canonical);  x = 0 gives 0.
  CTT_HD static void inv_words(const uint32_t* x, uint32_t* out) {
    constexpr S30 P = modulus();
    S30 f = P, g, d, e;
#pragma unroll
    for (int i = 0; i < L; i++) {
      g.v[i] = limb30(x, i);
      d.v[i] = 0;
      e.v[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;
    for (int it = 0; it < MAX_BATCHES; it++) {
      int32_t t[4];
      zeta = divsteps30(zeta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), t);
      update_de(d, e, t);
      update_fg(f, g, t);
      int32_t nz = 0;
#pragma unroll
      for (int i = 0; i < L; i++) nz |= g.v[i];
      if (nz == 0) break;
    }
    // g = 0, f = +-gcd = +-1 (0 input: f = +-p and d = 0), d = +-x^-1 in (-2p, p): bring sign(f) * d into [0, p)
    const int32_t fneg = f.v[L - 1] >> 31;
    int32_t add = d.v[L - 1] >> 31;
#pragma unroll
    for (int i = 0; i < L; i++) d.v[i] = ((d.v[i] + (P.v[i] & add)) ^ fneg) - fneg;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      d.v[i + 1] += d.v[i] >> 30;
      d.v[i] &= M30;
    }
    add = d.v[L - 1] >> 31;
#pragma unroll
    for (int i = 0; i < L; i++) d.v[i] += P.v[i] & add;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      d.v[i + 1] += d.v[i] >> 30;
      d.v[i] &= M30;
    }
    // 30-bit limbs -> 32-bit words
#pragma unroll
    for (int w = 0; w < N; w++) {
      const int pos = 32 * w, i = pos / 30, sh = pos - 30 * i;
      uint64_t acc = (uint64_t)(uint32_t)d.v[i] >> sh;
      if (i + 1 < L) acc |= (uint64_t)(uint32_t)d.v[i + 1] << (30 - sh);
      if (i + 2 < L) acc |= (uint64_t)(uint32_t)d.v[i + 2] << (60 - sh);
      out[w] = (uint32_t)acc;
    }
  }
};

}  // namespace ctt
