// host_fp64.h -- host-only prime field on 64-bit limbs (unsigned __int128 products).
//
// Same static interface and the same in-memory representation as ctt::Fp<PP> (Montgomery residue,
// little-endian limbs), so XYZZ<Fp64<PP>> can be laid over the bytes the device returns.  Used for the
// final Horner over (window, bit) and the affine normalisation of the result -- the only EC arithmetic
// the engine does on the CPU (about 2*W*c group operations per MSM, the serial tail the reference runs
// at ec_multi_scalar_mul.nim:250-254).
#pragma once
#include <string.h>

#include "ec.h"

namespace ctt {

template <class PP>
struct Fp64 {
  using Params = PP;
  static constexpr int N = PP::N / 2;
  static constexpr int NBYTES = 8 * N;
  static constexpr bool UNSAT = false;
  static constexpr int MULB = 1;
  typedef unsigned __int128 u128;
  uint64_t l[N];

  static inline uint64_t P(int i) { return (uint64_t)PP::P[2 * i] | ((uint64_t)PP::P[2 * i + 1] << 32); }
  static inline uint64_t m0inv() {
    // -p^-1 mod 2^64 from the 32-bit constant by one Newton step
    uint64_t p0 = P(0);
    uint64_t inv = (uint64_t)(0u - PP::M0INV);  // p^-1 mod 2^32
    inv *= 2 - p0 * inv;                        // mod 2^64
    return (uint64_t)0 - inv;
  }

  static inline Fp64 zero() { Fp64 r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
  static inline Fp64 one() {
    Fp64 r;
    for (int i = 0; i < N; i++) r.l[i] = (uint64_t)PP::ONE[2 * i] | ((uint64_t)PP::ONE[2 * i + 1] << 32);
    return r;
  }
  inline bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
  static inline bool eq(const Fp64& a, const Fp64& b) { uint64_t d = 0; for (int i = 0; i < N; i++) d |= a.l[i] ^ b.l[i]; return d == 0; }
  static inline Fp64 select(bool c, const Fp64& a, const Fp64& b) { return c ? a : b; }

  static inline Fp64 reduce_once(const uint64_t* t, uint64_t top) {
    uint64_t d[N];
    uint64_t bw = 0;
    for (int i = 0; i < N; i++) {
      u128 x = (u128)t[i] - P(i) - bw;
      d[i] = (uint64_t)x;
      bw = (uint64_t)(x >> 64) & 1;
    }
    Fp64 r;
    const bool ge = (bw == 0) | (top != 0);
    for (int i = 0; i < N; i++) r.l[i] = ge ? d[i] : t[i];
    return r;
  }
  static inline Fp64 add(const Fp64& a, const Fp64& b) {
    uint64_t t[N];
    uint64_t c = 0;
    for (int i = 0; i < N; i++) {
      u128 s = (u128)a.l[i] + b.l[i] + c;
      t[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    return reduce_once(t, c);
  }
  static inline Fp64 dbl(const Fp64& a) { return add(a, a); }
  static inline Fp64 sub(const Fp64& a, const Fp64& b) {
    uint64_t t[N];
    uint64_t bw = 0;
    for (int i = 0; i < N; i++) {
      u128 x = (u128)a.l[i] - b.l[i] - bw;
      t[i] = (uint64_t)x;
      bw = (uint64_t)(x >> 64) & 1;
    }
    Fp64 r;
    const uint64_t mask = (uint64_t)0 - bw;
    uint64_t c = 0;
    for (int i = 0; i < N; i++) {
      u128 s = (u128)t[i] + (P(i) & mask) + c;
      r.l[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    return r;
  }
  static inline Fp64 neg(const Fp64& a) { return a.is_zero() ? a : sub(zero(), a); }
  static inline Fp64 cneg(const Fp64& a, bool c) { return c ? neg(a) : a; }

  // Montgomery multiplication, coarsely integrated operand scanning (the reference's shape for the CPU,
  // limbs_montgomery.nim:268-310), every loop fully unrolled (N is 4 or 6): one row of a*b_i and one row of m*p per step,
  // each a chain of 64x64->128 multiply-adds whose sum a*b + t + carry cannot overflow 128 bits.  The host tail of an MSM
  // is one dependent chain of W*c doublings (msm_pipeline.h combine_groups): this routine IS its latency (round 2's
  // product-scanning form with a 192-bit accumulator took 65 ns per 381-bit product, this one half of that).
  static inline Fp64 mul(const Fp64& a, const Fp64& b) {
    uint64_t p[N];
#pragma GCC unroll 8
    for (int i = 0; i < N; i++) p[i] = P(i);
    const uint64_t mi = m0inv();
    uint64_t t[N + 1];
#pragma GCC unroll 8
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma GCC unroll 8
    for (int i = 0; i < N; i++) {
      const uint64_t bi = b.l[i];
      u128 c = 0;
#pragma GCC unroll 8
      for (int j = 0; j < N; j++) {
        c += (u128)a.l[j] * bi + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      u128 top = (u128)t[N] + (uint64_t)c;      // t has N+2 words for a moment: the extra bit is `over`
      t[N] = (uint64_t)top;
      const uint64_t over = (uint64_t)(top >> 64);
      const uint64_t m = t[0] * mi;
      c = ((u128)m * p[0] + t[0]) >> 64;
#pragma GCC unroll 8
      for (int j = 1; j < N; j++) {
        c += (u128)m * p[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      top = (u128)t[N] + (uint64_t)c;
      t[N - 1] = (uint64_t)top;
      t[N] = over + (uint64_t)(top >> 64);
    }
    return reduce_once(t, t[N]);
  }
  static inline Fp64 sqr(const Fp64& a) { return mul(a, a); }

  // 1/a by division steps on the 32-bit words of the residue (modinv.h): (aR)^-1, then one product with R^3.
  // (a^(p-2) cost 17-25 us of every MSM's host tail; this is ~3 us)
  static inline Fp64 inv(const Fp64& a) {
    uint32_t w[2 * N], o[2 * N];
    memcpy(w, a.l, sizeof(w));
    ModInv<PP>::inv_words(w, o);
    Fp64 t, r3;
    memcpy(t.l, o, sizeof(o));
    for (int i = 0; i < N; i++) r3.l[i] = (uint64_t)PP::R3[2 * i] | ((uint64_t)PP::R3[2 * i + 1] << 32);
    return mul(t, r3);
  }
};

// device field type -> host field type (canonical Montgomery form of the reference, 64-bit limbs)
template <class F> struct HostField;
template <class PP> struct HostField<Fp<PP>> {
  using type = Fp64<PP>;
  static inline type conv(const Fp<PP>& a) {  // same bytes
    type r;
    memcpy(r.l, a.l, sizeof(r.l));
    return r;
  }
};
template <class PP> struct HostField<Fp2<Fp<PP>>> {
  using type = Fp2<Fp64<PP>>;
  static inline type conv(const Fp2<Fp<PP>>& a) { return {HostField<Fp<PP>>::conv(a.c0), HostField<Fp<PP>>::conv(a.c1)}; }
};
// carry-free device field: value = x*R' (mod p), not reduced -> reduce, then one Montgomery product with
// C_OUT = R^2/R' gives x*R (fpu.h)
template <class UP> struct HostField<FpU<UP>> {
  using PP = typename UP::Sat;
  using type = Fp64<PP>;
  static inline type conv(const FpU<UP>& a) {
    constexpr int N = type::N;
    typedef unsigned __int128 u128;
    uint64_t v[N + 1];
    for (int i = 0; i <= N; i++) v[i] = 0;
    for (int i = 0; i < UP::NL; i++) {  // v += l[i] << (LB*i)
      const int pos = UP::LB * i, w = pos >> 6, sh = pos & 63;
      u128 x = (u128)a.l[i] << sh;
      uint64_t c = 0;
      for (int j = w; j <= N && (x != 0 || c != 0); j++) {
        u128 s = (u128)v[j] + (uint64_t)x + c;
        v[j] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
        x >>= 64;
      }
    }
    for (;;) {  // v < (small multiple of p): subtract p until v < p
      uint64_t d[N + 1];
      uint64_t bw = 0;
      for (int i = 0; i <= N; i++) {
        u128 x = (u128)v[i] - (i < N ? type::P(i) : 0) - bw;
        d[i] = (uint64_t)x;
        bw = (uint64_t)(x >> 64) & 1;
      }
      if (bw) break;
      for (int i = 0; i <= N; i++) v[i] = d[i];
    }
    type t, c;
    for (int i = 0; i < N; i++) {
      t.l[i] = v[i];
      c.l[i] = (uint64_t)UP::C_OUT[2 * i] | ((uint64_t)UP::C_OUT[2 * i + 1] << 32);
    }
    return type::mul(t, c);
  }
};

template <class UP> struct HostField<Fp2<FpU<UP>>> {
  using type = Fp2<Fp64<typename UP::Sat>>;
  static inline type conv(const Fp2<FpU<UP>>& a) { return {HostField<FpU<UP>>::conv(a.c0), HostField<FpU<UP>>::conv(a.c1)}; }
};

template <class F>
static inline XYZZ<typename HostField<F>::type> xyzz_to_host(const XYZZ<F>& p) {
  using H = HostField<F>;
  return {H::conv(p.x), H::conv(p.y), H::conv(p.zz), H::conv(p.zzz)};
}

}  // namespace ctt
