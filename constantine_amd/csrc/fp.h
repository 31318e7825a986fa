// fp.h -- prime-field and quadratic-extension arithmetic for the MSM path, 32-bit limbs.
//
// Replaces, on the device, the field layer the reference's MSM bottoms out in:
//   FF.prod/square  -> mulMont/squareMont   constantine/math/arithmetic/finite_fields.nim:268-281,
//                                           constantine/math/arithmetic/limbs_montgomery.nim:484-550
//   FF.sum/diff/neg/double                  finite_fields.nim:172-266
//   Fp2 complex mul/sqr                     constantine/math/extension_fields/towers.nim:758-878
//
// Representation is bit-identical to the reference's (Montgomery residue a*R mod p with
// R = 2^(64*L), little-endian limbs, platforms/abstractions.nim:131-143): a 64-bit LE limb array
// reinterpreted as 2L 32-bit limbs.  CDNA4 has no 64x64 multiplier, so the multiplier is a
// product-scanning (Comba / FIPS) Montgomery multiplication built on v_mad_u64_u32 with a
// three-word column accumulator: one v_mad_u64_u32 + one v_addc_co_u32 per partial product.
// Every element is kept fully reduced in [0,p).
//
// The same templates compile for the host (plain C++ path of mac()) where they serve the final
// window combine and the test harness; the device path is the one measured.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CTT_HD __host__ __device__ __forceinline__
#define CTT_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define CTT_HD inline __attribute__((always_inline))
#define CTT_HD_NOINLINE __attribute__((noinline))
#endif

#include "field_params.h"

namespace ctt {

// ---------------------------------------------------------------------------------------------
// (hi:lo) += a*b, 96-bit column accumulator
//
// Device form: v_mad_u64_u32 (32x32+64, carry-out in VCC) followed by v_addc_co_u32 into the third word.
// hipcc pads every inline-asm statement with an s_nop (it cannot see what is inside), so the MACs of a
// column are issued in groups of up to four per statement.
// ---------------------------------------------------------------------------------------------
struct Acc3 {
  uint64_t lo;
  uint32_t hi;
};

#define CTT_MAC_STR(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"

CTT_HD void mac_host(Acc3& acc, uint32_t a, uint32_t b) {
  uint64_t prod = (uint64_t)a * b;
  acc.lo += prod;
  acc.hi += (acc.lo < prod) ? 1u : 0u;
}

CTT_HD void mac(Acc3& acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) : "+v"(acc.lo), "+v"(acc.hi) : "v"(a), "v"(b) : "vcc");
#else
  mac_host(acc, a, b);
#endif
}
CTT_HD void mac2(Acc3& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) CTT_MAC_STR(4, 5) : "+v"(acc.lo), "+v"(acc.hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
#else
  mac_host(acc, a0, b0); mac_host(acc, a1, b1);
#endif
}
CTT_HD void mac3(Acc3& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) CTT_MAC_STR(4, 5) CTT_MAC_STR(6, 7)
      : "+v"(acc.lo), "+v"(acc.hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2) : "vcc");
#else
  mac_host(acc, a0, b0); mac_host(acc, a1, b1); mac_host(acc, a2, b2);
#endif
}
CTT_HD void mac4(Acc3& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3,
                 uint32_t b3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) CTT_MAC_STR(4, 5) CTT_MAC_STR(6, 7) CTT_MAC_STR(8, 9)
      : "+v"(acc.lo), "+v"(acc.hi)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
      : "vcc");
#else
  mac_host(acc, a0, b0); mac_host(acc, a1, b1); mac_host(acc, a2, b2); mac_host(acc, a3, b3);
#endif
}

// same with the second factor a compile-time constant of the field (lives in an SGPR on the device;
// gfx9 VOP3 takes no literal and only one SGPR per instruction)
CTT_HD void mac_k(Acc3& acc, uint32_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) : "+v"(acc.lo), "+v"(acc.hi) : "v"(a), "s"(k) : "vcc");
#else
  mac_host(acc, a, k);
#endif
}
CTT_HD void mac2_k(Acc3& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) CTT_MAC_STR(4, 5) : "+v"(acc.lo), "+v"(acc.hi) : "v"(a0), "s"(k0), "v"(a1), "s"(k1) : "vcc");
#else
  mac_host(acc, a0, k0); mac_host(acc, a1, k1);
#endif
}
CTT_HD void mac3_k(Acc3& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1, uint32_t a2, uint32_t k2) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) CTT_MAC_STR(4, 5) CTT_MAC_STR(6, 7)
      : "+v"(acc.lo), "+v"(acc.hi) : "v"(a0), "s"(k0), "v"(a1), "s"(k1), "v"(a2), "s"(k2) : "vcc");
#else
  mac_host(acc, a0, k0); mac_host(acc, a1, k1); mac_host(acc, a2, k2);
#endif
}
CTT_HD void mac4_k(Acc3& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1, uint32_t a2, uint32_t k2, uint32_t a3,
                   uint32_t k3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(CTT_MAC_STR(2, 3) CTT_MAC_STR(4, 5) CTT_MAC_STR(6, 7) CTT_MAC_STR(8, 9)
      : "+v"(acc.lo), "+v"(acc.hi)
      : "v"(a0), "s"(k0), "v"(a1), "s"(k1), "v"(a2), "s"(k2), "v"(a3), "s"(k3)
      : "vcc");
#else
  mac_host(acc, a0, k0); mac_host(acc, a1, k1); mac_host(acc, a2, k2); mac_host(acc, a3, k3);
#endif
}

// acc += sum_{i=I}^{END-1} a[i]*b[K-i], grouped four MACs per statement
template <int K, int I, int END>
struct MacAB {
  static CTT_HD void run(Acc3& acc, const uint32_t* a, const uint32_t* b) {
    if constexpr (END - I >= 4) {
      mac4(acc, a[I], b[K - I], a[I + 1], b[K - I - 1], a[I + 2], b[K - I - 2], a[I + 3], b[K - I - 3]);
      MacAB<K, I + 4, END>::run(acc, a, b);
    } else if constexpr (END - I == 3) {
      mac3(acc, a[I], b[K - I], a[I + 1], b[K - I - 1], a[I + 2], b[K - I - 2]);
    } else if constexpr (END - I == 2) {
      mac2(acc, a[I], b[K - I], a[I + 1], b[K - I - 1]);
    } else if constexpr (END - I == 1) {
      mac(acc, a[I], b[K - I]);
    }
  }
};

// acc += sum_{i=I}^{END-1} m[i]*P[K-i]; zero limbs of the modulus are skipped at compile time
// (Pallas/Vesta: three of eight limbs are zero)
template <class PP, int K, int I, int END>
struct MacMP {
  static constexpr bool nz(int i) { return i < END && PP::P[K - i] != 0u; }
  static CTT_HD void run(Acc3& acc, const uint32_t* m) {
    if constexpr (I >= END) {
      return;
    } else if constexpr (!nz(I)) {
      MacMP<PP, K, I + 1, END>::run(acc, m);
    } else if constexpr (nz(I + 1) && nz(I + 2) && nz(I + 3)) {
      mac4_k(acc, m[I], PP::P[K - I], m[I + 1], PP::P[K - I - 1], m[I + 2], PP::P[K - I - 2], m[I + 3], PP::P[K - I - 3]);
      MacMP<PP, K, I + 4, END>::run(acc, m);
    } else if constexpr (nz(I + 1) && nz(I + 2)) {
      mac3_k(acc, m[I], PP::P[K - I], m[I + 1], PP::P[K - I - 1], m[I + 2], PP::P[K - I - 2]);
      MacMP<PP, K, I + 3, END>::run(acc, m);
    } else if constexpr (nz(I + 1)) {
      mac2_k(acc, m[I], PP::P[K - I], m[I + 1], PP::P[K - I - 1]);
      MacMP<PP, K, I + 2, END>::run(acc, m);
    } else {
      mac_k(acc, m[I], PP::P[K - I]);
      MacMP<PP, K, I + 1, END>::run(acc, m);
    }
  }
};

CTT_HD void acc_shift(Acc3& acc) {
  acc.lo = (acc.lo >> 32) | ((uint64_t)acc.hi << 32);
  acc.hi = 0;
}

// ---------------------------------------------------------------------------------------------
// Fp
// ---------------------------------------------------------------------------------------------
template <class PP>
struct ModInv;   // modinv.h (included at the end of this file)

template <class PP>
struct Fp {
  using Params = PP;
  static constexpr int N = PP::N;
  static constexpr int NBYTES = 4 * N;
  static constexpr bool UNSAT = false;  // canonical representation: always in [0,p)
  static constexpr int MULB = 1;
  uint32_t l[N];

  CTT_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  CTT_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = PP::ONE[i];
    return r;
  }
  CTT_HD bool is_zero() const {
    uint32_t a = 0;
#pragma unroll
    for (int i = 0; i < N; i++) a |= l[i];
    return a == 0;
  }
  CTT_HD static bool eq(const Fp& a, const Fp& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d |= a.l[i] ^ b.l[i];
    return d == 0;
  }
  CTT_HD static Fp select(bool c, const Fp& a, const Fp& b) {  // c ? a : b
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
  }

  // r = t - p if t >= p else t, where t may carry one extra bit `top`
  CTT_HD static Fp reduce_once(const uint32_t* t, uint32_t top) {
    uint32_t d[N];
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t x = (uint64_t)t[i] - PP::P[i] - bw;
      d[i] = (uint32_t)x;
      bw = (uint32_t)(x >> 32) & 1u;
    }
    bool ge = (bw == 0) | (top != 0);
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = ge ? d[i] : t[i];
    return r;
  }

  CTT_HD static Fp add(const Fp& a, const Fp& b) {
    uint32_t t[N];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t s = (uint64_t)a.l[i] + b.l[i] + c;
      t[i] = (uint32_t)s;
      c = (uint32_t)(s >> 32);
    }
    return reduce_once(t, c);
  }
  CTT_HD static Fp dbl(const Fp& a) { return add(a, a); }

  CTT_HD static Fp sub(const Fp& a, const Fp& b) {
    uint32_t t[N];
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t x = (uint64_t)a.l[i] - b.l[i] - bw;
      t[i] = (uint32_t)x;
      bw = (uint32_t)(x >> 32) & 1u;
    }
    // add p back when the subtraction borrowed
    uint32_t mask = 0u - bw;
    uint32_t c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t s = (uint64_t)t[i] + (PP::P[i] & mask) + c;
      r.l[i] = (uint32_t)s;
      c = (uint32_t)(s >> 32);
    }
    return r;
  }
  CTT_HD static Fp neg(const Fp& a) {  // neg(0) = 0, finite_fields.nim:340-363
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.l[i];
    uint32_t mask = nz ? 0xffffffffu : 0u;
    uint32_t bw = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t x = (uint64_t)(PP::P[i] & mask) - a.l[i] - bw;
      r.l[i] = (uint32_t)x;
      bw = (uint32_t)(x >> 32) & 1u;
    }
    return r;
  }
  CTT_HD static Fp cneg(const Fp& a, bool c) { return select(c, neg(a), a); }

  // Montgomery product a*b*R^-1 mod p, product scanning with interleaved reduction (FIPS form of
  // limbs_montgomery.nim:268-310); result fully reduced.
  template <int K>
  static CTT_HD void mul_col(Acc3& acc, const uint32_t* a, const uint32_t* b, uint32_t* m, uint32_t* t) {
    if constexpr (K < N) {
      MacAB<K, 0, K + 1>::run(acc, a, b);
      MacMP<PP, K, 0, K>::run(acc, m);
      m[K] = (uint32_t)acc.lo * PP::M0INV;
      mac_k(acc, m[K], PP::P[0]);
    } else {
      MacAB<K, K - N + 1, N>::run(acc, a, b);
      MacMP<PP, K, K - N + 1, N>::run(acc, m);
      t[K - N] = (uint32_t)acc.lo;
    }
    acc_shift(acc);
    if constexpr (K + 1 < 2 * N) mul_col<K + 1>(acc, a, b, m, t);
  }
  CTT_HD static Fp mul(const Fp& a, const Fp& b) {
    Acc3 acc;
    acc.lo = 0;
    acc.hi = 0;
    uint32_t m[N];
    uint32_t t[N];
    mul_col<0>(acc, a.l, b.l, m, t);
    return reduce_once(t, (uint32_t)acc.lo);
  }

  // Montgomery square: cross products once, doubled in the accumulator
  CTT_HD static Fp sqr(const Fp& a) {
    Acc3 acc;
    acc.lo = 0;
    acc.hi = 0;
    uint32_t m[N];
    uint32_t t[N];
#pragma unroll
    for (int k = 0; k < 2 * N; k++) {
      // cross terms a_i*a_j, i<j, i+j=k  (accumulated apart, then doubled)
      Acc3 x;
      x.lo = 0;
      x.hi = 0;
      bool any = false;
#pragma unroll
      for (int i = 0; i < N; i++) {
        int j = k - i;
        if (j > i && j < N) {
          mac(x, a.l[i], a.l[j]);
          any = true;
        }
      }
      if (any) {
        // acc += 2*x  (x < 2^68 so 2x fits the 96-bit accumulator)
        uint64_t xlo2 = x.lo << 1;
        uint32_t xhi2 = (x.hi << 1) | (uint32_t)(x.lo >> 63);
        uint64_t s = acc.lo + xlo2;
        uint32_t cy = s < xlo2 ? 1u : 0u;
        acc.lo = s;
        acc.hi += xhi2 + cy;
      }
      if ((k & 1) == 0 && (k >> 1) < N) mac(acc, a.l[k >> 1], a.l[k >> 1]);
      if (k < N) {
#pragma unroll
        for (int i = 0; i < k; i++)
          if (PP::P[k - i] != 0u) mac_k(acc, m[i], PP::P[k - i]);
        m[k] = (uint32_t)acc.lo * PP::M0INV;
        mac_k(acc, m[k], PP::P[0]);
      } else {
#pragma unroll
        for (int i = k - N + 1; i < N; i++)
          if (PP::P[k - i] != 0u) mac_k(acc, m[i], PP::P[k - i]);
        t[k - N] = (uint32_t)acc.lo;
      }
      acc_shift(acc);
    }
    return reduce_once(t, (uint32_t)acc.lo);
  }

  // Montgomery -> canonical (fromMont, limbs_montgomery.nim:577-603): multiply by 1
  CTT_HD static Fp from_mont(const Fp& a) {
    Fp o = zero();
    o.l[0] = 1;
    return mul(a, o);
  }
  CTT_HD static Fp to_mont(const Fp& a) {
    Fp r2;
#pragma unroll
    for (int i = 0; i < N; i++) r2.l[i] = PP::R2[i];
    return mul(a, r2);
  }

  // 1/a (inv(0) = 0): same value as the reference's inv_vartime (finite_fields.nim:386-396).  Division steps on the plain
  // words of the residue (modinv.h) give (aR)^-1; one Montgomery product with R^3 makes that a^-1 R.
  CTT_HD static Fp inv(const Fp& a) {
    Fp t, r3;
    ModInv<PP>::inv_words(a.l, t.l);
#pragma unroll
    for (int i = 0; i < N; i++) r3.l[i] = PP::R3[i];
    return mul(t, r3);
  }
  // a^(p-2): rounds 1-2's inversion, kept as the cross-check of the one above (tests) and for the cost table in DESIGN.md
  CTT_HD static Fp inv_fermat(const Fp& a) {
    Fp r = one();
    for (int i = 32 * N - 1; i >= 0; i--) {
      r = sqr(r);
      if ((PP::PM2[i >> 5] >> (i & 31)) & 1u) r = mul(r, a);
    }
    return r;
  }
};

// ---------------------------------------------------------------------------------------------
// Fp2 = Fp[i]/(i^2+1)  (both in-scope towers use non-residue -1,
// named/config_fields_and_curves.nim:128,283)
// ---------------------------------------------------------------------------------------------
template <class F>
struct Fp2 {
  using Base = F;
  static constexpr int NBYTES = 2 * F::NBYTES;
  static constexpr bool UNSAT = false;
  static constexpr int MULB = 1;
  F c0, c1;

  CTT_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
  CTT_HD static Fp2 one() { return {F::one(), F::zero()}; }
  CTT_HD bool is_zero() const { return c0.is_zero() & c1.is_zero(); }
  CTT_HD static bool eq(const Fp2& a, const Fp2& b) { return F::eq(a.c0, b.c0) & F::eq(a.c1, b.c1); }
  CTT_HD static Fp2 select(bool c, const Fp2& a, const Fp2& b) {
    return {F::select(c, a.c0, b.c0), F::select(c, a.c1, b.c1)};
  }
  CTT_HD static Fp2 add(const Fp2& a, const Fp2& b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
  CTT_HD static Fp2 sub(const Fp2& a, const Fp2& b) { return {F::sub(a.c0, b.c0), F::sub(a.c1, b.c1)}; }
  CTT_HD static Fp2 dbl(const Fp2& a) { return {F::dbl(a.c0), F::dbl(a.c1)}; }
  CTT_HD static Fp2 neg(const Fp2& a) { return {F::neg(a.c0), F::neg(a.c1)}; }
  CTT_HD static Fp2 cneg(const Fp2& a, bool c) { return {F::cneg(a.c0, c), F::cneg(a.c1, c)}; }
  // Karatsuba complex product (prod_complex, towers.nim:818-850): 3 base multiplications
  CTT_HD static Fp2 mul(const Fp2& a, const Fp2& b) {
    F v0 = F::mul(a.c0, b.c0);
    F v1 = F::mul(a.c1, b.c1);
    F s = F::mul(F::add(a.c0, a.c1), F::add(b.c0, b.c1));
    return {F::sub(v0, v1), F::sub(F::sub(s, v0), v1)};
  }
  // complex squaring (square_complex, towers.nim:758-796): 2 base multiplications
  CTT_HD static Fp2 sqr(const Fp2& a) {
    F s = F::add(a.c0, a.c1);
    F d = F::sub(a.c0, a.c1);
    F m = F::mul(a.c0, a.c1);
    return {F::mul(s, d), F::dbl(m)};
  }
  CTT_HD static Fp2 inv(const Fp2& a) {
    F n = F::inv(F::add(F::sqr(a.c0), F::sqr(a.c1)));
    return {F::mul(a.c0, n), F::neg(F::mul(a.c1, n))};
  }
};

}  // namespace ctt

#include "modinv.h"
