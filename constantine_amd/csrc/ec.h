// ec.h -- short-Weierstrass (a = 0) group law in extended-Jacobian (XYZZ) coordinates.
//
// Device bucket type = the reference's EC_ShortW_JacExt:
//   constantine/math/elliptic/ec_shortweierstrass_jacobian_extended.nim:30-60  (X,Y,ZZ,ZZZ; x=X/ZZ, y=Y/ZZZ; neutral ZZ=0)
//   mixedSum_vartime  :258-310   (8M+2S)      -> xyzz_madd
//   mdouble           :232-256                -> xyzz_mdbl
//   sum_vartime       :173-230   (12M+2S)     -> xyzz_add
//   double            :149-171                -> xyzz_dbl
// Input points are EC_ShortW_Aff with neutral encoded as (0,0) (ec_shortweierstrass_affine.nim:47-62).
// All exceptional cases (either operand neutral, P == Q, P == -Q) are handled, as the reference's
// *_vartime formulas do (ec_shortweierstrass_jacobian.nim:798-896).
//
// The formulas are written once for both field flavours.  For the carry-free field (fpu.h) values are only
// bounded by multiples of p, tracked statically here in units of M = F::MULB (a product is < M*p):
//     affine input x, y < M;   stored XYZZ:  X < 4M,  Y < 2M,  ZZ, ZZZ < M.
// fsub<F,B>(a,b) = a - b + B*p needs b < B*p and yields bound(a)+B; the canonical field ignores B.
#pragma once
#include "fpu.h"

namespace ctt {

template <class F>
struct Affine {
  F x, y;
  CTT_HD bool is_inf() const { return x.is_zero() & y.is_zero(); }
  CTT_HD static Affine inf() { return {F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  CTT_HD bool is_inf() const { return zz.is_zero(); }
  CTT_HD static XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  CTT_HD static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one(), F::one()};
  }
};

// 2*(x,y) for an affine, non-neutral point (x, y < M); y == 0 gives ZZ = 0, i.e. the neutral
template <class F>
CTT_HD XYZZ<F> xyzz_mdbl(const F& x, const F& y) {
  constexpr int M = F::MULB;
  F U = F::dbl(y);                               // < 2M
  F V = F::sqr(U);                               // < M
  F W = F::mul(U, V);
  F S = F::mul(x, V);
  F xx = F::sqr(x);
  F Mm = F::add(F::dbl(xx), xx);                 // < 3M
  XYZZ<F> r;
  r.x = fsub<F, 2 * M>(F::sqr(Mm), F::dbl(S));   // < 3M
  r.y = fmul_sub<F, M>(Mm, fsub<F, 3 * M>(S, r.x), y, W);  // Mm*(S-X3) - y*W, < 2M
  r.zz = V;
  r.zzz = W;
  return r;
}

template <class F>
CTT_HD_NOINLINE XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
  constexpr int M = F::MULB;
  if (p.is_inf()) return p;
  F U = F::dbl(p.y);                             // < 4M
  F V = F::sqr(U);
  F W = F::mul(U, V);
  F S = F::mul(p.x, V);
  F xx = F::sqr(p.x);
  F Mm = F::add(F::dbl(xx), xx);                 // < 3M
  XYZZ<F> r;
  r.x = fsub<F, 2 * M>(F::sqr(Mm), F::dbl(S));   // < 3M
  r.y = fmul_sub<F, 2 * M>(Mm, fsub<F, 3 * M>(S, r.x), p.y, W);  // Mm*(S-X3) - Y*W, < 2M
  r.zz = F::mul(V, p.zz);
  r.zzz = F::mul(W, p.zzz);
  return r;
}

// exceptional case of the mixed addition: same x. equal -> doubling of the affine point, opposite -> neutral.
// Returns by value (the accumulator of the hot loop must never have its address taken: it would live in scratch).
template <class F>
CTT_HD_NOINLINE XYZZ<F> xyzz_madd_same_x(F qx, F qy, bool same_y) {
  if (same_y) return xyzz_mdbl<F>(qx, qy);
  return XYZZ<F>::inf();
}

// acc += (neg ? -q : q), q affine
template <class F>
CTT_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& q, bool neg) {
  constexpr int M = F::MULB;
  if (q.is_inf()) return;
  F qy = fcneg<F, M>(q.y, neg);                  // < M
  if (acc.is_inf()) {
    acc.x = q.x;
    acc.y = qy;
    acc.zz = F::one();
    acc.zzz = F::one();
    return;
  }
  F U2 = F::mul(q.x, acc.zz);                    // < M
  F S2 = F::mul(qy, acc.zzz);
  F P = fsub<F, 4 * M>(U2, acc.x);               // < 5M
  F R = fsub<F, 2 * M>(S2, acc.y);               // < 3M
  if (fis_zero_modp<F, 5 * M>(P)) {              // P == +-Q: rare, out of line
    acc = xyzz_madd_same_x<F>(q.x, qy, fis_zero_modp<F, 3 * M>(R));
    return;
  }
  F PP = F::sqr(P);
  F PPP = F::mul(P, PP);
  F Q = F::mul(acc.x, PP);
  F X3 = fsub<F, 2 * M>(fsub<F, M>(F::sqr(R), PPP), F::dbl(Q));        // < 4M
  F Y3 = fmul_sub<F, 2 * M>(R, fsub<F, 4 * M>(Q, X3), acc.y, PPP);     // R*(Q-X3) - Y1*PPP, < 2M
  acc.x = X3;
  acc.y = Y3;
  acc.zz = F::mul(acc.zz, PP);
  acc.zzz = F::mul(acc.zzz, PPP);
}

// acc += q, both XYZZ  (not on the hot path: kept out of line to bound code size / compile time)
template <class F>
CTT_HD_NOINLINE void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& q) {
  constexpr int M = F::MULB;
  if (q.is_inf()) return;
  if (acc.is_inf()) {
    acc = q;
    return;
  }
  F U1 = F::mul(acc.x, q.zz);
  F U2 = F::mul(q.x, acc.zz);
  F S1 = F::mul(acc.y, q.zzz);
  F S2 = F::mul(q.y, acc.zzz);
  F P = fsub<F, M>(U2, U1);                      // < 2M
  F R = fsub<F, M>(S2, S1);                      // < 2M
  if (fis_zero_modp<F, 2 * M>(P)) {
    if (fis_zero_modp<F, 2 * M>(R)) {
      acc = xyzz_dbl<F>(acc);
    } else {
      acc = XYZZ<F>::inf();
    }
    return;
  }
  F PP = F::sqr(P);
  F PPP = F::mul(P, PP);
  F Q = F::mul(U1, PP);
  F X3 = fsub<F, 2 * M>(fsub<F, M>(F::sqr(R), PPP), F::dbl(Q));       // < 4M
  F Y3 = fmul_sub<F, M>(R, fsub<F, 4 * M>(Q, X3), S1, PPP);          // R*(Q-X3) - S1*PPP, < 2M
  acc.x = X3;
  acc.y = Y3;
  acc.zz = F::mul(F::mul(acc.zz, q.zz), PP);
  acc.zzz = F::mul(F::mul(acc.zzz, q.zzz), PPP);
}

// x = X/ZZ, y = Y/ZZZ (fromJacobianExtended_vartime, jacobian_extended.nim:353-379, then affine).
// Canonical fields only (host tail, input generator).
template <class F>
CTT_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
  if (p.is_inf()) return Affine<F>::inf();
  // one inversion: 1/(ZZ*ZZZ)
  F i = F::inv(F::mul(p.zz, p.zzz));
  F izz = F::mul(i, p.zzz);
  F izzz = F::mul(i, p.zz);
  return {F::mul(p.x, izz), F::mul(p.y, izzz)};
}

}  // namespace ctt
