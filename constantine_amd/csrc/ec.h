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
//     affine input x, y < M;   stored XYZZ:  X < 4M + 1 (xyzz_madd, see XYZZ_XB),  Y < 2M,  ZZ, ZZZ < M.
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
  F V, xx, W, S;
  fsqr_pair<F>(U, x, V, xx);                     // V = U^2, xx = x^2   (< M)
  fmul_pair<F>(U, V, x, V, W, S);                // W = U*V, S = x*V
  F Mm = F::add(F::dbl(xx), xx);                 // < 3M
  XYZZ<F> r;
  r.x = fsub<F, 2 * M>(F::sqr(Mm), F::dbl(S));   // < 3M
  r.y = fmul_sub<F, M>(Mm, fsub<F, 3 * M>(S, r.x), y, W);  // Mm*(S-X3) - y*W, < 2M
  r.zz = V;
  r.zzz = W;
  return r;
}

template <class F>
CTT_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
  constexpr int M = F::MULB;
  if (p.is_inf()) return p;
  F U = F::dbl(p.y);                             // < 4M
  F V, xx, W, S;
  fsqr_pair<F>(U, p.x, V, xx);                   // V = U^2, xx = X^2
  fmul_pair<F>(U, V, p.x, V, W, S);              // W = U*V, S = X*V
  F Mm = F::add(F::dbl(xx), xx);                 // < 3M
  XYZZ<F> r;
  r.x = fsub<F, 2 * M>(F::sqr(Mm), F::dbl(S));   // < 3M
  r.y = fmul_sub<F, 2 * M>(Mm, fsub<F, 3 * M>(S, r.x), p.y, W);  // Mm*(S-X3) - Y*W, < 2M
  fmul_pair<F>(V, p.zz, W, p.zzz, r.zz, r.zzz);
  return r;
}

// exceptional case of the mixed addition: same x. equal -> doubling of the affine point, opposite -> neutral.
// Inlined on purpose: a device function call needs a stack frame in scratch memory, and any kernel whose scratch
// demand (bytes/lane x resident lanes) crosses the runtime's per-queue limit is dispatched through the slow
// allocate-per-launch path (measured: 2x on k_merge_tail, 2.5x on k_gen_points, box dependent).
template <class F>
CTT_HD XYZZ<F> xyzz_madd_same_x(const F& qx, const F& qy, bool same_y) {
  if (same_y) return xyzz_mdbl<F>(qx, qy);
  return XYZZ<F>::inf();
}

// acc += (neg ? -q : q), q affine -- the hot function (one call per (window, pair)).
// Bounds below are in units of p.  In: q.x, q.y < 2; acc.x < XB = 9, acc.y < 4, acc.zz, acc.zzz < 2.  Out: the same.
// Values that only feed products are left lazy (fpu.h "lazy forms") where the field's column budget allows it:
// L1 = a lazy operand against a normalised one, L2 = lazy against lazy (28-bit limbs only).
static constexpr int XYZZ_XB = 9;  // stored X < 9p (xyzz_madd's X3 = RR - PPP - 2Q + 7p); the other formulas give < 8p
// `empty` carries "acc is the neutral element" in a flag instead of acc.zz == 0 (the accumulate kernel's form: no
// 14-limb zero test per addition and no zero-fill when a bucket is flushed); when it is set acc's limbs are unspecified.
// `neg`: bool, or SignMask (fpu.h) -- the sign as a lane mask word, negated in arithmetic instead of through v_cndmask_b32
template <class F, class S = bool>
CTT_HD void xyzz_madd_flag(XYZZ<F>& acc, bool& empty, const F& qx, const F& qy_in, S neg) {
  constexpr int M = F::MULB;                                  // a product is < M*p, M = 2
  constexpr bool L1 = LazyOps<F>::ONE, L2 = LazyOps<F>::BOTH;
  constexpr int XB = XYZZ_XB;
  if (empty) {
    acc.x = qx;
    acc.y = fcneg<F, M>(qy_in, neg);                          // normalised, < 2
    acc.zz = F::one();
    acc.zzz = F::one();
    empty = false;
    return;
  }
  F qy = fcneg_lz<F, M, L1>(qy_in, neg);                      // < 3, lazy: feeds S2 only
  F U2, S2;
  fmul_pair<F>(qx, acc.zz, qy, acc.zzz, U2, S2);              // 2*2, 3*2
  F P = fsub_lz<F, XB, L2>(U2, acc.x);                        // < 2 + 10 = 12; feeds P^2, P*PP
  F R = fsub_lz<F, 2 * M, L2>(S2, acc.y);                     // < 2 + 5 = 7;   feeds R^2, R*T
  if (fis_zero_modp<F, M + XB + 1>(P)) {                      // P == +-Q: rare, out of line
    acc = xyzz_madd_same_x<F>(qx, fcneg<F, M>(qy_in, neg), fis_zero_modp<F, 3 * M + 1>(R));
    empty = acc.is_inf();
    return;
  }
  if constexpr (IsFp2<F>::value) {
    // quadratic extension: an element is 2 x 14 limbs, and the order of the products is the order that keeps the fewest
    // of them alive (every operand dies as early as it can: the kernel has 256 registers per lane and spills beyond)
    F PP = F::sqr(P);
    F PPP = F::mul(P, PP);                                    // P dead
    F Q = F::mul(acc.x, PP);                                  // X1 dead
    F X3 = fsub3<F, 7>(F::sqr(R), PPP, Q);                    // RR - PPP - 2Q + 7p < 9
    F T = fsub<F, XB>(Q, X3);                                 // Q dead
    acc.y = fmul_sub<F, 2 * M>(R, T, acc.y, PPP);             // R, T, Y1 dead
    acc.x = X3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
    return;
  }
  F PP, RR, PPP, Q;
  fsqr_pair<F>(P, R, PP, RR);                                 // 144, 49   (121, 36 when P, R are normalised: < 128)
  fmul_pair<F>(P, PP, acc.x, PP, PPP, Q);                     // 24, 18
  F X3 = fsub3<F, 7>(RR, PPP, Q);                             // RR - PPP - 2Q + 7p < 9   (PPP + 2Q < 6)
  F T = fsub_lz<F, XB, L1>(Q, X3);                            // Q - X3 + 10p < 12; feeds R*T (R lazy only when L2)
  F Y3 = fmul_sub_lz<F, 2 * M, L1>(R, T, acc.y, PPP);         // R*T - Y1*PPP: 7*12 + 5*2 = 94; < 2
  acc.x = X3;
  acc.y = Y3;
  F Z2, Z3;
  fmul_pair<F>(acc.zz, PP, acc.zzz, PPP, Z2, Z3);
  acc.zz = Z2;
  acc.zzz = Z3;
}

// The same addition with the accumulator held as separate X, Y and a ZZ/ZZZ holder (get / put) -- the form the accumulate
// kernel uses for the quadratic-extension fields.  It is the same arithmetic as xyzz_madd_flag; what differs is what hipcc's
// register allocator makes of it at the 256-register limit: BLS12-381 G2 accumulates in 7.88 ms per 2^20 pairs in this form
// against 8.76 ms in the struct form, while the base fields lose 104 instructions per addition in it (4636 against 4532,
// SQ_INSTS_VALU), so each kind of field gets the form it is faster in (msm_bodies.h accum_body).
template <class F>
struct ZInRegs {
  F zz, zzz;
  CTT_HD void get(F& a, F& b) const { a = zz; b = zzz; }
  CTT_HD void put(const F& a, const F& b) { zz = a; zzz = b; }
};
template <class F, class Z>
CTT_HD void xyzz_madd_core(F& X, F& Y, Z& z, bool& empty, const F& qx, const F& qy_in, bool neg) {
  constexpr int M = F::MULB;                                  // a product is < M*p, M = 2
  constexpr bool L1 = LazyOps<F>::ONE, L2 = LazyOps<F>::BOTH;
  constexpr int XB = XYZZ_XB;
  if (empty) {
    X = qx;
    Y = fcneg<F, M>(qy_in, neg);                              // normalised, < 2
    z.put(F::one(), F::one());
    empty = false;
    return;
  }
  F qy = fcneg_lz<F, M, L1>(qy_in, neg);                      // < 3, lazy: feeds S2 only
  F U2, S2;
  {
    F ZZ1, ZZZ1;
    z.get(ZZ1, ZZZ1);
    fmul_pair<F>(qx, ZZ1, qy, ZZZ1, U2, S2);                  // 2*2, 3*2
  }
  F P = fsub_lz<F, XB, L2>(U2, X);                            // < 2 + 10 = 12; feeds P^2, P*PP
  F R = fsub_lz<F, 2 * M, L2>(S2, Y);                         // < 2 + 5 = 7;   feeds R^2, R*T
  if (fis_zero_modp<F, M + XB + 1>(P)) {                      // P == +-Q: rare, out of line
    const XYZZ<F> r = xyzz_madd_same_x<F>(qx, fcneg<F, M>(qy_in, neg), fis_zero_modp<F, 3 * M + 1>(R));
    X = r.x;
    Y = r.y;
    z.put(r.zz, r.zzz);
    empty = r.is_inf();
    return;
  }
  if constexpr (IsFp2<F>::value) {
    // quadratic extension: an element is 2 x 14 limbs, and the order of the products is the order that keeps the fewest
    // of them alive (every operand dies as early as it can: the kernel has 256 registers per lane and spills beyond)
    F PP = F::sqr(P);
    F PPP = F::mul(P, PP);                                    // P dead
    F Q = F::mul(X, PP);                                      // X1 dead
    F X3 = fsub3<F, 7>(F::sqr(R), PPP, Q);                    // RR - PPP - 2Q + 7p < 9
    F T = fsub<F, XB>(Q, X3);                                 // Q dead
    Y = fmul_sub<F, 2 * M>(R, T, Y, PPP);                     // R, T, Y1 dead
    X = X3;
    F ZZ1, ZZZ1;
    z.get(ZZ1, ZZZ1);
    z.put(F::mul(ZZ1, PP), F::mul(ZZZ1, PPP));
    return;
  }
  F PP, RR, PPP, Q;
  fsqr_pair<F>(P, R, PP, RR);                                 // 144, 49   (121, 36 when P, R are normalised: < 128)
  fmul_pair<F>(P, PP, X, PP, PPP, Q);                         // 24, 18
  F X3 = fsub3<F, 7>(RR, PPP, Q);                             // RR - PPP - 2Q + 7p < 9   (PPP + 2Q < 6)
  F T = fsub_lz<F, XB, L1>(Q, X3);                            // Q - X3 + 10p < 12; feeds R*T (R lazy only when L2)
  F Y3 = fmul_sub_lz<F, 2 * M, L1>(R, T, Y, PPP);             // R*T - Y1*PPP: 7*12 + 5*2 = 94; < 2
  X = X3;
  Y = Y3;
  F ZZ1, ZZZ1, Z2, Z3;
  z.get(ZZ1, ZZZ1);
  fmul_pair<F>(ZZ1, PP, ZZZ1, PPP, Z2, Z3);
  z.put(Z2, Z3);
}

template <class F>
CTT_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& q, bool neg) {
  if (q.is_inf()) return;
  bool empty = acc.is_inf();
  xyzz_madd_flag<F>(acc, empty, q.x, q.y, neg);
  if (empty) acc = XYZZ<F>::inf();
}

// acc += q, both XYZZ; returns by value so that neither operand has its address taken.  Bounds in units of p: in X < 9, Y < 4,
// ZZ, ZZZ < 2; out the same.  Lazy operands as in xyzz_madd_flag.
template <class F>
CTT_HD XYZZ<F> xyzz_add_inl(const XYZZ<F>& a, const XYZZ<F>& q) {
  constexpr int M = F::MULB;
  constexpr bool L1 = LazyOps<F>::ONE, L2 = LazyOps<F>::BOTH;
  constexpr int XB = XYZZ_XB;
  if (q.is_inf()) return a;
  if (a.is_inf()) return q;
  F U1, U2, S1, S2;
  fmul_pair<F>(a.x, q.zz, q.x, a.zz, U1, U2);                 // 9*2
  fmul_pair<F>(a.y, q.zzz, q.y, a.zzz, S1, S2);               // 4*2
  F P = fsub_lz<F, M, L2>(U2, U1);                            // < 2 + 3 = 5
  F R = fsub_lz<F, M, L2>(S2, S1);                            // < 5
  if (fis_zero_modp<F, 2 * M + 1>(P)) {
    if (fis_zero_modp<F, 2 * M + 1>(R)) return xyzz_dbl<F>(a);
    return XYZZ<F>::inf();
  }
  F PP, RR, PPP, Q, Z2, Z3;
  fsqr_pair<F>(P, R, PP, RR);                                 // 25, 25
  fmul_pair<F>(P, PP, U1, PP, PPP, Q);
  XYZZ<F> r;
  r.x = fsub3<F, 7>(RR, PPP, Q);                              // RR - PPP - 2Q + 7p < 9
  F T = fsub_lz<F, XB, L1>(Q, r.x);                           // < 2 + 10 = 12
  r.y = fmul_sub_lz<F, M, L1>(R, T, S1, PPP);                 // R*T - S1*PPP: 5*12 + 3*2 = 66; < 2
  fmul_pair<F>(a.zz, q.zz, a.zzz, q.zzz, Z2, Z3);
  fmul_pair<F>(Z2, PP, Z3, PPP, r.zz, r.zzz);
  return r;
}

// out-of-line form for call sites off the hot path (bounds code size / compile time)
template <class F>
CTT_HD_NOINLINE void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& q) {
  acc = xyzz_add_inl<F>(acc, q);
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
