// oracle/msm_ref.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement ("port") of the reference's variable-time multi-scalar multiplication,
// used (a) as the parity checker for the HIP engine at sizes Python is too slow for and
// (b) as bench.py's `cpu_baseline` (kind = "port": Nim is not available in this image, so
// the reference itself cannot be built -- see DESIGN.md).  Nothing in the product path
// (constantine_amd/, libctt_msm_hip.so) may link or call this file.
//
// Parity status: PINNED -- tests/test_oracle_c.py checks this file against oracle/pyoracle.py,
// which is itself checked against the reference's golden vectors (tests/golden/*).
//
// What is restated, with the reference location it follows (paths relative to the reference root):
//   Montgomery CIOS multiplication, 64-bit limbs ... constantine/math/arithmetic/limbs_montgomery.nim:180-217
//   fromMont / getMont .............................. limbs_montgomery.nim:420-441,577-648
//   modular add/sub/neg/double ....................... constantine/math/arithmetic/finite_fields.nim:172-266
//   Montgomery constants (m0ninv, R mod p, R^2) ...... constantine/named/deriv/precompute.nim:248-373
//   Fp2 = Fp[i]/(i^2+1) ............................... constantine/math/extension_fields/towers.nim:758-878
//   Jacobian double / mixed add / add (a = 0) ........ constantine/math/elliptic/ec_shortweierstrass_jacobian.nim:564-592,655-679,798-896
//   Booth signed windows ............................. constantine/math/arithmetic/bigints.nim:360-380,806-859
//   bestBucketBitSize ................................ constantine/math/elliptic/ec_multi_scalar_mul_scheduler.nim:172-223
//   serial signed-window Pippenger ................... constantine/math/elliptic/ec_multi_scalar_mul.nim:177-296
//   window-level + msm-level (point sharding) threads  constantine/math/elliptic/ec_multi_scalar_mul_parallel.nim:148-208,386-431,519-553
//   balanced chunking ................................ constantine/threadpool/partitioners.nim:44-77
//
//   batched-affine bucket accumulation for c >= 9 ... constantine/math/elliptic/ec_multi_scalar_mul_scheduler.nim:266-553 (scheduler,
//                                                      collision queue, sparseVectorAddition), additions ec_shortweierstrass_batch_ops.nim:424-455,
//                                                      dispatch ec_multi_scalar_mul.nim:478-490 (round 3: rounds 1-2 kept Jacobian buckets throughout,
//                                                      which understated the CPU baseline)
//
//   endomorphism pre-split on G1 (round 4) .......... constantine/math/elliptic/ec_multi_scalar_mul.nim:398-453 (applyEndomorphism, withEndo),
//                                                      dispatch :455-490 (serial: c <= 13) and ec_multi_scalar_mul_parallel.nim:519-553 (parallel:
//                                                      c in {2..6, 9, 10}); decomposition constantine/math/endomorphisms/split_scalars.nim:37-123
//                                                      (Babai rounding with the lattices of named/constants/*_endomorphisms.nim);
//                                                      phi(x, y) = (beta x, y), named/zoo_endomorphisms.nim:79-92.  BLS12-381, BN254-Snarks, Pallas,
//                                                      Vesta G1 (M = 2: N -> 2N points, mini-scalars of ceil(bits/2) + 1 bits).
//
//   endomorphism pre-split on G2 (round 5) .......... the same applyEndomorphism with M = 4 (ec_multi_scalar_mul.nim:398-432: `elif ECaff.F is Fp2: 4`):
//                                                      k -> 4 mini-scalars of ceil(bits/4) + 1 = 65 bits over (P, psi P, psi^2 P, psi^3 P), the
//                                                      untwist-Frobenius-twist psi(x, y) = (conj(x) c2, conj(y) c3) (computeEndomorphisms,
//                                                      named/zoo_endomorphisms.nim:94-104), 4 x 4 lattices and Babai vectors of
//                                                      named/constants/{bls12_381,bn254_snarks}_endomorphisms.nim:39-60 / :39-64; same dispatch as G1.
//                                                      This is the configuration of the reference's regression test for issue #366 (BN254 G2,
//                                                      N = 22529: c = 13 divides the 65-bit mini-scalars).
//
// Deliberate simplification (documented in DESIGN.md): the field inversion is a^(p-2), not the reference's division steps.  It only
// changes the operation count, never the group element returned.
//
// Build: g++ -O3 -march=native -shared -fPIC -pthread oracle/msm_ref.cpp -o oracle/libmsm_ref.so

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

typedef unsigned __int128 u128;
typedef uint64_t u64;

// ------------------------------------------------------------------------------------------
// Prime fields, N 64-bit limbs, Montgomery form
// ------------------------------------------------------------------------------------------

template <int N_>
struct FieldCtx {
  u64 p[N_];
  u64 m0inv;     // -p^-1 mod 2^64
  u64 one[N_];   // R mod p
  u64 r2[N_];    // R^2 mod p
  u64 pm2[N_];   // p - 2 (Fermat inversion exponent)
};

static void hex_to_limbs(const char* hex, u64* out, int n) {
  memset(out, 0, sizeof(u64) * n);
  size_t len = strlen(hex);
  for (size_t i = 0; i < len; i++) {
    char ch = hex[len - 1 - i];
    u64 v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : ch - 'A' + 10;
    out[i / 16] |= v << (4 * (i % 16));
  }
}

template <int N>
static inline u64 add_n(u64* r, const u64* a, const u64* b) {
  u64 c = 0;
  for (int i = 0; i < N; i++) {
    u128 s = (u128)a[i] + b[i] + c;
    r[i] = (u64)s;
    c = (u64)(s >> 64);
  }
  return c;
}
template <int N>
static inline u64 sub_n(u64* r, const u64* a, const u64* b) {
  u64 bw = 0;
  for (int i = 0; i < N; i++) {
    u128 d = (u128)a[i] - b[i] - bw;
    r[i] = (u64)d;
    bw = (u64)(d >> 64) & 1;
  }
  return bw;
}
template <int N>
static inline bool geq_n(const u64* a, const u64* b) {
  for (int i = N - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}

template <class Tag>
struct Fp {
  static constexpr int N = Tag::N;
  static FieldCtx<N> ctx;
  u64 l[N];

  static void init() {
    FieldCtx<N>& c = ctx;
    hex_to_limbs(Tag::modulus(), c.p, N);
    // m0inv by Newton iteration on 2-adic inverse (precompute.nim "negInvModWord")
    u64 inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - c.p[0] * inv;
    c.m0inv = (u64)0 - inv;
    // R mod p and R^2 mod p by repeated modular doubling of 1
    u64 t[N];
    memset(t, 0, sizeof t);
    t[0] = 1;
    for (int i = 0; i < 2 * 64 * N; i++) {
      u64 carry = add_n<N>(t, t, t);
      if (carry || geq_n<N>(t, c.p)) sub_n<N>(t, t, c.p);
      if (i == 64 * N - 1) memcpy(c.one, t, sizeof t);
    }
    memcpy(c.r2, t, sizeof t);
    u64 two[N];
    memset(two, 0, sizeof two);
    two[0] = 2;
    sub_n<N>(c.pm2, c.p, two);
  }

  static Fp zero() { Fp r; memset(r.l, 0, sizeof r.l); return r; }
  static Fp one() { Fp r; memcpy(r.l, ctx.one, sizeof r.l); return r; }
  bool is_zero() const { u64 a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
  bool operator==(const Fp& o) const { return memcmp(l, o.l, sizeof l) == 0; }

  static inline Fp add(const Fp& a, const Fp& b) {
    Fp r;
    u64 c = add_n<N>(r.l, a.l, b.l);
    if (c || geq_n<N>(r.l, ctx.p)) sub_n<N>(r.l, r.l, ctx.p);
    return r;
  }
  static inline Fp sub(const Fp& a, const Fp& b) {
    Fp r;
    if (sub_n<N>(r.l, a.l, b.l)) add_n<N>(r.l, r.l, ctx.p);
    return r;
  }
  static inline Fp neg(const Fp& a) {
    if (a.is_zero()) return a;
    Fp r;
    sub_n<N>(r.l, ctx.p, a.l);
    return r;
  }
  static inline Fp dbl(const Fp& a) { return add(a, a); }

  // CIOS, limbs_montgomery.nim:180-217
  static inline Fp mul(const Fp& a, const Fp& b) {
    u64 t[N + 2];
    memset(t, 0, sizeof t);
    const u64* p = ctx.p;
    const u64 m0 = ctx.m0inv;
    for (int i = 0; i < N; i++) {
      u64 c = 0;
      for (int j = 0; j < N; j++) {
        u128 s = (u128)a.l[j] * b.l[i] + t[j] + c;
        t[j] = (u64)s;
        c = (u64)(s >> 64);
      }
      u128 s = (u128)t[N] + c;
      t[N] = (u64)s;
      t[N + 1] = (u64)(s >> 64);
      u64 m = t[0] * m0;
      s = (u128)m * p[0] + t[0];
      c = (u64)(s >> 64);
      for (int j = 1; j < N; j++) {
        s = (u128)m * p[j] + t[j] + c;
        t[j - 1] = (u64)s;
        c = (u64)(s >> 64);
      }
      s = (u128)t[N] + c;
      t[N - 1] = (u64)s;
      t[N] = t[N + 1] + (u64)(s >> 64);
    }
    Fp r;
    if (t[N] || geq_n<N>(t, p)) sub_n<N>(r.l, t, p); else memcpy(r.l, t, sizeof r.l);
    return r;
  }
  static inline Fp sqr(const Fp& a) { return mul(a, a); }

  static Fp from_mont(const Fp& a) {  // fromMont: multiply by 1
    Fp o = zero();
    o.l[0] = 1;
    return mul(a, o);
  }
  static Fp to_mont(const Fp& a) {  // getMont: multiply by R^2
    Fp r2;
    memcpy(r2.l, ctx.r2, sizeof r2.l);
    return mul(a, r2);
  }
  static Fp inv(const Fp& a) {  // a^(p-2); same value as the reference's inv_vartime
    Fp r = one();
    for (int i = 64 * N - 1; i >= 0; i--) {
      r = sqr(r);
      if ((ctx.pm2[i / 64] >> (i % 64)) & 1) r = mul(r, a);
    }
    return r;
  }
  static Fp from_u64(u64 v) {
    Fp r = zero();
    r.l[0] = v;
    return to_mont(r);
  }
};
template <class Tag> FieldCtx<Fp<Tag>::N> Fp<Tag>::ctx;

// Fp2 = Fp[i]/(i^2+1), towers.nim:758-878
template <class F>
struct Fp2 {
  F c0, c1;
  static void init() {}
  static Fp2 zero() { return {F::zero(), F::zero()}; }
  static Fp2 one() { return {F::one(), F::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
  static Fp2 add(const Fp2& a, const Fp2& b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
  static Fp2 sub(const Fp2& a, const Fp2& b) { return {F::sub(a.c0, b.c0), F::sub(a.c1, b.c1)}; }
  static Fp2 neg(const Fp2& a) { return {F::neg(a.c0), F::neg(a.c1)}; }
  static Fp2 dbl(const Fp2& a) { return add(a, a); }
  static Fp2 mul(const Fp2& a, const Fp2& b) {  // Karatsuba, prod_complex towers.nim:818-850
    F v0 = F::mul(a.c0, b.c0), v1 = F::mul(a.c1, b.c1);
    F s = F::mul(F::add(a.c0, a.c1), F::add(b.c0, b.c1));
    return {F::sub(v0, v1), F::sub(F::sub(s, v0), v1)};
  }
  static Fp2 sqr(const Fp2& a) {  // square_complex towers.nim:758-796
    F s = F::add(a.c0, a.c1), d = F::sub(a.c0, a.c1);
    F m = F::mul(a.c0, a.c1);
    return {F::mul(s, d), F::dbl(m)};
  }
  static Fp2 inv(const Fp2& a) {
    F n = F::inv(F::add(F::sqr(a.c0), F::sqr(a.c1)));
    return {F::mul(a.c0, n), F::neg(F::mul(a.c1, n))};
  }
};

template <class T> struct IsFp2 { static constexpr bool value = false; };
template <class F> struct IsFp2<Fp2<F>> { static constexpr bool value = true; };

// ------------------------------------------------------------------------------------------
// Curves
// ------------------------------------------------------------------------------------------

template <class F>
struct Aff { F x, y; bool is_inf() const { return x.is_zero() && y.is_zero(); } };  // neutral = (0,0)

template <class F>
struct Jac {
  F x, y, z;
  static Jac inf() { return {F::one(), F::one(), F::zero()}; }  // jacobian.nim:58-64
  bool is_inf() const { return z.is_zero(); }

  static Jac from_aff(const Aff<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one()};
  }

  // dbl-2009-l, a = 0
  static Jac dbl(const Jac& p) {
    if (p.is_inf()) return p;
    F A = F::sqr(p.x), B = F::sqr(p.y), C = F::sqr(B);
    F D = F::dbl(F::sub(F::sub(F::sqr(F::add(p.x, B)), A), C));
    F E = F::add(F::dbl(A), A), Fq = F::sqr(E);
    Jac r;
    r.x = F::sub(Fq, F::dbl(D));
    F C8 = F::dbl(F::dbl(F::dbl(C)));
    r.z = F::dbl(F::mul(p.y, p.z));
    r.y = F::sub(F::mul(E, F::sub(D, r.x)), C8);
    return r;
  }

  // mixedSum_vartime (jacobian.nim:798-896): handles infinity, P == Q and P == -Q
  static Jac madd(const Jac& p, const Aff<F>& q, bool negq) {
    if (q.is_inf()) return p;
    F qy = negq ? F::neg(q.y) : q.y;
    if (p.is_inf()) return {q.x, qy, F::one()};
    F Z1Z1 = F::sqr(p.z);
    F U2 = F::mul(q.x, Z1Z1);
    F S2 = F::mul(F::mul(qy, p.z), Z1Z1);
    F H = F::sub(U2, p.x);
    F Rr = F::sub(S2, p.y);
    if (H.is_zero()) {
      if (Rr.is_zero()) return dbl(p);
      return inf();
    }
    F HH = F::sqr(H), HHH = F::mul(H, HH), V = F::mul(p.x, HH);
    Jac r;
    r.x = F::sub(F::sub(F::sqr(Rr), HHH), F::dbl(V));
    r.y = F::sub(F::mul(Rr, F::sub(V, r.x)), F::mul(p.y, HHH));
    r.z = F::mul(p.z, H);
    return r;
  }

  // sum_vartime (jacobian.nim:655-679 and above)
  static Jac add(const Jac& p, const Jac& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    F Z1Z1 = F::sqr(p.z), Z2Z2 = F::sqr(q.z);
    F U1 = F::mul(p.x, Z2Z2), U2 = F::mul(q.x, Z1Z1);
    F S1 = F::mul(F::mul(p.y, q.z), Z2Z2), S2 = F::mul(F::mul(q.y, p.z), Z1Z1);
    F H = F::sub(U2, U1), Rr = F::sub(S2, S1);
    if (H.is_zero()) {
      if (Rr.is_zero()) return dbl(p);
      return inf();
    }
    F HH = F::sqr(H), HHH = F::mul(H, HH), V = F::mul(U1, HH);
    Jac r;
    r.x = F::sub(F::sub(F::sqr(Rr), HHH), F::dbl(V));
    r.y = F::sub(F::mul(Rr, F::sub(V, r.x)), F::mul(S1, HHH));
    r.z = F::mul(F::mul(p.z, q.z), H);
    return r;
  }

  Aff<F> to_aff() const {
    if (is_inf()) return {F::zero(), F::zero()};
    F zi = F::inv(z), zi2 = F::sqr(zi);
    return {F::mul(x, zi2), F::mul(y, F::mul(zi2, zi))};
  }
};

// ------------------------------------------------------------------------------------------
// Scalars (BigInt[bits], 4 x u64 LE, canonical) and window recoding
// ------------------------------------------------------------------------------------------

struct Scalar { u64 l[4]; };

// getWindowAt + signedWindowEncoding, bigints.nim:360-380,806-859
static inline void booth_digit(const Scalar& k, int w, int c, uint32_t& val, bool& neg) {
  int i = w * c;
  u64 d;
  if (i == 0) {
    d = (k.l[0] << 1);
  } else {
    int pos = i - 1;
    int slot = pos >> 6, sh = pos & 63;
    d = slot < 4 ? k.l[slot] >> sh : 0;
    if (sh + c + 1 > 64 && slot + 1 < 4) d |= k.l[slot + 1] << (64 - sh);
  }
  d &= ((u64)1 << (c + 1)) - 1;
  u64 ng = d >> c;
  u64 e = (d + 1) >> 1;
  u64 v = ng ? ((u64)1 << c) - e : e;
  val = (uint32_t)(v & (((u64)1 << c) - 1));
  neg = ng != 0;
}

// scheduler.nim:172-223 (float32, like the reference)
static int best_bucket_bit_size(size_t n, int bits, bool is_signed, bool manual) {
  const float A = 10.f, D = 6.f;
  const int s = is_signed ? 1 : 0;
  float b = (float)bits, best_cost = INFINITY;
  int best = 2;
  for (int c = 2; c <= 20; c++) {
    float b_over_c = b / (float)c;
    float acc = b_over_c * (float)((double)n + (double)((u64)1 << (c - s)) - 2.0) * A;
    float fin = (b_over_c - 1.f) * ((float)c * D + A);
    float cost = acc + fin;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  if (manual) {
    if (best >= 14) best--;
    if (best >= 15) best--;
    if (best >= 16) best--;
  }
  return best;
}

// ------------------------------------------------------------------------------------------
// MSM
// ------------------------------------------------------------------------------------------

// bucketAccumReduce, ec_multi_scalar_mul.nim:204-235 (accumulate :177-184, bucketReduce :186-197)
template <class F>
static Jac<F> window_sum(const Scalar* coefs, const Aff<F>* pts, size_t n, int w, int c, Jac<F>* buckets) {
  const size_t B = (size_t)1 << (c - 1);
  for (size_t i = 0; i < B; i++) buckets[i] = Jac<F>::inf();
  for (size_t j = 0; j < n; j++) {
    uint32_t val; bool neg;
    booth_digit(coefs[j], w, c, val, neg);
    if (val == 0) continue;
    buckets[val - 1] = Jac<F>::madd(buckets[val - 1], pts[j], neg);
  }
  Jac<F> acc = buckets[B - 1], r = buckets[B - 1];
  for (size_t k = B - 1; k-- > 0;) {
    acc = Jac<F>::add(acc, buckets[k]);
    r = Jac<F>::add(r, acc);
  }
  return r;
}

// ------------------------------------------------------------------------------------------
// Affine buckets with a scheduler (the reference's choice for c >= 9, ec_multi_scalar_mul.nim:478-490): points are queued
// per bucket, at most one pending addition per bucket in a batch (a second one waits in the collision queue), and a full
// queue is flushed as ONE vector of affine additions sharing a single inversion (sparseVectorAddition): 6 field
// multiplications per addition + a share of the inversion, against 11 for a Jacobian mixed addition.
// ------------------------------------------------------------------------------------------
template <class F>
struct AffineBuckets {
  struct Sched { uint32_t bucket; uint32_t point; bool neg; };
  std::vector<Aff<F>> pt;          // bucket sums (valid where full[b])
  std::vector<uint8_t> full, busy;  // bucket holds a point; bucket has an addition pending in the queue
  std::vector<Sched> queue, collisions;
  std::vector<F> num, den, pre;    // per queued addition: lambda numerator / denominator, running products
  size_t qlen;
  const Aff<F>* points = nullptr;

  explicit AffineBuckets(int c) {
    const size_t B = (size_t)1 << (c - 1);
    pt.resize(B);
    full.assign(B, 0);
    busy.assign(B, 0);
    const long ql = 4L * c * c - 16L * c - 128;   // deriveSchedulerConstants, scheduler.nim:266-269
    qlen = (size_t)(ql > 32 ? ql : 32);
    queue.reserve(qlen);
    collisions.reserve(qlen);
    num.resize(qlen);
    den.resize(qlen);
    pre.resize(qlen);
  }
  void reset(const Aff<F>* p) {
    points = p;
    std::fill(full.begin(), full.end(), 0);
    std::fill(busy.begin(), busy.end(), 0);
    queue.clear();
    collisions.clear();
  }
  // one vector of affine additions bucket[b] += +-P (affine chord-and-tangent with Montgomery's simultaneous inversion;
  // P == bucket doubles, P == -bucket empties the bucket: the special cases of scheduler.nim:414-553)
  void flush() {
    const size_t m = queue.size();
    F run = F::one();
    for (size_t i = 0; i < m; i++) {
      const Sched& q = queue[i];
      const Aff<F>& b = pt[q.bucket];
      const Aff<F>& p = points[q.point];
      const F py = q.neg ? F::neg(p.y) : p.y;
      F dx = F::sub(p.x, b.x);
      if (dx.is_zero()) {
        if (py == b.y && !py.is_zero()) {             // doubling: lambda = 3 x^2 / 2 y
          F xx = F::sqr(b.x);
          num[i] = F::add(F::dbl(xx), xx);
          den[i] = F::dbl(b.y);
        } else {                                      // opposite points (or a point of order two): the sum is the neutral element
          num[i] = F::zero();
          den[i] = F::zero();
        }
      } else {
        num[i] = F::sub(py, b.y);
        den[i] = dx;
      }
      pre[i] = run;
      if (!den[i].is_zero()) run = F::mul(run, den[i]);
    }
    F inv = F::inv(run);
    for (size_t i = m; i-- > 0;) {
      const Sched& q = queue[i];
      busy[q.bucket] = 0;
      if (den[i].is_zero()) {
        full[q.bucket] = 0;
        continue;
      }
      const F dinv = F::mul(inv, pre[i]);
      inv = F::mul(inv, den[i]);
      const F lam = F::mul(num[i], dinv);
      Aff<F>& b = pt[q.bucket];
      const Aff<F>& p = points[q.point];
      const F x3 = F::sub(F::sub(F::sqr(lam), b.x), p.x);
      const F y3 = F::sub(F::mul(lam, F::sub(b.x, x3)), b.y);
      b.x = x3;
      b.y = y3;
    }
    queue.clear();
    // the collisions of this batch are scheduled again (rescheduleCollisions)
    std::vector<Sched> again;
    again.swap(collisions);
    for (const Sched& q : again) schedule(q);
  }
  void schedule(const Sched& q) {
    const Aff<F>& p = points[q.point];
    if (p.is_inf()) return;
    if (!full[q.bucket] && !busy[q.bucket]) {          // empty bucket: the point moves in, no addition
      pt[q.bucket] = {p.x, q.neg ? F::neg(p.y) : p.y};
      full[q.bucket] = 1;
      return;
    }
    if (busy[q.bucket]) {                             // a second addition to the same bucket must wait for the first
      collisions.push_back(q);
      if (collisions.size() >= qlen) flush();
      return;
    }
    busy[q.bucket] = 1;
    queue.push_back(q);
    if (queue.size() >= qlen) flush();
  }
  void finish() { while (!queue.empty() || !collisions.empty()) flush(); }
};

// miniMSM_affine, ec_multi_scalar_mul.nim:329-345: schedAccumulate, then bucketReduce over the affine buckets
template <class F>
static Jac<F> window_sum_affine(const Scalar* coefs, const Aff<F>* pts, size_t n, int w, int c, AffineBuckets<F>& bk) {
  const size_t B = (size_t)1 << (c - 1);
  bk.reset(pts);
  for (size_t j = 0; j < n; j++) {
    uint32_t val; bool neg;
    booth_digit(coefs[j], w, c, val, neg);
    if (val == 0) continue;
    bk.schedule({val - 1, (uint32_t)j, neg});
  }
  bk.finish();
  Jac<F> acc = Jac<F>::inf(), r = Jac<F>::inf();
  for (size_t k = B; k-- > 0;) {
    if (bk.full[k]) acc = Jac<F>::madd(acc, bk.pt[k], false);
    r = Jac<F>::add(r, acc);
  }
  return r;
}
static const int AFFINE_BUCKETS_FROM_C = getenv("ORACLE_JACOBIAN_BUCKETS") ? 99 : 9;   // ec_multi_scalar_mul.nim:478-490 ($ORACLE_JACOBIAN_BUCKETS: rounds 1-2's form, for A/B)

// msmImpl_vartime, ec_multi_scalar_mul.nim:256-296 (serial: windows top -> bottom, one bucket array)
template <class F>
static Jac<F> msm_serial(const Scalar* coefs, const Aff<F>* pts, size_t n, int bits, int c) {
  const int W = bits / c + 1;
  const bool affine = c >= AFFINE_BUCKETS_FROM_C && n < (1ull << 32);
  std::vector<Jac<F>> buckets((size_t)1 << (c - 1));
  AffineBuckets<F> ab(affine ? c : 2);
  Jac<F> r = Jac<F>::inf();
  for (int w = W - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) r = Jac<F>::dbl(r);
    // the top window has fewer bits, hence few buckets and a collision per point in the scheduler: the reference keeps the
    // non-affine accumulation for it (ec_multi_scalar_mul.nim:364-372: kTopWindow -> bucketAccumReduce)
    const bool aff_w = affine && w != W - 1;
    r = Jac<F>::add(r, aff_w ? window_sum_affine<F>(coefs, pts, n, w, c, ab) : window_sum<F>(coefs, pts, n, w, c, buckets.data()));
  }
  return r;
}

// ec_multi_scalar_mul_parallel.nim:148-208 (one task per window) x :386-431 (msm-level split)
template <class F>
static Jac<F> msm_parallel(const Scalar* coefs, const Aff<F>* pts, size_t n, int bits, int c, int nthreads) {
  const int W = bits / c + 1;
  int winpar = bits / c, msmpar = 1;
  while ((long)winpar * msmpar < nthreads) msmpar <<= 1;
  if ((size_t)msmpar > n) msmpar = 1;
  // balancedChunksPrioNumber, partitioners.nim:44-77
  std::vector<size_t> start(msmpar + 1);
  size_t base = n / msmpar, cutoff = n % msmpar;
  start[0] = 0;
  for (int i = 0; i < msmpar; i++) start[i + 1] = start[i] + base + ((size_t)i < cutoff ? 1 : 0);

  const int ntasks = msmpar * W;
  std::vector<Jac<F>> sums(ntasks);
  std::atomic<int> next(0);
  const bool affine = c >= AFFINE_BUCKETS_FROM_C && n < (1ull << 32);
  auto worker = [&]() {
    std::vector<Jac<F>> buckets((size_t)1 << (c - 1));
    AffineBuckets<F> ab(affine ? c : 2);
    for (;;) {
      int t = next.fetch_add(1);
      if (t >= ntasks) break;
      int chunk = t / W, w = W - 1 - (t % W);
      const size_t cn = start[chunk + 1] - start[chunk];
      // (the top window keeps the Jacobian accumulation, as in the reference: few buckets, a collision per point)
      sums[chunk * W + w] = (affine && w != W - 1) ? window_sum_affine<F>(coefs + start[chunk], pts + start[chunk], cn, w, c, ab)
                                                   : window_sum<F>(coefs + start[chunk], pts + start[chunk], cn, w, c, buckets.data());
    }
  };
  std::vector<std::thread> th;
  for (int i = 1; i < nthreads; i++) th.emplace_back(worker);
  worker();
  for (auto& t : th) t.join();

  Jac<F> total = Jac<F>::inf();
  for (int ch = 0; ch < msmpar; ch++) {
    Jac<F> r = Jac<F>::inf();
    for (int w = W - 1; w >= 0; w--) {
      for (int k = 0; k < c; k++) r = Jac<F>::dbl(r);
      r = Jac<F>::add(r, sums[ch * W + w]);
    }
    total = Jac<F>::add(total, r);
  }
  return total;
}

template <class F>
static Jac<F> scalar_mul(const Scalar& k, const Aff<F>& p) {
  Jac<F> r = Jac<F>::inf();
  for (int i = 255; i >= 0; i--) {
    r = Jac<F>::dbl(r);
    if ((k.l[i / 64] >> (i % 64)) & 1) r = Jac<F>::madd(r, p, false);
  }
  return r;
}

// ------------------------------------------------------------------------------------------
// Endomorphism pre-split on G1 (a = 0 curves: phi(x, y) = (beta x, y) = [lambda](x, y))
// ------------------------------------------------------------------------------------------
// 256-bit two's-complement helpers (the mini-scalars are ~128 bits; everything wraps mod 2^256 like the reference's BigInt[frBits])
struct U256 { u64 l[4]; };
static inline U256 u256_from(const u64* w, int n) { U256 r{{0, 0, 0, 0}}; for (int i = 0; i < n && i < 4; i++) r.l[i] = w[i]; return r; }
static inline U256 u256_add(const U256& a, const U256& b) { U256 r; add_n<4>(r.l, a.l, b.l); return r; }
static inline U256 u256_sub(const U256& a, const U256& b) { U256 r; sub_n<4>(r.l, a.l, b.l); return r; }
static inline U256 u256_neg(const U256& a) { U256 z{{0, 0, 0, 0}}; return u256_sub(z, a); }
// a (na limbs) * b (nb limbs), limbs [lo, lo + 4) of the exact product
static inline U256 mul_limbs(const u64* a, int na, const u64* b, int nb, int lo) {
  u64 t[12];
  memset(t, 0, sizeof t);
  for (int i = 0; i < na; i++) {
    u64 c = 0;
    for (int j = 0; j < nb; j++) {
      u128 x = (u128)a[i] * b[j] + t[i + j] + c;
      t[i + j] = (u64)x;
      c = (u64)(x >> 64);
    }
    t[i + nb] += c;
  }
  U256 r{{0, 0, 0, 0}};
  for (int i = 0; i < 4 && lo + i < na + nb; i++) r.l[i] = t[lo + i];
  return r;
}
struct EndoG1 {
  bool enabled = false;
  u64 babai[2][3];      // (BigInt, isNeg) of named/constants/<curve>_endomorphisms.nim: <= 130 bits
  bool babai_neg[2];
  u64 lat[2][2][2];     // lattice[basis][miniscalar]: <= 128 bits
  bool lat_neg[2][2];
};
static void hex3(const char* h, u64* out, int n) { hex_to_limbs(h, out, n); }
// decomposeEndo (split_scalars.nim:37-123): alphas[i] = high words of babai[i] * k (w = 4 words), k_0 = k -+ sum alpha_i b_i0,
// k_1 = -+ sum alpha_i b_i1; a negative mini-scalar is negated and its point with it (:112-123)
static void decompose_endo(const EndoG1& E, const Scalar& k, Scalar mini[2], bool neg[2]) {
  U256 alpha[2];
  for (int i = 0; i < 2; i++) alpha[i] = mul_limbs(E.babai[i], 3, k.l, 4, 4);   // (babai_i * k) >> 256
  U256 kk[2] = {u256_from(k.l, 4), U256{{0, 0, 0, 0}}};
  for (int j = 0; j < 2; j++)
    for (int i = 0; i < 2; i++) {
      if ((E.lat[i][j][0] | E.lat[i][j][1]) == 0) continue;
      const U256 ab = mul_limbs(alpha[i].l, 4, E.lat[i][j], 2, 0);
      if (E.lat_neg[i][j] != E.babai_neg[i]) kk[j] = u256_add(kk[j], ab); else kk[j] = u256_sub(kk[j], ab);
    }
  for (int j = 0; j < 2; j++) {
    neg[j] = (kk[j].l[3] >> 63) != 0;
    if (neg[j]) kk[j] = u256_neg(kk[j]);
    memcpy(mini[j].l, kk[j].l, sizeof(mini[j].l));
  }
}
// applyEndomorphism (ec_multi_scalar_mul.nim:398-432): (k_i, P_i) -> (k_i0, +-P_i), (k_i1, +-phi(P_i)), interleaved as there
template <class F>
static void apply_endomorphism(const EndoG1& E, const F& beta, const Scalar* coefs, const Aff<F>* pts, size_t n,
                               std::vector<Scalar>& ec, std::vector<Aff<F>>& ep) {
  ec.resize(2 * n);
  ep.resize(2 * n);
  for (size_t i = 0; i < n; i++) {
    bool neg[2];
    decompose_endo(E, coefs[i], &ec[2 * i], neg);
    Aff<F> p0 = pts[i], p1 = pts[i];
    p1.x = F::mul(p1.x, beta);        // the neutral (0,0) stays (0,0)
    if (neg[0]) p0.y = F::neg(p0.y);
    if (neg[1]) p1.y = F::neg(p1.y);
    ep[2 * i] = p0;
    ep[2 * i + 1] = p1;
  }
}
template <class F> struct EndoFor { static const EndoG1* get(int) { return nullptr; } static F beta(int) { return F::zero(); } };

// ------------------------------------------------------------------------------------------
// Endomorphism pre-split on G2 (M = 4): psi = twist^-1 o Frobenius o twist acts on the order-r subgroup as [p mod r]
// ------------------------------------------------------------------------------------------
struct EndoG2 {
  u64 babai[4][4];      // <= 193 bits
  bool babai_neg[4];
  u64 lat[4][4];        // lattice[basis][miniscalar]: <= 64 bits
  bool lat_neg[4][4];
};
// decomposeEndo with M = 4 (split_scalars.nim:37-123): alphas[i] = high words of babai[i] * k (w = 4 words),
// k_j = [j == 0] k -+ sum_i alpha_i b_ij; a negative mini-scalar is negated and its point with it
static void decompose_endo4(const EndoG2& E, const Scalar& k, Scalar mini[4], bool neg[4]) {
  U256 alpha[4];
  for (int i = 0; i < 4; i++) alpha[i] = mul_limbs(E.babai[i], 4, k.l, 4, 4);   // (babai_i * k) >> 256
  U256 kk[4] = {u256_from(k.l, 4), U256{{0, 0, 0, 0}}, U256{{0, 0, 0, 0}}, U256{{0, 0, 0, 0}}};
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      if (E.lat[i][j] == 0) continue;
      const U256 ab = mul_limbs(alpha[i].l, 4, &E.lat[i][j], 1, 0);
      if (E.lat_neg[i][j] != E.babai_neg[i]) kk[j] = u256_add(kk[j], ab); else kk[j] = u256_sub(kk[j], ab);
    }
  for (int j = 0; j < 4; j++) {
    neg[j] = (kk[j].l[3] >> 63) != 0;
    if (neg[j]) kk[j] = u256_neg(kk[j]);
    memcpy(mini[j].l, kk[j].l, sizeof(mini[j].l));
  }
}
template <class F2>
struct PsiCoef { F2 c2, c3; };
template <class F2>
static F2 fp2_pow(F2 a, const u64* e, int nl) {
  F2 r = F2::one();
  for (int i = nl * 64 - 1; i >= 0; i--) {
    r = F2::sqr(r);
    if ((e[i / 64] >> (i % 64)) & 1) r = F2::mul(r, a);
  }
  return r;
}
// frobenius_psi (constantine/math/pairings... named/zoo_endomorphisms.nim:94-104 -> extension_fields frobenius): (x, y) -> (conj(x) c2, conj(y) c3)
template <class F2>
static Aff<F2> psi(const PsiCoef<F2>& C, const Aff<F2>& p) {
  if (p.is_inf()) return p;
  F2 cx{p.x.c0, decltype(p.x.c0)::neg(p.x.c1)}, cy{p.y.c0, decltype(p.y.c0)::neg(p.y.c1)};
  return {F2::mul(cx, C.c2), F2::mul(cy, C.c3)};
}
// applyEndomorphism with M = 4 (ec_multi_scalar_mul.nim:398-432): (k_i, P_i) -> (k_ij, +-psi^j P_i), j = 0..3, interleaved as there
template <class F2>
static void apply_endomorphism4(const EndoG2& E, const PsiCoef<F2>& C, const Scalar* coefs, const Aff<F2>* pts, size_t n,
                                std::vector<Scalar>& ec, std::vector<Aff<F2>>& ep) {
  ec.resize(4 * n);
  ep.resize(4 * n);
  for (size_t i = 0; i < n; i++) {
    bool neg[4];
    decompose_endo4(E, coefs[i], &ec[4 * i], neg);
    Aff<F2> q = pts[i];
    for (int j = 0; j < 4; j++) {
      if (j) q = psi<F2>(C, q);
      Aff<F2> t = q;
      if (neg[j]) t.y = F2::neg(t.y);     // the neutral (0,0) stays (0,0)
      ep[4 * i + j] = t;
    }
  }
}
template <class F> struct EndoG2For { static const EndoG2* get(int) { return nullptr; } static const PsiCoef<F>& psi_coef(int) { static const PsiCoef<F> z{}; return z; } };

// ------------------------------------------------------------------------------------------
// Curve table
// ------------------------------------------------------------------------------------------

struct BlsFpTag { static constexpr int N = 6; static const char* modulus() { return "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab"; } };
struct BlsFrTag { static constexpr int N = 4; static const char* modulus() { return "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001"; } };
struct BnFpTag { static constexpr int N = 4; static const char* modulus() { return "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47"; } };
struct BnFrTag { static constexpr int N = 4; static const char* modulus() { return "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001"; } };
struct PallasFpTag { static constexpr int N = 4; static const char* modulus() { return "40000000000000000000000000000000224698fc094cf91b992d30ed00000001"; } };
struct VestaFpTag { static constexpr int N = 4; static const char* modulus() { return "40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001"; } };

typedef Fp<BlsFpTag> BlsFp;  typedef Fp<BlsFrTag> BlsFr;
typedef Fp<BnFpTag> BnFp;    typedef Fp<BnFrTag> BnFr;
typedef Fp<PallasFpTag> PallasFp;  // = Vesta Fr
typedef Fp<VestaFpTag> VestaFp;    // = Pallas Fr

static std::atomic<bool> g_init(false);
static void ensure_init() {
  static std::atomic<int> once(0);
  int exp = 0;
  if (once.compare_exchange_strong(exp, 1)) {
    BlsFp::init(); BlsFr::init(); BnFp::init(); BnFr::init(); PallasFp::init(); VestaFp::init();
    g_init.store(true);
  } else {
    while (!g_init.load()) std::this_thread::yield();
  }
}

enum { C_BLS_G1 = 0, C_BLS_G2 = 1, C_BN_G1 = 2, C_BN_G2 = 3, C_PALLAS = 4, C_VESTA = 5 };

static int curve_bits(int id) { return (id == C_BN_G1 || id == C_BN_G2) ? 254 : 255; }

static u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  u64 z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// EndomorphismThreshold (zoo_endomorphisms.nim:124) and the two dispatch tables: serial c <= 13 (ec_multi_scalar_mul.nim:455-490),
// parallel c in {2..6, 9, 10} (ec_multi_scalar_mul_parallel.nim:519-553); c is the window size chosen for (N, bits) BEFORE the split
static bool endo_applies(int c_best, int bits, int nthreads) {
  static const bool off = getenv("ORACLE_NO_ENDO") != nullptr;
  if (off || bits < 152) return false;
  return nthreads > 1 ? ((c_best >= 2 && c_best <= 6) || c_best == 9 || c_best == 10) : c_best <= 13;
}

template <class F>
static int do_msm(int curve, const void* scalars, const void* points, size_t n, int bits, int nthreads, int c_override, void* out) {
  const Scalar* k = (const Scalar*)scalars;
  const Aff<F>* p = (const Aff<F>*)points;
  Aff<F>* o = (Aff<F>*)out;
  if (n == 0) { *o = {F::zero(), F::zero()}; return 0; }
  int c = c_override;
  bool endo = false;
  if (c <= 0) {
    c = best_bucket_bit_size(n, bits, true, true);
    endo = EndoFor<F>::get(curve) != nullptr && endo_applies(c, bits, nthreads);
    // parallel dispatch uses c-1 for c >= 11 (ec_multi_scalar_mul_parallel.nim:545-551); serial caps at 16
    if (nthreads > 1 && c >= 11) c -= 1;
    if (c > 16) c = 16;
  }
  Jac<F> r;
  if constexpr (IsFp2<F>::value) {
    const bool endo2 = c_override <= 0 && EndoG2For<F>::get(curve) != nullptr &&
                       endo_applies(best_bucket_bit_size(n, bits, true, true), bits, nthreads);
    if (endo2) {
      std::vector<Scalar> ec;
      std::vector<Aff<F>> ep;
      apply_endomorphism4<F>(*EndoG2For<F>::get(curve), EndoG2For<F>::psi_coef(curve), k, p, n, ec, ep);
      const int L = (bits + 3) / 4 + 1;   // computeEndoRecodedLength(bits, 4) (split_scalars.nim:315-316): 65
      r = nthreads > 1 ? msm_parallel<F>(ec.data(), ep.data(), 4 * n, L, c, nthreads) : msm_serial<F>(ec.data(), ep.data(), 4 * n, L, c);
      *o = r.to_aff();
      return c;
    }
  }
  if (endo) {
    std::vector<Scalar> ec;
    std::vector<Aff<F>> ep;
    apply_endomorphism<F>(*EndoFor<F>::get(curve), EndoFor<F>::beta(curve), k, p, n, ec, ep);
    const int L = (bits + 1) / 2 + 1;   // computeEndoRecodedLength (split_scalars.nim:315-316)
    r = nthreads > 1 ? msm_parallel<F>(ec.data(), ep.data(), 2 * n, L, c, nthreads) : msm_serial<F>(ec.data(), ep.data(), 2 * n, L, c);
  } else {
    r = nthreads > 1 ? msm_parallel<F>(k, p, n, bits, c, nthreads) : msm_serial<F>(k, p, n, bits, c);
  }
  *o = r.to_aff();
  return c;
}

template <class Fr>
static void fr_from_mont_arr(const void* in, void* out, size_t n) {
  const Fr* a = (const Fr*)in;
  Fr* o = (Fr*)out;
  for (size_t i = 0; i < n; i++) o[i] = Fr::from_mont(a[i]);
}
template <class Fr>
static void fr_to_mont_arr(const void* in, void* out, size_t n) {
  const Fr* a = (const Fr*)in;
  Fr* o = (Fr*)out;
  for (size_t i = 0; i < n; i++) o[i] = Fr::to_mont(a[i]);
}

// P_i = [s_i]G, s_i = 128-bit synth scalar | 1 (same definition as pyoracle.synth_point)
template <class F>
static void gen_points(const Aff<F>& G, u64 seed, size_t first, size_t n, void* out, int nthreads) {
  Aff<F>* o = (Aff<F>*)out;
  // fixed-base table: T[w][d] = d * 2^(8w) * G, w < 16, d < 256
  std::vector<Aff<F>> T(16 * 256);
  Jac<F> base = Jac<F>::from_aff(G);
  for (int w = 0; w < 16; w++) {
    Jac<F> acc = Jac<F>::inf();
    for (int d = 0; d < 256; d++) {
      T[w * 256 + d] = acc.to_aff();
      acc = Jac<F>::add(acc, base);
    }
    base = acc;  // 256 * previous base
  }
  auto work = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      u64 s[2];
      u64 sd = seed ^ 0xA5A5A5A5A5A5A5A5ull;
      s[0] = splitmix64(sd + 4 * (first + i) + 0) | 1;
      s[1] = splitmix64(sd + 4 * (first + i) + 1);
      Jac<F> r = Jac<F>::inf();
      for (int w = 0; w < 16; w++) {
        int d = (s[w / 8] >> (8 * (w % 8))) & 0xff;
        if (d) r = Jac<F>::madd(r, T[w * 256 + d], false);
      }
      o[i] = r.to_aff();
    }
  };
  if (nthreads < 1) nthreads = 1;
  std::vector<std::thread> th;
  size_t per = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; t++) {
    size_t lo = t * per, hi = lo + per > n ? n : lo + per;
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& t : th) t.join();
}

template <class F> static Aff<F> aff_from_u64(u64 x, u64 y) { return {F::from_u64(x), F::from_u64(y)}; }

// Random curve points with UNKNOWN discrete logarithms, the way the reference's benches make their inputs
// (helpers/prng_unsafe.nim:306-316 random_unsafe(ECP) -> trySetFromCoordX; benchmarks/bench_elliptic_parallel_template.nim:78-102 then
// clears the cofactor): x from a 2x-width uniform integer reduced mod p (every coordinate of x over Fp2), accepted when x^3 + b is a
// square, y = a square root, then [h] (x, y).  The stream is this file's splitmix64, not the reference's xoshiro: what matters to the
// tests built on it is that nobody knows log_G of these points -- the synthetic inputs [s_i]G of gen_points let a test compute the
// MSM as one scalar multiplication; these do not.
// Square roots: p = 3 mod 4 (BLS12-381, BN254-Snarks) a^((p+1)/4) (the reference: sqrt_p3mod4, finite_fields_square_root.nim);
// the Pasta primes have p - 1 = 2^32 t: Tonelli-Shanks (the reference's default for them, sqrt_tonelli_shanks);
// Fp2 = Fp[i]/(i^2+1) by the complex method over the base field's root (the reference: sqrt_if_square for QuadraticExt).
template <class B>
static B fp_pow(const B& a, const u64* e, int nl) {
  B r = B::one();
  for (int i = nl * 64 - 1; i >= 0; i--) {
    r = B::sqr(r);
    if ((e[i / 64] >> (i % 64)) & 1) r = B::mul(r, a);
  }
  return r;
}
template <class B>
struct SqrtCtx {
  static constexpr int N = B::N;
  bool p3mod4;
  u64 e34[N];      // (p + 1) / 4
  int S;           // p - 1 = 2^S T, T odd
  u64 T[N], Th[N]; // T, (T + 1) / 2
  B c;             // z^T for a non-residue z
  SqrtCtx() {
    const u64* P = B::ctx.p;
    p3mod4 = (P[0] & 3) == 3;
    u64 one[N] = {1}, t[N];
    add_n<N>(t, P, one);   // p + 1 does not overflow: the moduli have spare bits
    for (int i = 0; i < N; i++) e34[i] = (t[i] >> 2) | (i + 1 < N ? t[i + 1] << 62 : 0);
    u64 pm1[N];
    memcpy(pm1, P, sizeof pm1);
    pm1[0] -= 1;
    S = 0;
    memcpy(T, pm1, sizeof T);
    while (!(T[0] & 1)) {
      for (int i = 0; i < N; i++) T[i] = (T[i] >> 1) | (i + 1 < N ? T[i + 1] << 63 : 0);
      S++;
    }
    add_n<N>(t, T, one);
    for (int i = 0; i < N; i++) Th[i] = (t[i] >> 1) | (i + 1 < N ? t[i + 1] << 63 : 0);
    u64 half[N];                                        // (p - 1) / 2: Euler's criterion
    for (int i = 0; i < N; i++) half[i] = (pm1[i] >> 1) | (i + 1 < N ? pm1[i + 1] << 63 : 0);
    const B minus1 = B::neg(B::one());
    for (u64 z = 2;; z++) {
      const B zz = B::from_u64(z);
      if (fp_pow<B>(zz, half, N) == minus1) { c = fp_pow<B>(zz, T, N); break; }
    }
  }
  static const SqrtCtx& get() { static const SqrtCtx k; return k; }
};
template <class B>
static bool f_sqrt(const B& a, B& r) {
  const SqrtCtx<B>& K = SqrtCtx<B>::get();
  if (a.is_zero()) { r = a; return true; }
  if (K.p3mod4) {
    r = fp_pow<B>(a, K.e34, B::N);
    return B::sqr(r) == a;
  }
  B x = fp_pow<B>(a, K.Th, B::N), t = fp_pow<B>(a, K.T, B::N), c = K.c;
  int m = K.S;
  const B one = B::one();
  while (!(t == one)) {
    int i = 0;
    B tt = t;
    while (!(tt == one)) { tt = B::sqr(tt); i++; if (i == m) return false; }   // not a square
    B b = c;
    for (int k = 0; k < m - i - 1; k++) b = B::sqr(b);
    x = B::mul(x, b);
    c = B::sqr(b);
    t = B::mul(t, c);
    m = i;
  }
  r = x;
  return B::sqr(r) == a;
}
template <class B>
static bool f_sqrt(const Fp2<B>& a, Fp2<B>& r) {
  if (a.c1.is_zero()) {
    B s;
    if (f_sqrt<B>(a.c0, s)) { r = {s, B::zero()}; return true; }
    if (f_sqrt<B>(B::neg(a.c0), s)) { r = {B::zero(), s}; return true; }   // i^2 = -1
    return false;
  }
  B s;
  if (!f_sqrt<B>(B::add(B::sqr(a.c0), B::sqr(a.c1)), s)) return false;      // the norm of a square is a square
  const B inv2 = B::inv(B::from_u64(2));
  for (int k = 0; k < 2; k++) {
    const B d = B::mul(k == 0 ? B::add(a.c0, s) : B::sub(a.c0, s), inv2);
    B x0;
    if (!f_sqrt<B>(d, x0) || x0.is_zero()) continue;
    const B x1 = B::mul(a.c1, B::inv(B::dbl(x0)));
    const Fp2<B> cand{x0, x1};
    if (Fp2<B>::sqr(cand) == a) { r = cand; return true; }
  }
  return false;
}
// one coordinate from a 2x-width uniform integer: (hi * 2^(64 N) + lo) mod p
template <class B>
static B draw_fp(u64 key, u64& ctr) {
  constexpr int N = B::N;
  B xl, xh;
  for (int q = 0; q < N; q++) xl.l[q] = splitmix64(key + ctr++);
  for (int q = 0; q < N; q++) xh.l[q] = splitmix64(key + ctr++);
  while (geq_n<N>(xl.l, B::ctx.p)) sub_n<N>(xl.l, xl.l, B::ctx.p);
  while (geq_n<N>(xh.l, B::ctx.p)) sub_n<N>(xh.l, xh.l, B::ctx.p);
  // to_mont(lo) = lo R, to_mont(hi) = hi R; x R = lo R + mont_mul(hi R, R^2) (to_mont(one()) = R^2 as a residue)
  return B::add(B::to_mont(xl), B::mul(B::to_mont(xh), B::to_mont(B::one())));
}
template <class B> static void draw_x(B& x, u64 key, u64& ctr) { x = draw_fp<B>(key, ctr); }
template <class B> static void draw_x(Fp2<B>& x, u64 key, u64& ctr) { x.c0 = draw_fp<B>(key, ctr); x.c1 = draw_fp<B>(key, ctr); }

// [k]P, k = cof_limbs little-endian words, plain double-and-add
template <class F>
static Jac<F> mul_words(const Aff<F>& P, const u64* k, int limbs) {
  Jac<F> r = Jac<F>::inf();
  for (int bit = 64 * limbs - 1; bit >= 0; bit--) {
    r = Jac<F>::dbl(r);
    if ((k[bit / 64] >> (bit % 64)) & 1) r = Jac<F>::madd(r, P, false);
  }
  return r;
}
template <class F>
static Jac<F> mul_words(const Jac<F>& P, const u64* k, int limbs) {
  Jac<F> r = Jac<F>::inf();
  for (int bit = 64 * limbs - 1; bit >= 0; bit--) {
    r = Jac<F>::dbl(r);
    if ((k[bit / 64] >> (bit % 64)) & 1) r = Jac<F>::add(r, P);
  }
  return r;
}
template <class F> static Jac<F> jac_neg(const Jac<F>& p) { return {p.x, F::neg(p.y), p.z}; }

// clear: (x, y) on the curve -> a point of the prime-order subgroup (Jacobian)
template <class F, class Clear>
static void gen_points_unknown_log(const F& b, Clear clear, u64 seed, size_t first, size_t n, void* out, int nthreads) {
  Aff<F>* o = (Aff<F>*)out;
  auto work = [&](size_t lo_i, size_t hi_i) {
    for (size_t i = lo_i; i < hi_i; i++) {
      const u64 key = seed * 0x9E3779B97F4A7C15ull + (first + i) * 4096;
      u64 ctr = 0;
      for (;;) {
        F X;
        draw_x(X, key, ctr);
        const F rhs = F::add(F::mul(F::sqr(X), X), b);
        F y;
        if (!f_sqrt(rhs, y)) continue;              // not a square: next candidate
        const Aff<F> P{X, y};
        const Jac<F> r = clear(P);
        if (r.is_inf()) continue;                   // (a point of the cofactor's torsion: next candidate)
        o[i] = r.to_aff();
        break;
      }
    }
  };
  if (nthreads < 1) nthreads = 1;
  std::vector<std::thread> th;
  size_t per = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; t++) {
    size_t lo = t * per, hi = lo + per > n ? n : lo + per;
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& t : th) t.join();
}

static BlsFp fp_hex(const char* h) { BlsFp r; hex_to_limbs(h, r.l, 6); return BlsFp::to_mont(r); }
static BnFp bn_hex(const char* h) { BnFp r; hex_to_limbs(h, r.l, 4); return BnFp::to_mont(r); }

// lattices, Babai coefficients and cube roots of unity of named/constants/{bls12_381,bn254_snarks,pallas,vesta}_endomorphisms.nim
struct EndoTable {
  EndoG1 e[6];
  BlsFp beta_bls;
  BnFp beta_bn;
  PallasFp beta_pallas;
  VestaFp beta_vesta;
};
static EndoG1 make_endo(const char* l00, bool n00, const char* l01, bool n01, const char* l10, bool n10, const char* l11, bool n11,
                        const char* b0, bool bn0, const char* b1, bool bn1) {
  EndoG1 e;
  e.enabled = true;
  hex3(l00, e.lat[0][0], 2); hex3(l01, e.lat[0][1], 2); hex3(l10, e.lat[1][0], 2); hex3(l11, e.lat[1][1], 2);
  e.lat_neg[0][0] = n00; e.lat_neg[0][1] = n01; e.lat_neg[1][0] = n10; e.lat_neg[1][1] = n11;
  hex3(b0, e.babai[0], 3); hex3(b1, e.babai[1], 3);
  e.babai_neg[0] = bn0; e.babai_neg[1] = bn1;
  return e;
}
static const EndoTable& endo_table() {
  static const EndoTable t = []() {
    EndoTable t;
    t.e[C_BLS_G1] = make_endo("ac45a4010001a4020000000100000000", false, "1", false, "1", false, "ac45a4010001a40200000000ffffffff", true,
                              "17c6becf1e01faadd63f6e522f6cfee2e", false, "2", false);
    t.e[C_BN_G1] = make_endo("6f4d8248eeb859fc8211bbeb7d4f1128", false, "89d3256894d213e3", true, "89d3256894d213e3", true,
                             "6f4d8248eeb859fd0be4e1541221250b", true, "24ccef014a773d2d25398fd0300ff6565", false, "2d91d232ec7e0b3d7", true);
    t.e[C_PALLAS] = make_endo("49e69d1640a899538cb1279300000000", true, "49e69d1640f049157fcae1c700000001", false,
                              "93cd3a2c8198e2690c7c095a00000001", false, "49e69d1640a899538cb1279300000000", false,
                              "1279a745902a2654e32c49e4bffffffff", true, "1279a745903c12455ff2b871c00000003", false);
    t.e[C_VESTA] = make_endo("49e69d1640a899538cb1279300000001", true, "49e69d1640f049157fcae1c700000000", false,
                             "93cd3a2c8198e2690c7c095a00000001", false, "49e69d1640a899538cb1279300000001", false,
                             "1279a745902a2654e32c49e4c00000003", true, "1279a745903c12455ff2b871bffffffff", false);
    t.beta_bls = fp_hex("5f19672fdf76ce51ba69c6076a0f77eaddb3a93be6f89688de17d813620a00022e01fffffffefffe");
    t.beta_bn = bn_hex("30644e72e131a0295e6dd9e7e0acccb0c28f069fbb966e3de4bd44e5607cfd48");
    { PallasFp r; hex_to_limbs("2d33357cb532458ed3552a23a8554e5005270d29d19fc7d27b7fd22f0201b547", r.l, 4); t.beta_pallas = PallasFp::to_mont(r); }
    { VestaFp r; hex_to_limbs("397e65a7d7c1ad71aee24b27e308f0a61259527ec1d4752e619d1840af55f1b1", r.l, 4); t.beta_vesta = VestaFp::to_mont(r); }
    return t;
  }();
  return t;
}
template <> struct EndoFor<BlsFp> { static const EndoG1* get(int c) { return c == C_BLS_G1 ? &endo_table().e[c] : nullptr; } static BlsFp beta(int) { return endo_table().beta_bls; } };
template <> struct EndoFor<BnFp> { static const EndoG1* get(int c) { return c == C_BN_G1 ? &endo_table().e[c] : nullptr; } static BnFp beta(int) { return endo_table().beta_bn; } };
template <> struct EndoFor<PallasFp> { static const EndoG1* get(int c) { return c == C_PALLAS ? &endo_table().e[c] : nullptr; } static PallasFp beta(int) { return endo_table().beta_pallas; } };
template <> struct EndoFor<VestaFp> { static const EndoG1* get(int c) { return c == C_VESTA ? &endo_table().e[c] : nullptr; } static VestaFp beta(int) { return endo_table().beta_vesta; } };

// G2: lattices and Babai vectors of named/constants/bls12_381_endomorphisms.nim:39-60 and bn254_snarks_endomorphisms.nim:39-64; the psi
// coefficients are DERIVED here, c2 = t^((p-1)/3), c3 = t^((p-1)/2) with t = 1/xi for the M-twist of BLS12-381 (xi = 1 + i) and t = xi
// for the D-twist of BN254-Snarks (xi = 9 + i) -- tests/test_oracle_c.py checks psi(G) = [p mod r]G against the big-integer oracle
static EndoG2 make_endo4(const char* const lat[4][4], const bool latn[4][4], const char* const bab[4], const bool babn[4]) {
  EndoG2 e;
  for (int i = 0; i < 4; i++) {
    hex_to_limbs(bab[i], e.babai[i], 4);
    e.babai_neg[i] = babn[i];
    for (int j = 0; j < 4; j++) {
      hex_to_limbs(lat[i][j], &e.lat[i][j], 1);
      e.lat_neg[i][j] = latn[i][j];
    }
  }
  return e;
}
template <class B>
static PsiCoef<Fp2<B>> make_psi(u64 xi0, u64 xi1, bool invert) {
  Fp2<B> t{B::from_u64(xi0), B::from_u64(xi1)};
  if (invert) t = Fp2<B>::inv(t);
  // (p - 1) / 3 and (p - 1) / 2 from the modulus limbs
  const u64* P = B::ctx.p;
  u64 pm1[B::N], e3[B::N], e2[B::N];
  for (int i = 0; i < B::N; i++) pm1[i] = P[i];
  pm1[0] -= 1;                                      // p is odd
  u64 rem = 0;
  for (int i = B::N - 1; i >= 0; i--) {            // / 3
    u128 cur = ((u128)rem << 64) | pm1[i];
    e3[i] = (u64)(cur / 3);
    rem = (u64)(cur % 3);
  }
  for (int i = 0; i < B::N; i++) e2[i] = (pm1[i] >> 1) | (i + 1 < B::N ? pm1[i + 1] << 63 : 0);
  return {fp2_pow<Fp2<B>>(t, e3, B::N), fp2_pow<Fp2<B>>(t, e2, B::N)};
}
struct EndoG2Table {
  EndoG2 bls, bn;
  PsiCoef<Fp2<BlsFp>> psi_bls;
  PsiCoef<Fp2<BnFp>> psi_bn;
};
static const EndoG2Table& endo_g2_table() {
  static const EndoG2Table t = []() {
    EndoG2Table t;
    static const char* const X = "d201000000010000";
    static const char* const lat_bls[4][4] = {{X, "1", "0", "0"}, {"0", X, "1", "0"}, {"0", "0", X, "1"}, {"1", "0", "1", X}};
    static const bool latn_bls[4][4] = {{false, false, false, false}, {false, false, false, false}, {false, false, false, false}, {false, false, true, true}};
    static const char* const bab_bls[4] = {"1381204ca56cd56b533cfcc0d3e76ec2892078a5e8573b29c", "17c6becf1e01faadd63f6e522f6cfee2e", "1cfbe4f7bd0027db2", "2"};
    static const bool babn_bls[4] = {false, true, false, false};
    t.bls = make_endo4(lat_bls, latn_bls, bab_bls, babn_bls);
    static const char* const A = "89d3256894d213e2"; static const char* const Bq = "44e992b44a6909f2"; static const char* const Cq = "44e992b44a6909f1"; static const char* const D = "89d3256894d213e3";
    static const char* const lat_bn[4][4] = {{A, Bq, Cq, Cq}, {Cq, Cq, Cq, D}, {Bq, Cq, Cq, A}, {D, Cq, Bq, Cq}};
    static const bool latn_bn[4][4] = {{false, false, true, false}, {true, false, true, true}, {false, false, false, true}, {false, true, true, true}};
    static const char* const bab_bn[4] = {"9e80318ab0d92b9308e5da66fc7184ae46f4bda995d51bb1", "9e80318ab0d92b9555b4ca7ba3e5577f2dff291532e42728",
                                          "9e80318ab0d92b9555b4ca7ba3e55782071c4c43fac4daff", "9e80318ab0d92b9555b4ca7ba3e5577dc170977dcef3cd3f"};
    static const bool babn_bn[4] = {false, true, false, false};
    t.bn = make_endo4(lat_bn, latn_bn, bab_bn, babn_bn);
    t.psi_bls = make_psi<BlsFp>(1, 1, true);
    t.psi_bn = make_psi<BnFp>(9, 1, false);
    return t;
  }();
  return t;
}
template <> struct EndoG2For<Fp2<BlsFp>> { static const EndoG2* get(int c) { return c == C_BLS_G2 ? &endo_g2_table().bls : nullptr; } static const PsiCoef<Fp2<BlsFp>>& psi_coef(int) { return endo_g2_table().psi_bls; } };
template <> struct EndoG2For<Fp2<BnFp>> { static const EndoG2* get(int c) { return c == C_BN_G2 ? &endo_g2_table().bn : nullptr; } static const PsiCoef<Fp2<BnFp>>& psi_coef(int) { return endo_g2_table().psi_bn; } };

// psi on Jacobian coordinates: x = X / Z^2, y = Y / Z^3 and psi(x, y) = (conj(x) c2, conj(y) c3) give (conj(X) c2, conj(Y) c3, conj(Z))
template <class F2>
static Jac<F2> psi_jac(const PsiCoef<F2>& C, const Jac<F2>& p) {
  if (p.is_inf()) return p;
  auto conj = [](const F2& a) { return F2{a.c0, decltype(a.c0)::neg(a.c1)}; };
  return {F2::mul(conj(p.x), C.c2), F2::mul(conj(p.y), C.c3), conj(p.z)};
}

extern "C" {

// returns the window size c used (>0), or <0 on bad curve id. out = affine point in the C-API layout
// (Montgomery limbs; (0,0) = neutral). Scalars are canonical BigInt[bits] (4 x u64 LE).
int oracle_msm(int curve, const void* scalars, const void* points, size_t n, int nthreads, int c_override, void* out) {
  ensure_init();
  int bits = curve_bits(curve);
  switch (curve) {
    case C_BLS_G1: return do_msm<BlsFp>(curve, scalars, points, n, bits, nthreads, c_override, out);
    case C_BLS_G2: return do_msm<Fp2<BlsFp>>(curve, scalars, points, n, bits, nthreads, c_override, out);
    case C_BN_G1: return do_msm<BnFp>(curve, scalars, points, n, bits, nthreads, c_override, out);
    case C_BN_G2: return do_msm<Fp2<BnFp>>(curve, scalars, points, n, bits, nthreads, c_override, out);
    case C_PALLAS: return do_msm<PallasFp>(curve, scalars, points, n, bits, nthreads, c_override, out);
    case C_VESTA: return do_msm<VestaFp>(curve, scalars, points, n, bits, nthreads, c_override, out);
  }
  return -1;
}

// Fr Montgomery <-> canonical (batchFromField, finite_fields.nim:915-920)
int oracle_fr_from_mont(int curve, const void* in, void* out, size_t n) {
  ensure_init();
  switch (curve) {
    case C_BLS_G1: case C_BLS_G2: fr_from_mont_arr<BlsFr>(in, out, n); return 0;
    case C_BN_G1: case C_BN_G2: fr_from_mont_arr<BnFr>(in, out, n); return 0;
    case C_PALLAS: fr_from_mont_arr<VestaFp>(in, out, n); return 0;
    case C_VESTA: fr_from_mont_arr<PallasFp>(in, out, n); return 0;
  }
  return -1;
}
int oracle_fr_to_mont(int curve, const void* in, void* out, size_t n) {
  ensure_init();
  switch (curve) {
    case C_BLS_G1: case C_BLS_G2: fr_to_mont_arr<BlsFr>(in, out, n); return 0;
    case C_BN_G1: case C_BN_G2: fr_to_mont_arr<BnFr>(in, out, n); return 0;
    case C_PALLAS: fr_to_mont_arr<VestaFp>(in, out, n); return 0;
    case C_VESTA: fr_to_mont_arr<PallasFp>(in, out, n); return 0;
  }
  return -1;
}

// out[i] = [s_i]G for i in [first, first+n): deterministic subgroup points in the C-API affine layout
int oracle_gen_points(int curve, u64 seed, size_t first, size_t n, void* out, int nthreads) {
  ensure_init();
  switch (curve) {
    case C_BLS_G1: {
      Aff<BlsFp> G = {fp_hex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
                      fp_hex("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")};
      gen_points<BlsFp>(G, seed, first, n, out, nthreads); return 0; }
    case C_BLS_G2: {
      Aff<Fp2<BlsFp>> G = {
        {fp_hex("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
         fp_hex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")},
        {fp_hex("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
         fp_hex("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")}};
      gen_points<Fp2<BlsFp>>(G, seed, first, n, out, nthreads); return 0; }
    case C_BN_G1: gen_points<BnFp>(aff_from_u64<BnFp>(1, 2), seed, first, n, out, nthreads); return 0;
    case C_BN_G2: {
      Aff<Fp2<BnFp>> G = {
        {bn_hex("1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED"),
         bn_hex("198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2")},
        {bn_hex("12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA"),
         bn_hex("090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B")}};
      gen_points<Fp2<BnFp>>(G, seed, first, n, out, nthreads); return 0; }
    case C_PALLAS: {
      Aff<PallasFp> G = {PallasFp::neg(PallasFp::one()), PallasFp::from_u64(2)};
      gen_points<PallasFp>(G, seed, first, n, out, nthreads); return 0; }
    case C_VESTA: {
      Aff<VestaFp> G = {VestaFp::neg(VestaFp::one()), VestaFp::from_u64(2)};
      gen_points<VestaFp>(G, seed, first, n, out, nthreads); return 0; }
  }
  return -1;
}

// out = [k]P (affine, C-API layout); k = 4 x u64 LE
int oracle_scalar_mul(int curve, const void* k, const void* p, void* out) {
  ensure_init();
  const Scalar& s = *(const Scalar*)k;
  switch (curve) {
    case C_BLS_G1: *(Aff<BlsFp>*)out = scalar_mul<BlsFp>(s, *(const Aff<BlsFp>*)p).to_aff(); return 0;
    case C_BLS_G2: *(Aff<Fp2<BlsFp>>*)out = scalar_mul<Fp2<BlsFp>>(s, *(const Aff<Fp2<BlsFp>>*)p).to_aff(); return 0;
    case C_BN_G1: *(Aff<BnFp>*)out = scalar_mul<BnFp>(s, *(const Aff<BnFp>*)p).to_aff(); return 0;
    case C_BN_G2: *(Aff<Fp2<BnFp>>*)out = scalar_mul<Fp2<BnFp>>(s, *(const Aff<Fp2<BnFp>>*)p).to_aff(); return 0;
    case C_PALLAS: *(Aff<PallasFp>*)out = scalar_mul<PallasFp>(s, *(const Aff<PallasFp>*)p).to_aff(); return 0;
    case C_VESTA: *(Aff<VestaFp>*)out = scalar_mul<VestaFp>(s, *(const Aff<VestaFp>*)p).to_aff(); return 0;
  }
  return -1;
}

int oracle_best_bucket_bit_size(size_t n, int bits) { return best_bucket_bit_size(n, bits, true, true); }

// out[i], i in [first, first + n): random points of the prime-order subgroup whose discrete logarithms nobody knows
// (gen_points_unknown_log), every curve of the path.  Cofactor clearing -- the reference has both forms (clearCofactorReference: the
// multiplication by the cofactor; clearCofactorFast, zoo_subgroups.nim:34-40 -> bls12_381_subgroups.nim / bn254_snarks_subgroups.nim:
// the endomorphism-accelerated maps); this port takes the cheap one where the cofactor is long:
//   BLS12-381 G1   [1 - x]P, x = -0xd201000000010000 (64 bits, weight 6) instead of the 126-bit cofactor (Wahby-Boneh; the reference's fast form)
//   BLS12-381 G2   [x^2 - x - 1]P + [x - 1]psi(P) + psi^2([2]P) (Budroni-Pintore; the reference's fast form) instead of 512 bits
//   BN254-Snarks G2  the multiplication by #E'(Fp2) / r = 2p - r (256 bits)
//   BN254-Snarks G1, Pallas, Vesta: prime order, nothing to clear.
// tests/test_oracle_c.py holds samples of every curve to the big-integer oracle: on the curve, distinct, [r]P neutral.
int oracle_gen_points_unknown_log(int curve, u64 seed, size_t first, size_t n, void* out, int nthreads) {
  ensure_init();
  static const u64 absx[1] = {0xd201000000010000ull};   // |x| of BLS12-381; x < 0
  switch (curve) {
    case C_BLS_G1: {
      static const u64 h[1] = {0xd201000000010001ull};  // 1 - x
      gen_points_unknown_log<BlsFp>(BlsFp::from_u64(4), [](const Aff<BlsFp>& P) { return mul_words<BlsFp>(P, h, 1); }, seed, first, n, out, nthreads);
      return 0; }
    case C_BN_G1: gen_points_unknown_log<BnFp>(BnFp::from_u64(3), [](const Aff<BnFp>& P) { return Jac<BnFp>::from_aff(P); }, seed, first, n, out, nthreads); return 0;
    case C_PALLAS: gen_points_unknown_log<PallasFp>(PallasFp::from_u64(5), [](const Aff<PallasFp>& P) { return Jac<PallasFp>::from_aff(P); }, seed, first, n, out, nthreads); return 0;
    case C_VESTA: gen_points_unknown_log<VestaFp>(VestaFp::from_u64(5), [](const Aff<VestaFp>& P) { return Jac<VestaFp>::from_aff(P); }, seed, first, n, out, nthreads); return 0;
    case C_BLS_G2: {   // y^2 = x^3 + 4 (1 + i), M-twist (config_fields_and_curves.nim:269-287)
      using F2 = Fp2<BlsFp>;
      const F2 b{BlsFp::from_u64(4), BlsFp::from_u64(4)};
      const PsiCoef<F2>& K = endo_g2_table().psi_bls;
      auto clear = [&K](const Aff<F2>& P) {
        const Jac<F2> Pj = Jac<F2>::from_aff(P);
        const Jac<F2> t1 = jac_neg(mul_words<F2>(P, absx, 1));        // [x]P
        const Jac<F2> t2 = jac_neg(mul_words<F2>(t1, absx, 1));       // [x^2]P
        const Jac<F2> t1mP = Jac<F2>::add(t1, jac_neg(Pj));           // [x - 1]P
        Jac<F2> r = Jac<F2>::add(t2, jac_neg(t1));                    // [x^2 - x]P
        r = Jac<F2>::add(r, jac_neg(Pj));                             // [x^2 - x - 1]P
        r = Jac<F2>::add(r, psi_jac<F2>(K, t1mP));                    // + psi([x - 1]P)
        return Jac<F2>::add(r, psi_jac<F2>(K, psi_jac<F2>(K, Jac<F2>::dbl(Pj))));   // + psi^2([2]P)
      };
      gen_points_unknown_log<F2>(b, clear, seed, first, n, out, nthreads); return 0; }
    case C_BN_G2: {    // y^2 = x^3 + 3 / (9 + i), D-twist (config_fields_and_curves.nim:116-133); cofactor 2p - r
      using F2 = Fp2<BnFp>;
      static const u64 h[4] = {0x345f2299c0f9fa8dull, 0x06ceecda572a2489ull, 0xb85045b68181585eull, 0x30644e72e131a029ull};
      const F2 xi{BnFp::from_u64(9), BnFp::from_u64(1)};
      const F2 b = F2::mul(F2{BnFp::from_u64(3), BnFp::zero()}, F2::inv(xi));
      gen_points_unknown_log<F2>(b, [](const Aff<F2>& P) { return mul_words<F2>(P, h, 4); }, seed, first, n, out, nthreads); return 0; }
  }
  return -1;
}

// out = psi(p) for a G2 point (affine, C-API layout): the endomorphism of the M = 4 pre-split (tests: psi(G) = [p mod r]G)
int oracle_psi_g2(int curve, const void* p, void* out) {
  ensure_init();
  switch (curve) {
    case C_BLS_G2: *(Aff<Fp2<BlsFp>>*)out = psi<Fp2<BlsFp>>(endo_g2_table().psi_bls, *(const Aff<Fp2<BlsFp>>*)p); return 0;
    case C_BN_G2: *(Aff<Fp2<BnFp>>*)out = psi<Fp2<BnFp>>(endo_g2_table().psi_bn, *(const Aff<Fp2<BnFp>>*)p); return 0;
  }
  return -1;
}
// mini[4] (4 x u64 each, magnitudes) and neg[4] of the M = 4 decomposition of k (tests: sum_j (-1)^neg_j mini_j lambda^j = k mod r)
int oracle_decompose_g2(int curve, const void* k, void* mini, int* neg) {
  ensure_init();
  const EndoG2* E = curve == C_BLS_G2 ? &endo_g2_table().bls : curve == C_BN_G2 ? &endo_g2_table().bn : nullptr;
  if (!E) return -1;
  bool ng[4];
  decompose_endo4(*E, *(const Scalar*)k, (Scalar*)mini, ng);
  for (int j = 0; j < 4; j++) neg[j] = ng[j] ? 1 : 0;
  return 0;
}

}  // extern "C"
