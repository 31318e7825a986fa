"""
oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Big-integer CPU restatement of the reference's multi-scalar-multiplication path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity status: PINNED.  This oracle is checked (tests/test_oracle_golden.py) against
the reference's own golden vectors, re-encoded under tests/golden/ by
tests/golden/make_golden.py:
  * tests/math_elliptic_curves/vectors/tv_{BLS12_381,BN254_Snarks}_scalar_mul_{G1,G2}_*bit.json
    and tv_{Pallas,Vesta}_scalar_mul_G1_255bit.json  (40 [k]P=Q vectors each)
  * tests/protocol_ethereum_evm_precompiles/eip-2537/multiexp_{G1,G2}_bls.json

What it restates (reference file:line, relative to the reference root):
  * curve/field parameters      constantine/named/config_fields_and_curves.nim:116-133,214-229,269-287
  * affine group law, neutral   constantine/math/elliptic/ec_shortweierstrass_affine.nim:47-62
    (the neutral element of EC_ShortW_Aff is encoded as (0,0))
  * Fp2 = Fp[i]/(i^2+1)          constantine/math/extension_fields/towers.nim:758-878
  * Booth signed windows         constantine/math/arithmetic/bigints.nim:360-380,806-859
  * bucket method (BDLO12)       constantine/math/elliptic/ec_multi_scalar_mul.nim:40-95 (reference),
                                 :204-296 (signed windows, top/extra window rule)
  * window-size heuristic        constantine/math/elliptic/ec_multi_scalar_mul_scheduler.nim:172-223
  * Montgomery constants         constantine/named/deriv/precompute.nim:248-373
  * in-memory layout             constantine/platforms/abstractions.nim:131-143 (LE limbs of 64-bit words)

Everything is plain Python integers and affine formulas: there is no Montgomery
arithmetic inside the oracle, only at the (de)serialisation boundary.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# Fields
# --------------------------------------------------------------------------------------


class FpField:
    """Prime field; elements are python ints in [0,p)."""

    degree = 1

    def __init__(self, p: int):
        self.p = p
        self.nbits = p.bit_length()
        self.limbs64 = (self.nbits + 63) // 64
        self.nbytes = self.limbs64 * 8
        self.R = 1 << (64 * self.limbs64)  # Montgomery radix (precompute.nim:248-373)
        self.Rinv = pow(self.R, -1, p)
        self.zero = 0
        self.one = 1

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def neg(self, a): return (-a) % self.p
    def mul(self, a, b): return (a * b) % self.p
    def sqr(self, a): return (a * a) % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def is_zero(self, a): return a == 0
    def from_int(self, v): return v % self.p

    # --- (de)serialisation: Montgomery residue, little-endian 64-bit limbs -------------
    def to_mont_bytes(self, a) -> bytes:
        return ((a * self.R) % self.p).to_bytes(self.nbytes, "little")

    def from_mont_bytes(self, b: bytes):
        return (int.from_bytes(b[: self.nbytes], "little") * self.Rinv) % self.p


class Fp2Field:
    """Fp[i]/(i^2+1); elements are tuples (c0, c1). towers.nim:758-878."""

    degree = 2

    def __init__(self, base: FpField):
        self.base = base
        self.p = base.p
        self.nbytes = 2 * base.nbytes
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b): p = self.p; return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)
    def sub(self, a, b): p = self.p; return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)
    def neg(self, a): p = self.p; return ((-a[0]) % p, (-a[1]) % p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a): return self.mul(a, a)

    def inv(self, a):
        p = self.p
        n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return ((a[0] * n) % p, (-a[1] * n) % p)

    def is_zero(self, a): return a[0] == 0 and a[1] == 0
    def from_int(self, v): return (v % self.p, 0)

    def to_mont_bytes(self, a) -> bytes:
        return self.base.to_mont_bytes(a[0]) + self.base.to_mont_bytes(a[1])

    def from_mont_bytes(self, b: bytes):
        n = self.base.nbytes
        return (self.base.from_mont_bytes(b[:n]), self.base.from_mont_bytes(b[n: 2 * n]))


# --------------------------------------------------------------------------------------
# Curves (all short Weierstrass, a = 0)
# --------------------------------------------------------------------------------------


class Curve:
    """y^2 = x^3 + b over `field`; affine points are (x, y) tuples, None = neutral."""

    def __init__(self, name, field, b, order, scalar_bits, gen, fr_modulus=None):
        self.name = name
        self.F = field
        self.b = b
        self.order = order
        self.scalar_bits = scalar_bits  # bits of BigInt[bits] taken by the C API (254 / 255)
        self.gen = gen
        self.Fr = FpField(fr_modulus if fr_modulus is not None else order)
        self.coord_bytes = field.nbytes
        self.aff_bytes = 2 * field.nbytes
        self.scalar_bytes = 32

    # -- group law -----------------------------------------------------------------------
    def is_on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sqr(y) == F.add(F.mul(F.sqr(x), x), self.b)

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def double(self, P):
        if P is None:
            return None
        F = self.F
        x, y = P
        if F.is_zero(y):
            return None
        xx = F.sqr(x)
        lam = F.mul(F.add(F.add(xx, xx), xx), F.inv(F.add(y, y)))
        x3 = F.sub(F.sub(F.sqr(lam), x), x)
        y3 = F.sub(F.mul(lam, F.sub(x, x3)), y)
        return (x3, y3)

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        F = self.F
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if y1 == y2:
                return self.double(P)
            return None
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    def scalar_mul(self, k: int, P):
        """Plain left-to-right double-and-add; k is any non-negative integer (NOT reduced)."""
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.double(R)
            if bit == "1":
                R = self.add(R, P)
        return R

    # -- MSM ------------------------------------------------------------------------------
    def msm_naive(self, scalars, points):
        """sum_i [k_i]P_i  -- the 'naive' side of t_ec_template.nim:1440-1483."""
        R = None
        for k, P in zip(scalars, points):
            R = self.add(R, self.scalar_mul(k, P))
        return R

    def msm_pippenger(self, scalars, points, c=None):
        """Signed-window bucket method, ec_multi_scalar_mul.nim:204-296 + Appendix-B of SURVEY.md.

        W = floor(bits/c)+1 windows (the extra top window absorbs the Booth carry,
        ec_multi_scalar_mul.nim:278-289), 2^(c-1) buckets per window."""
        n = len(scalars)
        bits = self.scalar_bits
        if c is None:
            c = best_bucket_bit_size(n, bits, True, True)
        W = bits // c + 1
        B = 1 << (c - 1)
        sums = []
        for w in range(W):
            buckets = [None] * (B + 1)
            for k, P in zip(scalars, points):
                val, neg = booth_digit(k, w, c)
                if val:
                    buckets[val] = self.add(buckets[val], self.neg(P) if neg else P)
            acc = None
            s = None
            for kk in range(B, 0, -1):  # bucketReduce, ec_multi_scalar_mul.nim:186-197
                acc = self.add(acc, buckets[kk])
                s = self.add(s, acc)
            sums.append(s)
        r = sums[W - 1]
        for w in range(W - 2, -1, -1):  # final Horner, ec_multi_scalar_mul.nim:250-254
            for _ in range(c):
                r = self.double(r)
            r = self.add(r, sums[w])
        return r

    # -- serialisation (C-API layouts, include/constantine/curves/*.h) ---------------------
    def aff_to_bytes(self, P) -> bytes:
        F = self.F
        if P is None:  # neutral = (0,0), ec_shortweierstrass_affine.nim:52-62
            return bytes(self.aff_bytes)
        return F.to_mont_bytes(P[0]) + F.to_mont_bytes(P[1])

    def aff_from_bytes(self, b: bytes):
        F = self.F
        n = F.nbytes
        x = F.from_mont_bytes(b[:n])
        y = F.from_mont_bytes(b[n: 2 * n])
        if F.is_zero(x) and F.is_zero(y):
            return None
        return (x, y)

    def jac_from_bytes(self, b: bytes):
        """EC_ShortW_Jac (X,Y,Z): x=X/Z^2, y=Y/Z^3; neutral has Z=0 (jacobian.nim:28-64)."""
        F = self.F
        n = F.nbytes
        X, Y, Z = (F.from_mont_bytes(b[i * n:(i + 1) * n]) for i in range(3))
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def prj_from_bytes(self, b: bytes):
        """EC_ShortW_Prj (X,Y,Z): x=X/Z, y=Y/Z; neutral (0,1,0) (projective.nim:28-62)."""
        F = self.F
        n = F.nbytes
        X, Y, Z = (F.from_mont_bytes(b[i * n:(i + 1) * n]) for i in range(3))
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        return (F.mul(X, zi), F.mul(Y, zi))

    def points_to_array(self, points) -> np.ndarray:
        buf = b"".join(self.aff_to_bytes(P) for P in points)
        return np.frombuffer(buf, dtype=np.uint8).reshape(len(points), self.aff_bytes).copy()

    def scalars_to_array(self, scalars) -> np.ndarray:
        """BigInt[bits]: canonical (non-Montgomery) integer, 4 LE 64-bit limbs."""
        buf = b"".join(int(k).to_bytes(32, "little") for k in scalars)
        return np.frombuffer(buf, dtype=np.uint8).reshape(len(scalars), 32).copy()

    def fr_scalars_to_array(self, scalars) -> np.ndarray:
        """Fr elements in Montgomery form (what *_fr_coefs_* entry points take)."""
        buf = b"".join(self.Fr.to_mont_bytes(int(k) % self.Fr.p) for k in scalars)
        return np.frombuffer(buf, dtype=np.uint8).reshape(len(scalars), 32).copy()


# --------------------------------------------------------------------------------------
# Window recoding & window-size heuristic
# --------------------------------------------------------------------------------------


def booth_digit(k: int, w: int, c: int):
    """Signed Booth digit of window w (bits [w*c-1, w*c+c-1]) -> (val, neg).

    bigints.nim:806-859: d = c+1 bits starting one bit below the window (bit -1 := 0),
    neg = top bit of d, e = (d+1)>>1, val = neg ? 2^c - e : e.   val in [0, 2^(c-1)]."""
    i = w * c
    if i == 0:
        d = (k << 1) & ((1 << (c + 1)) - 1)
    else:
        d = (k >> (i - 1)) & ((1 << (c + 1)) - 1)
    neg = d >> c
    e = (d + 1) >> 1
    val = ((1 << c) - e) if neg else e
    val &= (1 << c) - 1
    return val, bool(neg)


def best_bucket_bit_size(n: int, bits: int, signed: bool = True, manual: bool = True) -> int:
    """ec_multi_scalar_mul_scheduler.nim:172-223, float32 arithmetic like the reference."""
    f = np.float32
    A, D = f(10), f(6)
    s = 1 if signed else 0
    b = f(bits)
    best, best_cost = 2, f(np.inf)
    for c in range(2, 21):
        b_over_c = b / f(c)
        acc = b_over_c * f(n + (1 << (c - s)) - 2) * A
        fin = (b_over_c - f(1)) * (f(c) * D + A)
        cost = acc + fin
        if cost < best_cost:
            best_cost, best = cost, c
    if manual:
        if best >= 14:
            best -= 1
        if best >= 15:
            best -= 1
        if best >= 16:
            best -= 1
    return best


# --------------------------------------------------------------------------------------
# Curve instances
# --------------------------------------------------------------------------------------

_BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
_BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
_BN_P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
_BN_R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
_PALLAS_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
_VESTA_P = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001

_FP_BLS = FpField(_BLS_P)
_FP_BN = FpField(_BN_P)
_FP2_BLS = Fp2Field(_FP_BLS)
_FP2_BN = Fp2Field(_FP_BN)

BLS12_381_G1 = Curve(
    "bls12_381_g1", _FP_BLS, 4, _BLS_R, 255,
    (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
     0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1))

BLS12_381_G2 = Curve(
    "bls12_381_g2", _FP2_BLS, (4, 4), _BLS_R, 255,
    ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
      0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
     (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
      0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)))

BN254_SNARKS_G1 = Curve("bn254_snarks_g1", _FP_BN, 3, _BN_R, 254, (1, 2))

# b' = 3/(9+i)  (D-twist)
_BN_G2_B = _FP2_BN.mul((3, 0), _FP2_BN.inv((9, 1)))
BN254_SNARKS_G2 = Curve(
    "bn254_snarks_g2", _FP2_BN, _BN_G2_B, _BN_R, 254,
    ((0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
      0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
     (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
      0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B)))

PALLAS = Curve("pallas", FpField(_PALLAS_P), 5, _VESTA_P, 255, (_PALLAS_P - 1, 2))
VESTA = Curve("vesta", FpField(_VESTA_P), 5, _PALLAS_P, 255, (_VESTA_P - 1, 2))

CURVES = {c.name: c for c in (BLS12_381_G1, BLS12_381_G2, BN254_SNARKS_G1, BN254_SNARKS_G2, PALLAS, VESTA)}


# --------------------------------------------------------------------------------------
# Deterministic synthetic inputs (shared definition with the C oracle and the HIP generator)
# --------------------------------------------------------------------------------------

_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    """One splitmix64 output for state x (helpers/prng_unsafe.nim:50-93 seeds xoshiro with it)."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def synth_scalar(seed: int, i: int, bits: int) -> int:
    """Uniform integer in [0, 2^bits), NOT reduced mod r (prng_unsafe.nim:166-183)."""
    v = 0
    for k in range(4):
        v |= splitmix64((seed + 4 * i + k) & _M64) << (64 * k)
    return v & ((1 << bits) - 1)


def synth_scalars(seed: int, n: int, bits: int):
    return [synth_scalar(seed, i, bits) for i in range(n)]


def synth_point(curve: Curve, seed: int, i: int):
    """P_i = [s_i]G with s_i = synth_scalar(seed ^ 0xA5A5.., i, 128): uniform-looking subgroup points."""
    s = synth_scalar(seed ^ 0xA5A5A5A5A5A5A5A5, i, 128) | 1
    return curve.scalar_mul(s, curve.gen)
