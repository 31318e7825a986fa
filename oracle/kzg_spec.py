"""
Spec restatement of the EIP-4844 host logic in plain Python integers -- TEST INFRASTRUCTURE (like everything under oracle/):
what the C++ host side of the protocol symbols (constantine_amd/csrc/protocols.hip) and the device quotient kernel are checked
against.  Only tests/ import it.  Follows
    constantine/ethereum_eip4844_kzg.nim:103-166 (fromDigest, fiatShamirChallenge, bytes_to_bls_field), :297-444
    constantine/commitments/kzg.nim:186-223 (kzg_commit, kzg_prove), math/polynomials/polynomials.nim (getQuotientPoly)
    constantine/serialization/codecs_bls12_381.nim (ZCash compressed G1)
and is itself pinned by the reference's vectors (tests/test_kzg_golden.py).
"""
import numpy as np

FIELD_ELEMENTS_PER_BLOB = 4096
BYTES_PER_BLOB = FIELD_ELEMENTS_PER_BLOB * 32
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
MONT = 1 << 384      # Montgomery radix of the C-API representation of Fp[BLS12_381]
FR_MONT = 1 << 256


class SpecError(ValueError):
    """.status = the name of the cttEthKzg_* status the reference returns"""

    def __init__(self, status):
        super().__init__(status)
        self.status = status


def deserialize_g1_compressed(b48: bytes):
    """ZCash/IETF BLS12-381 G1 compressed encoding -> affine (x, y) ints, or None for the point at infinity."""
    if len(b48) != 48 or not (b48[0] & 0x80):
        raise SpecError("cttEthKzg_EccInvalidEncoding")
    if b48[0] & 0x40:
        if (b48[0] & 0x3F) or any(b48[1:]):
            raise SpecError("cttEthKzg_EccInvalidEncoding")
        return None
    x = int.from_bytes(b48, "big") & ((1 << 381) - 1)
    if x >= P:
        raise SpecError("cttEthKzg_EccCoordinateGreaterThanOrEqualModulus")
    y2 = (pow(x, 3, P) + 4) % P
    y = pow(y2, (P + 1) // 4, P)  # p = 3 (mod 4)
    if y * y % P != y2:
        raise SpecError("cttEthKzg_EccPointNotOnCurve")
    if bool(b48[0] & 0x20) != (y > (P - 1) // 2):
        y = P - y
    return (x, y)


def serialize_g1_compressed(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80 | (0x20 if y > (P - 1) // 2 else 0)
    return bytes(out)


def aff_mont_bytes(pt) -> bytes:
    """affine point -> the C-API bytes (Montgomery residues, little-endian; the neutral is (0,0))"""
    if pt is None:
        return bytes(96)
    return ((pt[0] * MONT) % P).to_bytes(48, "little") + ((pt[1] * MONT) % P).to_bytes(48, "little")


def bit_reversal_permutation(seq):
    n = len(seq)
    bits = n.bit_length() - 1
    assert 1 << bits == n
    return [seq[int(format(i, f"0{bits}b")[::-1], 2)] for i in range(n)]


def blob_to_bigint_polynomial(blob: bytes) -> np.ndarray:
    """4096 big-endian scalars -> (4096, 32) little-endian BigInt[255] array; every scalar must be < r."""
    if len(blob) != BYTES_PER_BLOB:
        raise SpecError("cttEthKzg_InputsLengthsMismatch")
    be = np.frombuffer(blob, dtype=np.uint8).reshape(FIELD_ELEMENTS_PER_BLOB, 32)
    for i in range(FIELD_ELEMENTS_PER_BLOB):
        if int.from_bytes(bytes(be[i]), "big") >= R:
            raise SpecError("cttEthKzg_ScalarLargerThanCurveOrder")
    return np.ascontiguousarray(be[:, ::-1])


def bytes_to_bls_field(b32: bytes) -> int:
    if len(b32) != 32:
        raise SpecError("cttEthKzg_InputsLengthsMismatch")
    v = int.from_bytes(b32, "big")
    if v >= R:
        raise SpecError("cttEthKzg_ScalarLargerThanCurveOrder")
    return v


_domain_brp_cache = None


def domain_brp():
    """The 4096 roots of unity in bit-reversed order (ctx.domain_brp, ethereum_kzg_srs.nim)."""
    global _domain_brp_cache
    if _domain_brp_cache is None:
        w = pow(7, (R - 1) // FIELD_ELEMENTS_PER_BLOB, R)
        roots, x = [], 1
        for _ in range(FIELD_ELEMENTS_PER_BLOB):
            roots.append(x)
            x = x * w % R
        _domain_brp_cache = bit_reversal_permutation(roots)
    return _domain_brp_cache


def batch_inverse(vals):
    """Montgomery's trick over Fr; every value must be non-zero."""
    n = len(vals)
    pre, run = [0] * n, 1
    for i, v in enumerate(vals):
        pre[i] = run
        run = run * v % R
    inv = pow(run, -1, R)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = inv * pre[i] % R
        inv = inv * vals[i] % R
    return out


def quotient_polynomial(poly, z):
    """getQuotientPoly: -> (q, y) with y = p(z) and q = (p - y) / (X - z), both in evaluation form over the bit-reversed
    domain; z may be one of the roots of unity."""
    dom = domain_brp()
    n = FIELD_ELEMENTS_PER_BLOB
    try:
        m = dom.index(z)
    except ValueError:
        m = -1
    if m < 0:
        inv = batch_inverse([(z - w) % R for w in dom])                         # 1 / (z - w_i)
        s = sum(p * w % R * iv for p, w, iv in zip(poly, dom, inv)) % R
        y = (pow(z, n, R) - 1) * pow(n, -1, R) % R * s % R                        # barycentric evaluation
        q = [(y - p) * iv % R for p, iv in zip(poly, inv)]                        # (p_i - y) / (w_i - z)
        return q, y
    y = poly[m]
    others = [i for i in range(n) if i != m]
    inv = batch_inverse([(dom[i] - z) % R for i in others])                      # 1 / (w_i - z)
    q = [0] * n
    zinv = pow(z, -1, R)
    acc = 0
    for i, iv in zip(others, inv):
        q[i] = (poly[i] - y) * iv % R
        acc += q[i] * dom[i] % R * zinv                                           # q_m = - sum q_i * w_i / z
    q[m] = (-acc) % R
    return q, y


def compute_challenge(blob: bytes, commitment_bytes: bytes) -> int:
    """fiatShamirChallenge: sha256(domain | 16-byte big-endian degree | blob | commitment) reduced mod r."""
    import hashlib
    data = (b"FSBLOBVERIFY_V1_" + (0).to_bytes(8, "big") + FIELD_ELEMENTS_PER_BLOB.to_bytes(8, "big") + blob + commitment_bytes)
    return int.from_bytes(hashlib.sha256(data).digest(), "big") % R
