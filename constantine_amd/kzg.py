"""
EIP-4844 blob -> KZG commitment on top of the MSM engine (the MSM's immediate caller).

Mirrors the reference's
    blob_to_kzg_commitment(ctx, dst, blob) -> cttEthKzgStatus      constantine/ethereum_eip4844_kzg.nim:297-330
      -> blob_to_bigint_polynomial                                   (4096 big-endian 32-byte scalars, each < r)
      -> kzg_commit(ctx.srs_lagrange_brp_g1, r, poly)               constantine/commitments/kzg.nim:186
      -> serialize_g1_compressed                                      constantine/serialization/codecs_bls12_381.nim
The commitment is the 4096-point BLS12-381 G1 MSM of the blob's field elements with the Lagrange-form SRS in
bit-reversal order; the SRS is the textbook cached-base case, so it is uploaded and converted once
(constantine_amd.CachedBases) and every commitment moves only 128 KiB of scalars.

Commitments and opening proofs are implemented (the MSM callers: blob_to_kzg_commitment, compute_kzg_proof,
compute_blob_kzg_proof); verification needs pairings and is out of scope, like the cell/PeerDAS functions.
"""
from enum import Enum

import numpy as np

from .msm import CachedBases

FIELD_ELEMENTS_PER_BLOB = 4096
BYTES_PER_FIELD_ELEMENT = 32
BYTES_PER_BLOB = FIELD_ELEMENTS_PER_BLOB * BYTES_PER_FIELD_ELEMENT
BYTES_PER_COMMITMENT = 48

_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
_MONT = 1 << 384  # Montgomery radix of the C-API representation of Fp[BLS12_381]


class cttEthKzgStatus(Enum):  # ethereum_eip4844_kzg.nim:87-97 (the members this module can return)
    cttEthKzg_Success = 0
    cttEthKzg_InputsLengthsMismatch = 2
    cttEthKzg_ScalarLargerThanCurveOrder = 4
    cttEthKzg_EccInvalidEncoding = 5
    cttEthKzg_EccCoordinateGreaterThanOrEqualModulus = 6
    cttEthKzg_EccPointNotOnCurve = 7
    cttEthKzg_EccPointNotInSubGroup = 8


class KzgError(ValueError):
    def __init__(self, status):
        super().__init__(status.name)
        self.status = status


def deserialize_g1_compressed(b48: bytes):
    """ZCash/IETF BLS12-381 G1 compressed encoding -> affine (x, y) ints, or None for the point at infinity."""
    if len(b48) != 48 or not (b48[0] & 0x80):
        raise KzgError(cttEthKzgStatus.cttEthKzg_EccInvalidEncoding)
    if b48[0] & 0x40:
        if (b48[0] & 0x3F) or any(b48[1:]):
            raise KzgError(cttEthKzgStatus.cttEthKzg_EccInvalidEncoding)
        return None
    x = int.from_bytes(b48, "big") & ((1 << 381) - 1)
    if x >= _P:
        raise KzgError(cttEthKzgStatus.cttEthKzg_EccCoordinateGreaterThanOrEqualModulus)
    y2 = (pow(x, 3, _P) + 4) % _P
    y = pow(y2, (_P + 1) // 4, _P)  # p = 3 (mod 4)
    if y * y % _P != y2:
        raise KzgError(cttEthKzgStatus.cttEthKzg_EccPointNotOnCurve)
    if bool(b48[0] & 0x20) != (y > (_P - 1) // 2):
        y = _P - y
    return (x, y)


def serialize_g1_compressed(P) -> bytes:
    if P is None:
        return bytes([0xC0]) + bytes(47)
    x, y = P
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80 | (0x20 if y > (_P - 1) // 2 else 0)
    return bytes(out)


def _bit_reversal_permutation(seq):
    n = len(seq)
    bits = n.bit_length() - 1
    assert 1 << bits == n
    return [seq[int(format(i, f"0{bits}b")[::-1], 2)] for i in range(n)]


def _aff_mont_bytes(P) -> bytes:
    if P is None:
        return bytes(96)
    return ((P[0] * _MONT) % _P).to_bytes(48, "little") + ((P[1] * _MONT) % _P).to_bytes(48, "little")


def _aff_from_mont_bytes(b: bytes):
    inv = pow(_MONT, -1, _P)
    x = int.from_bytes(b[:48], "little") * inv % _P
    y = int.from_bytes(b[48:96], "little") * inv % _P
    return None if (x == 0 and y == 0) else (x, y)


class EthereumKZGContext:
    """Holds srs_lagrange_brp_g1 resident on the GPU (constantine/commitments_setups/ethereum_kzg_srs.nim)."""

    def __init__(self, srs_lagrange_g1_compressed: bytes, ctx=None):
        """`srs_lagrange_g1_compressed`: 4096 x 48 bytes, the G1 Lagrange points in ceremony (file) order."""
        if len(srs_lagrange_g1_compressed) != FIELD_ELEMENTS_PER_BLOB * 48:
            raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
        pts = [deserialize_g1_compressed(srs_lagrange_g1_compressed[48 * i:48 * i + 48])
               for i in range(FIELD_ELEMENTS_PER_BLOB)]
        pts = _bit_reversal_permutation(pts)
        arr = np.frombuffer(b"".join(_aff_mont_bytes(P) for P in pts), dtype=np.uint8)
        self.srs_lagrange_brp_g1 = arr.reshape(FIELD_ELEMENTS_PER_BLOB, 96).copy()
        self.hip_ctx = ctx   # ctt_hip_msm_ctx* (None = the process default context): the GPU everything of this context runs on
        self._bases = CachedBases("bls12_381_g1", self.srs_lagrange_brp_g1, ctx=ctx)

    @classmethod
    def from_ckzg_text(cls, path, ctx=None):
        """The c-kzg text format of the Ethereum ceremony ("4096\\n65\\n" + one hex point per line), as shipped by the
        reference in constantine/commitments_setups/trusted_setup_ethereum_kzg4844_reference.dat."""
        tok = open(path).read().split()
        n1 = int(tok[0])
        if n1 != FIELD_ELEMENTS_PER_BLOB:
            raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
        return cls(b"".join(bytes.fromhex(h) for h in tok[2:2 + n1]), ctx=ctx)

    def delete(self):
        self._bases.close()


def blob_to_bigint_polynomial(blob: bytes) -> np.ndarray:
    """4096 big-endian scalars -> (4096, 32) little-endian BigInt[255] array; every scalar must be < r."""
    if len(blob) != BYTES_PER_BLOB:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    be = np.frombuffer(blob, dtype=np.uint8).reshape(FIELD_ELEMENTS_PER_BLOB, 32)
    # vectorised range check against r (big-endian lexicographic compare)
    r_be = np.frombuffer(_R.to_bytes(32, "big"), dtype=np.uint8)
    diff = be.astype(np.int16) - r_be.astype(np.int16)
    first = np.argmax(diff != 0, axis=1)
    rows = np.arange(FIELD_ELEMENTS_PER_BLOB)
    lead = diff[rows, first]
    if np.any(lead >= 0):  # >= r  (lead == 0 only when all bytes equal, i.e. the scalar is r itself)
        raise KzgError(cttEthKzgStatus.cttEthKzg_ScalarLargerThanCurveOrder)
    return np.ascontiguousarray(be[:, ::-1])


def blob_to_kzg_commitment(ctx: EthereumKZGContext, blob: bytes) -> bytes:
    """commitment = [p(tau)]_1 as 48 compressed bytes; raises KzgError(status) where the reference returns a status."""
    poly = blob_to_bigint_polynomial(blob)
    r = ctx._bases.msm(poly, coord="aff")
    return serialize_g1_compressed(_aff_from_mont_bytes(bytes(r)))


# ---- proofs: compute_kzg_proof / compute_blob_kzg_proof (ethereum_eip4844_kzg.nim:332-375, :409-444) -------------
# kzg_prove (commitments/kzg.nim:204-223): quotient polynomial in evaluation form over the bit-reversed roots of unity,
# then ONE 4096-point MSM against the Lagrange SRS -- the same cached-base MSM as the commitment.  The field
# arithmetic over Fr (4096 elements, one batched inversion) runs on the GPU too (ctt_hip_fr_quotient: round 3; rounds 1-2
# spent 4 ms per proof on it in host integers); the host formulas below remain for the branch the device leaves to the
# caller -- z one of the roots of unity -- and as what the device result is tested against.
_PRIMITIVE_ROOT_OF_UNITY = 7
_FIAT_SHAMIR_PROTOCOL_DOMAIN = b"FSBLOBVERIFY_V1_"
_domain_brp_cache = None


def _domain_brp():
    """The 4096 roots of unity in bit-reversed order (ctx.domain_brp, ethereum_kzg_srs.nim)."""
    global _domain_brp_cache
    if _domain_brp_cache is None:
        w = pow(_PRIMITIVE_ROOT_OF_UNITY, (_R - 1) // FIELD_ELEMENTS_PER_BLOB, _R)
        roots, x = [], 1
        for _ in range(FIELD_ELEMENTS_PER_BLOB):
            roots.append(x)
            x = x * w % _R
        _domain_brp_cache = _bit_reversal_permutation(roots)
    return _domain_brp_cache


def _batch_inverse(vals):
    """Montgomery's trick over Fr; every value must be non-zero."""
    n = len(vals)
    pre, run = [0] * n, 1
    for i, v in enumerate(vals):
        pre[i] = run
        run = run * v % _R
    inv = pow(run, -1, _R)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = inv * pre[i] % _R
        inv = inv * vals[i] % _R
    return out


def _bytes_to_bls_field(b32: bytes) -> int:
    if len(b32) != 32:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    v = int.from_bytes(b32, "big")
    if v >= _R:
        raise KzgError(cttEthKzgStatus.cttEthKzg_ScalarLargerThanCurveOrder)
    return v


def quotient_polynomial(poly, z):
    """getQuotientPoly (math/polynomials/polynomials.nim): -> (q, y) with y = p(z) and q = (p - y) / (X - z), both in
    evaluation form over the bit-reversed domain; z may be one of the roots of unity."""
    dom = _domain_brp()
    n = FIELD_ELEMENTS_PER_BLOB
    try:
        m = dom.index(z)
    except ValueError:
        m = -1
    if m < 0:
        inv = _batch_inverse([(z - w) % _R for w in dom])                       # 1 / (z - w_i)
        s = sum(p * w % _R * iv for p, w, iv in zip(poly, dom, inv)) % _R
        y = (pow(z, n, _R) - 1) * pow(n, -1, _R) % _R * s % _R                     # barycentric evaluation
        q = [(y - p) * iv % _R for p, iv in zip(poly, inv)]                         # (p_i - y) / (w_i - z)
        return q, y
    y = poly[m]
    others = [i for i in range(n) if i != m]
    inv = _batch_inverse([(dom[i] - z) % _R for i in others])                      # 1 / (w_i - z)
    q = [0] * n
    zinv = pow(z, -1, _R)
    acc = 0
    for i, iv in zip(others, inv):
        q[i] = (poly[i] - y) * iv % _R
        acc += q[i] * dom[i] % _R * zinv                                           # q_m = - sum q_i * w_i / z
    q[m] = (-acc) % _R
    return q, y


_FR_MONT_R = 1 << 256


def quotient_polynomial_device(ctx: EthereumKZGContext, poly_le: np.ndarray, z: int):
    """(q on the device as a torch uint8 tensor (n, 32) of canonical little-endian scalars, y as an int), or None when z is
    one of the roots of unity (ctt_hip_fr_quotient returns -2: the host formula handles that branch)."""
    import ctypes

    import torch

    from . import _lib
    L = _lib.lib()
    n = FIELD_ELEMENTS_PER_BLOB
    dev = getattr(ctx, "_fr_dev", None)
    if dev is None:     # the domain as Montgomery residues and the output buffer, resident like the SRS
        dom = np.frombuffer(b"".join((w * _FR_MONT_R % _R).to_bytes(32, "little") for w in _domain_brp()), dtype=np.uint8)
        dev = ctx._fr_dev = (torch.from_numpy(dom.reshape(n, 32).copy()).cuda(), torch.empty((n, 32), dtype=torch.uint8, device="cuda"))
    d_dom, d_q = dev
    d_poly = torch.from_numpy(np.ascontiguousarray(poly_le)).cuda()
    if L.ctt_hip_msm_wait_stream(ctx.hip_ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0:
        raise RuntimeError("ctt_hip_msm_wait_stream failed")
    y = np.zeros(32, dtype=np.uint8)
    zb = np.frombuffer(z.to_bytes(32, "little"), dtype=np.uint8).copy()
    rc = L.ctt_hip_fr_quotient(ctx.hip_ctx, 0, ctypes.c_void_p(d_q.data_ptr()), y.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_void_p(d_poly.data_ptr()), ctypes.c_void_p(d_dom.data_ptr()),
                               zb.ctypes.data_as(ctypes.c_void_p), n)
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError("ctt_hip_fr_quotient failed")
    return d_q, int.from_bytes(bytes(y), "little")


def _prove(ctx: EthereumKZGContext, blob: bytes, z: int):
    poly_le = blob_to_bigint_polynomial(blob)
    dev = quotient_polynomial_device(ctx, poly_le, z)
    if dev is not None:
        d_q, y = dev
        r = ctx._bases.msm(d_q, coord="aff")     # the quotient never leaves the GPU
    else:                                        # z is a root of unity: the reference's other formula, on the host
        poly = [int.from_bytes(bytes(row), "little") for row in poly_le]
        q, y = quotient_polynomial(poly, z)
        q_le = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in q), dtype=np.uint8).reshape(FIELD_ELEMENTS_PER_BLOB, 32)
        r = ctx._bases.msm(q_le, coord="aff")
    return serialize_g1_compressed(_aff_from_mont_bytes(bytes(r))), y.to_bytes(32, "big")


def compute_kzg_proof(ctx: EthereumKZGContext, blob: bytes, z_bytes: bytes):
    """-> (proof 48 B, y 32 B): [proof]_1 = [(p(tau) - p(z)) / (tau - z)]_1 and y = p(z)."""
    z = _bytes_to_bls_field(z_bytes)
    return _prove(ctx, blob, z)


def _subgroup_check_g1(ctx: EthereumKZGContext, P):
    """[r]P == neutral on the GPU the context's SRS lives on (the reference validates commitments the same way it
    validates any deserialised point: on the curve and in the prime-order subgroup)."""
    if P is None:
        return
    from .msm import subgroup_check
    pt = np.frombuffer(_aff_mont_bytes(P), dtype=np.uint8).reshape(1, 96)
    if not subgroup_check("bls12_381_g1", pt, ctx=ctx.hip_ctx)[0]:
        raise KzgError(cttEthKzgStatus.cttEthKzg_EccPointNotInSubGroup)


def compute_challenge(blob: bytes, commitment_bytes: bytes) -> int:
    """Fiat-Shamir challenge of the blob proof (`fiatShamirChallenge`, ethereum_eip4844_kzg.nim:126-148):
    sha256(domain | 16-byte big-endian degree | blob | commitment) reduced mod r."""
    import hashlib
    data = (_FIAT_SHAMIR_PROTOCOL_DOMAIN + (0).to_bytes(8, "big") + FIELD_ELEMENTS_PER_BLOB.to_bytes(8, "big")
            + blob + commitment_bytes)
    return int.from_bytes(hashlib.sha256(data).digest(), "big") % _R


def compute_blob_kzg_proof(ctx: EthereumKZGContext, blob: bytes, commitment_bytes: bytes) -> bytes:
    """Proof that the blob matches the commitment, at the Fiat-Shamir challenge."""
    if len(blob) != BYTES_PER_BLOB or len(commitment_bytes) != 48:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    _subgroup_check_g1(ctx, deserialize_g1_compressed(commitment_bytes))
    blob_to_bigint_polynomial(blob)  # validates the field elements before hashing, as the reference does
    return _prove(ctx, blob, compute_challenge(blob, commitment_bytes))[0]
