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

Only the commitment is implemented (the MSM caller); proofs, verification and cell/PeerDAS functions are out of scope.
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
