"""
EIP-4844 blob -> KZG commitment and opening proofs on top of the MSM engine (the MSM's immediate caller):
ctypes callers of the reference's own C symbols, which libctt_msm_hip.so exports (include/ctt_msm_hip.h part 3,
constantine_amd/csrc/protocols.hip):

    ctt_eth_kzg_blob_to_kzg_commitment    include/constantine/protocols/ethereum_eip4844_kzg.h:106   (ethereum_eip4844_kzg.nim:297-330)
    ctt_eth_kzg_compute_kzg_proof         :126                                                       (:332-375)
    ctt_eth_kzg_compute_blob_kzg_proof    :153                                                       (:409-444)
    ctt_eth_kzg_context_new / _delete     :200, :238

Wire parsing, range checks, the Fiat-Shamir hash and point (de)compression are C++ on the host, every MSM, the subgroup check
and the quotient polynomial run on the GPU.  Nothing is computed in Python here; the only checks this module adds are the byte
LENGTHS, which the C signatures fix by type (ctt_eth_kzg_blob = 131072 bytes ...) and a Python `bytes` does not.  Verification
needs pairings and is out of scope, like the cell / PeerDAS functions.
"""
import ctypes
from enum import Enum

import numpy as np

from . import _lib

FIELD_ELEMENTS_PER_BLOB = 4096
BYTES_PER_FIELD_ELEMENT = 32
BYTES_PER_BLOB = FIELD_ELEMENTS_PER_BLOB * BYTES_PER_FIELD_ELEMENT
BYTES_PER_COMMITMENT = 48


class cttEthKzgStatus(Enum):  # ethereum_eip4844_kzg.h:29-40
    cttEthKzg_Success = 0
    cttEthKzg_VerificationFailure = 1
    cttEthKzg_InputsLengthsMismatch = 2
    cttEthKzg_ScalarZero = 3
    cttEthKzg_ScalarLargerThanCurveOrder = 4
    cttEthKzg_EccInvalidEncoding = 5
    cttEthKzg_EccCoordinateGreaterThanOrEqualModulus = 6
    cttEthKzg_EccPointNotOnCurve = 7
    cttEthKzg_EccPointNotInSubGroup = 8
    cttEthKzg_CellIndicesNotAscending = 9


class cttEthTrustedSetupStatus(Enum):  # ethereum_eip4844_kzg.h:63-67
    cttEthTS_Success = 0
    cttEthTS_MissingOrInaccessibleFile = 1
    cttEthTS_InvalidFile = 2


class KzgError(ValueError):
    def __init__(self, status):
        super().__init__(status.name)
        self.status = status


def _check(rc):
    if rc == _lib.GPU_UNAVAILABLE:      # CTT_HIP_STATUS_GPU_UNAVAILABLE: outside the reference's enum, nothing was written
        raise _lib.GpuUnavailable("EIP-4844 KZG")
    if rc != 0:
        raise KzgError(cttEthKzgStatus(rc))


def _buf(b):
    return (ctypes.c_uint8 * len(b)).from_buffer_copy(b)


class EthereumKZGContext:
    """ctt_eth_kzg_context: srs_lagrange_brp_g1 cached on the GPU (constantine/commitments_setups/ethereum_kzg_srs.nim)."""

    def __init__(self, srs_lagrange_g1_compressed: bytes, device=0, table=True):
        """`srs_lagrange_g1_compressed`: 4096 x 48 bytes, the G1 Lagrange points in ceremony (file) order
        (ctt_hip_eth_kzg_context_from_srs).  table=True (what ctt_eth_kzg_context_new does) caches the SRS as a window table."""
        self.L = _lib.lib()
        if len(srs_lagrange_g1_compressed) != FIELD_ELEMENTS_PER_BLOB * 48:
            raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
        h = ctypes.c_void_p()
        rc = self.L.ctt_hip_eth_kzg_context_from_srs(ctypes.byref(h), _buf(srs_lagrange_g1_compressed), FIELD_ELEMENTS_PER_BLOB,
                                                     int(device), 1 if table else 0)
        if rc == _lib.GPU_UNAVAILABLE:
            raise _lib.GpuUnavailable("KZG context")
        if rc != 0:
            raise ValueError(cttEthTrustedSetupStatus(rc).name)
        self.handle = h

    @classmethod
    def from_ckzg_text(cls, path, precompute=None):
        """ctt_eth_kzg_context_new(ctx, filepath, cttEthTSFormat_ckzg4844): the c-kzg text format of the Ethereum ceremony, as
        shipped by the reference in constantine/commitments_setups/trusted_setup_ethereum_kzg4844_reference.dat.
        precompute = (t, b): through ctt_eth_kzg_context_new_with_precompute (the reference's table sizes; the same context here)."""
        self = cls.__new__(cls)
        self.L = _lib.lib()
        h = ctypes.c_void_p()
        if precompute is not None:
            rc = self.L.ctt_eth_kzg_context_new_with_precompute(ctypes.byref(h), str(path).encode(), 0, int(precompute[0]), int(precompute[1]))
        else:
            rc = self.L.ctt_eth_kzg_context_new(ctypes.byref(h), str(path).encode(), 0)
        if rc == _lib.GPU_UNAVAILABLE:
            raise _lib.GpuUnavailable("KZG context")
        if rc != 0:
            raise ValueError(cttEthTrustedSetupStatus(rc).name)
        self.handle = h
        return self

    def delete(self):
        if self.handle:
            self.L.ctt_eth_kzg_context_delete(self.handle)
            self.handle = None


def blob_to_kzg_commitment(ctx: EthereumKZGContext, blob: bytes) -> bytes:
    """commitment = [p(tau)]_1 as 48 compressed bytes; raises KzgError(status) where the reference returns a status."""
    if len(blob) != BYTES_PER_BLOB:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    out = (ctypes.c_uint8 * 48)()
    _check(ctx.L.ctt_eth_kzg_blob_to_kzg_commitment(ctx.handle, out, _buf(blob)))
    return bytes(out)


def compute_kzg_proof(ctx: EthereumKZGContext, blob: bytes, z_bytes: bytes):
    """-> (proof 48 B, y 32 B): [proof]_1 = [(p(tau) - p(z)) / (tau - z)]_1 and y = p(z)."""
    if len(blob) != BYTES_PER_BLOB or len(z_bytes) != 32:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    proof, y = (ctypes.c_uint8 * 48)(), (ctypes.c_uint8 * 32)()
    _check(ctx.L.ctt_eth_kzg_compute_kzg_proof(ctx.handle, proof, y, _buf(blob), _buf(z_bytes)))
    return bytes(proof), bytes(y)


def compute_blob_kzg_proof(ctx: EthereumKZGContext, blob: bytes, commitment_bytes: bytes) -> bytes:
    """Proof that the blob matches the commitment, at the Fiat-Shamir challenge."""
    if len(blob) != BYTES_PER_BLOB or len(commitment_bytes) != 48:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    proof = (ctypes.c_uint8 * 48)()
    _check(ctx.L.ctt_eth_kzg_compute_blob_kzg_proof(ctx.handle, proof, _buf(blob), _buf(commitment_bytes)))
    return bytes(proof)


# the reference's _parallel forms (ethereum_eip4844_kzg_parallel.nim: `tp` first): the same GPU path, the thread pool is not used
def blob_to_kzg_commitment_parallel(tp, ctx: EthereumKZGContext, blob: bytes) -> bytes:
    if len(blob) != BYTES_PER_BLOB:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    out = (ctypes.c_uint8 * 48)()
    _check(ctx.L.ctt_eth_kzg_blob_to_kzg_commitment_parallel(tp, ctx.handle, out, _buf(blob)))
    return bytes(out)


def compute_kzg_proof_parallel(tp, ctx: EthereumKZGContext, blob: bytes, z_bytes: bytes):
    if len(blob) != BYTES_PER_BLOB or len(z_bytes) != 32:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    proof, y = (ctypes.c_uint8 * 48)(), (ctypes.c_uint8 * 32)()
    _check(ctx.L.ctt_eth_kzg_compute_kzg_proof_parallel(tp, ctx.handle, proof, y, _buf(blob), _buf(z_bytes)))
    return bytes(proof), bytes(y)


def compute_blob_kzg_proof_parallel(tp, ctx: EthereumKZGContext, blob: bytes, commitment_bytes: bytes) -> bytes:
    if len(blob) != BYTES_PER_BLOB or len(commitment_bytes) != 48:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    proof = (ctypes.c_uint8 * 48)()
    _check(ctx.L.ctt_eth_kzg_compute_blob_kzg_proof_parallel(tp, ctx.handle, proof, _buf(blob), _buf(commitment_bytes)))
    return bytes(proof)


# ---- the host-only pieces of the library (no GPU), as the tests reach them ----------------------------------------------------
def g1_decompress(b48: bytes) -> bytes:
    """ctt_hip_bls12_381_g1_decompress: 48 compressed bytes -> affine Montgomery {x, y} (96 bytes, (0,0) = neutral)."""
    if len(b48) != 48:
        raise KzgError(cttEthKzgStatus.cttEthKzg_EccInvalidEncoding)
    out = (ctypes.c_uint8 * 96)()
    _check(_lib.lib().ctt_hip_bls12_381_g1_decompress(out, _buf(b48)))
    return bytes(out)


def g1_compress(aff96: bytes) -> bytes:
    out = (ctypes.c_uint8 * 48)()
    _lib.lib().ctt_hip_bls12_381_g1_compress(out, _buf(aff96))
    return bytes(out)


def blob_to_bigint_polynomial(blob: bytes) -> np.ndarray:
    """ctt_hip_eth_kzg_blob_to_scalars: 4096 big-endian scalars -> (4096, 32) little-endian BigInt[255]; each must be < r."""
    if len(blob) != BYTES_PER_BLOB:
        raise KzgError(cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch)
    out = np.zeros((FIELD_ELEMENTS_PER_BLOB, 32), dtype=np.uint8)
    _check(_lib.lib().ctt_hip_eth_kzg_blob_to_scalars(out.ctypes.data_as(ctypes.c_void_p), _buf(blob)))
    return out


def compute_challenge(blob: bytes, commitment_bytes: bytes) -> bytes:
    """ctt_hip_eth_kzg_challenge -> 32 big-endian bytes"""
    assert len(blob) == BYTES_PER_BLOB and len(commitment_bytes) == 48
    z = (ctypes.c_uint8 * 32)()
    _lib.lib().ctt_hip_eth_kzg_challenge(z, _buf(blob), _buf(commitment_bytes))
    return bytes(z)


def quotient_polynomial_host(poly_le: np.ndarray, z: int):
    """ctt_hip_eth_kzg_quotient_host -> (q as (4096, 32) uint8 little-endian, y as int)"""
    poly_le = np.ascontiguousarray(poly_le, dtype=np.uint8)
    assert poly_le.shape == (FIELD_ELEMENTS_PER_BLOB, 32)
    q = np.zeros_like(poly_le)
    y = (ctypes.c_uint8 * 32)()
    _lib.lib().ctt_hip_eth_kzg_quotient_host(q.ctypes.data_as(ctypes.c_void_p), y, poly_le.ctypes.data_as(ctypes.c_void_p),
                                             _buf(z.to_bytes(32, "little")))
    return q, int.from_bytes(bytes(y), "little")


def sha256(data: bytes) -> bytes:
    out = (ctypes.c_uint8 * 32)()
    _lib.lib().ctt_hip_sha256(out, _buf(data) if data else None, len(data))
    return bytes(out)
