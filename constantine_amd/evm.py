"""
EIP-2537 BLS12_G1MSM / BLS12_G2MSM precompile wire format on top of the MSM engine.

Mirrors the reference's
    eth_evm_bls12381_g1msm(r, inputs) -> CttEVMStatus     constantine/ethereum_evm_precompiles.nim:894-975
    eth_evm_bls12381_g2msm(r, inputs) -> CttEVMStatus     constantine/ethereum_evm_precompiles.nim:977-...
Input: pairs of (point, scalar); a G1 point is 64 B x | 64 B y (big-endian, top 16 bytes zero, (0,0) = infinity), a G2
point is x.c0 | x.c1 | y.c0 | y.c1; the scalar is 32 B big-endian, any value < 2^256 (reduced mod r here, as upstream).
Points are checked to be on the curve and in the prime-order subgroup.  Output: the affine sum, same encoding.

The curve checks run on the host in plain integer arithmetic; the subgroup checks ([r]P = infinity, all points in one
launch: ctt_hip_subgroup_check) and the MSM itself (the Constantine-compatible C symbol) run on the GPU.
"""
from enum import Enum

import numpy as np

from .msm import multiScalarMul_vartime

_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
_MONT = 1 << 384
_MONT_INV = pow(_MONT, -1, _P)


class CttEVMStatus(Enum):  # constantine/ethereum_evm_precompiles.nim (the members these two precompiles return)
    cttEVM_Success = 0
    cttEVM_InvalidInputSize = 1
    cttEVM_IntLargerThanModulus = 3
    cttEVM_PointNotOnCurve = 4
    cttEVM_PointNotInSubgroup = 5


class EvmError(ValueError):
    def __init__(self, status):
        super().__init__(status.name)
        self.status = status


# ---- host-side field helpers (plain integers; only for parsing / on-curve checks / output normalisation) ------
def _fp(b64: bytes) -> int:
    if any(b64[:16]):
        raise EvmError(CttEVMStatus.cttEVM_IntLargerThanModulus)   # "invalid field element top bytes"
    v = int.from_bytes(b64, "big")
    if v >= _P:
        raise EvmError(CttEVMStatus.cttEVM_IntLargerThanModulus)
    return v


def _fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % _P, (a[0] * b[1] + a[1] * b[0]) % _P)


def _mont(v: int) -> bytes:
    return ((v * _MONT) % _P).to_bytes(48, "little")


def _unmont(b: bytes) -> int:
    return int.from_bytes(b, "little") * _MONT_INV % _P


def _jac_to_affine(F2: bool, r: np.ndarray):
    """(X, Y, Z) Montgomery bytes -> affine integer coordinates or None."""
    nb = 96 if F2 else 48
    raw = bytes(r)

    def el(i):
        b = raw[i * nb:(i + 1) * nb]
        return (_unmont(b[:48]), _unmont(b[48:96])) if F2 else _unmont(b)

    X, Y, Z = el(0), el(1), el(2)
    if F2:
        if Z == (0, 0):
            return None
        n = pow(Z[0] * Z[0] + Z[1] * Z[1], -1, _P)
        zi = (Z[0] * n % _P, -Z[1] * n % _P)
        zi2 = _fp2_mul(zi, zi)
        return _fp2_mul(X, zi2), _fp2_mul(Y, _fp2_mul(zi2, zi))
    if Z == 0:
        return None
    zi = pow(Z, -1, _P)
    return X * zi * zi % _P, Y * zi * zi * zi % _P


_OK, _TOO_LARGE, _OFF_CURVE = 0, 1, 2


def _fp_or_none(b64: bytes):
    try:
        return _fp(b64)
    except EvmError:
        return None


def _validate(curve: str, pts: np.ndarray, parse):
    """The reference handles the pairs in order, each point fully (coordinates below the modulus, on the curve, in the
    subgroup: ethereum_evm_precompiles.nim fromRawCoords) before the next one: the status is that of the FIRST offending
    pair.  `parse[i]` is the host-side verdict on pair i (_OK, _TOO_LARGE, _OFF_CURVE; rows of pairs that did not parse are
    the neutral point).  The subgroup checks ([r]P = neutral) of all points run as one GPU launch."""
    from .msm import subgroup_check
    bad = next((i for i, st in enumerate(parse) if st != _OK), None)
    if bad is None:
        if not subgroup_check(curve, pts).all():
            raise EvmError(CttEVMStatus.cttEVM_PointNotInSubgroup)
        return
    # an earlier point outside the subgroup comes first: check the points in front of the offending pair
    if bad and not subgroup_check(curve, pts[:bad]).all():
        raise EvmError(CttEVMStatus.cttEVM_PointNotInSubgroup)
    raise EvmError(CttEVMStatus.cttEVM_IntLargerThanModulus if parse[bad] == _TOO_LARGE else CttEVMStatus.cttEVM_PointNotOnCurve)


def _scalars(recs, off):
    ks = [int.from_bytes(rec[off:off + 32], "big") % _R for rec in recs]
    return np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), dtype=np.uint8).reshape(len(ks), 32)


def eth_evm_bls12381_g1msm(inputs: bytes) -> bytes:
    if len(inputs) == 0 or len(inputs) % 160 != 0:
        raise EvmError(CttEVMStatus.cttEVM_InvalidInputSize)
    recs = [inputs[i:i + 160] for i in range(0, len(inputs), 160)]
    rows, parse = [], []
    for rec in recs:
        x, y = _fp_or_none(rec[0:64]), _fp_or_none(rec[64:128])
        if x is None or y is None:
            parse.append(_TOO_LARGE)
            rows.append(bytes(96))
            continue
        on_curve = (x == 0 and y == 0) or (y * y - x * x * x - 4) % _P == 0
        parse.append(_OK if on_curve else _OFF_CURVE)
        rows.append(_mont(x) + _mont(y) if on_curve else bytes(96))
    pts = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), 96)
    _validate("bls12_381_g1", pts, parse)
    res = _jac_to_affine(False, multiScalarMul_vartime("bls12_381_g1", _scalars(recs, 128), pts, coord="jac"))
    if res is None:
        return bytes(128)
    return res[0].to_bytes(64, "big") + res[1].to_bytes(64, "big")


def eth_evm_bls12381_g2msm(inputs: bytes) -> bytes:
    if len(inputs) == 0 or len(inputs) % 288 != 0:
        raise EvmError(CttEVMStatus.cttEVM_InvalidInputSize)
    recs = [inputs[i:i + 288] for i in range(0, len(inputs), 288)]
    rows, parse = [], []
    for rec in recs:
        co = [_fp_or_none(rec[i:i + 64]) for i in (0, 64, 128, 192)]
        if any(v is None for v in co):
            parse.append(_TOO_LARGE)
            rows.append(bytes(192))
            continue
        x, y = (co[0], co[1]), (co[2], co[3])
        ok = True
        if not (x == (0, 0) and y == (0, 0)):
            x3 = _fp2_mul(_fp2_mul(x, x), x)
            y2 = _fp2_mul(y, y)
            ok = ((y2[0] - x3[0] - 4) % _P, (y2[1] - x3[1] - 4) % _P) == (0, 0)   # b' = 4(1 + i)
        parse.append(_OK if ok else _OFF_CURVE)
        rows.append(_mont(x[0]) + _mont(x[1]) + _mont(y[0]) + _mont(y[1]) if ok else bytes(192))
    pts = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), 192)
    _validate("bls12_381_g2", pts, parse)
    res = _jac_to_affine(True, multiScalarMul_vartime("bls12_381_g2", _scalars(recs, 256), pts, coord="jac"))
    if res is None:
        return bytes(256)
    (x0, x1), (y0, y1) = res
    return b"".join(v.to_bytes(64, "big") for v in (x0, x1, y0, y1))
