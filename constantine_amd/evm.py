"""
EIP-2537 BLS12_G1MSM / BLS12_G2MSM precompiles on top of the MSM engine: ctypes callers of the reference's own
C symbols, which libctt_msm_hip.so exports (include/ctt_msm_hip.h part 3, constantine_amd/csrc/protocols.hip):

    ctt_eth_evm_bls12381_g1msm(r, r_len, inputs, inputs_len) -> ctt_evm_status     ethereum_evm_precompiles.h:386  (.nim:894-975)
    ctt_eth_evm_bls12381_g2msm(r, r_len, inputs, inputs_len) -> ctt_evm_status     ethereum_evm_precompiles.h:419  (.nim:977-1060)

Input: pairs of (point, scalar); a G1 point is 64 B x | 64 B y (big-endian, top 16 bytes zero, (0,0) = infinity), a G2 point is
x.c0 | x.c1 | y.c0 | y.c1; the scalar is 32 B big-endian, any value < 2^256.  Output: the affine sum, same encoding.  Parsing and
the curve checks are C++ on the host; the subgroup checks (one launch for all points) and the MSM run on the GPU.
"""
import ctypes
from enum import Enum

from . import _lib


class CttEVMStatus(Enum):  # ethereum_evm_precompiles.h:19-27
    cttEVM_Success = 0
    cttEVM_InvalidInputSize = 1
    cttEVM_InvalidOutputSize = 2
    cttEVM_IntLargerThanModulus = 3
    cttEVM_PointNotOnCurve = 4
    cttEVM_PointNotInSubgroup = 5
    cttEVM_VerificationFailure = 6


class EvmError(ValueError):
    def __init__(self, status):
        super().__init__(status.name)
        self.status = status


def _call(sym, out_len, inputs: bytes, r_len=None) -> bytes:
    L = _lib.lib()
    r = (ctypes.c_uint8 * out_len)()
    buf = (ctypes.c_uint8 * max(1, len(inputs))).from_buffer_copy(inputs if inputs else b"\0")
    rc = getattr(L, sym)(r, out_len if r_len is None else r_len, buf, len(inputs))
    if rc == _lib.GPU_UNAVAILABLE:      # CTT_HIP_STATUS_GPU_UNAVAILABLE: outside ctt_evm_status, r untouched
        raise _lib.GpuUnavailable(sym)
    if rc != 0:
        raise EvmError(CttEVMStatus(rc))
    return bytes(r)


def eth_evm_bls12381_g1msm(inputs: bytes, r_len=None) -> bytes:
    return _call("ctt_eth_evm_bls12381_g1msm", 128, inputs, r_len)


def eth_evm_bls12381_g2msm(inputs: bytes, r_len=None) -> bytes:
    return _call("ctt_eth_evm_bls12381_g2msm", 256, inputs, r_len)
