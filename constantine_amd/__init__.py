"""
constantine_amd -- MI355X-native multi-scalar-multiplication engine behind Constantine's MSM API.

Only what the MSM hot path needs lives here:
  csrc/     HIP kernels (gfx950) + the C ABI  (include/ctt_msm_hip.h)  -> libctt_msm_hip.so
  _lib.py   ctypes loader (fails loudly if the HIP library is missing)
  msm.py    host-side mirror of the reference's entry points (multiScalarMul_vartime[_parallel],
            Halo2-ZAL CttEngine.msm) over the C ABI
  parallel.py  point-sharded multi-GPU MSM, one process per GPU (torch.distributed / RCCL); the in-library form, one
            process driving several GPUs, is ctt_hip_msm_set_devices / $CTT_HIP_DEVICES (msm.set_devices)
  synth.py  synthetic benchmark inputs (seeded scalars)
  kzg.py, evm.py  ctypes callers of the reference's KZG / EIP-2537 MSM-precompile C symbols (csrc/protocols.hip: host side in C++)
"""
from .curves import CURVES, CurveInfo  # noqa: F401
from .msm import (  # noqa: F401
    CachedBases,
    CttEngine,
    DeviceMsm,
    MsmRefused,
    msm_available,
    msm_host,
    batchAffine_vartime,
    sum_reduce_vartime,
    multiScalarMul_vartime,
    multiScalarMul_vartime_parallel,
    set_devices,
    set_shard_min,
)
