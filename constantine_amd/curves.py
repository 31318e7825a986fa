"""Curve table for the host-side mirror (ids match include/ctt_msm_hip.h)."""
from dataclasses import dataclass


@dataclass(frozen=True)
class CurveInfo:
    name: str          # python-side name
    cid: int           # CTT_HIP_* id
    sym: str           # C symbol stem: ctt_<sym>_{jac,prj}_multi_scalar_mul_...
    coord_bytes: int   # one coordinate (Fp or Fp2)
    scalar_bits: int   # BigInt[bits] of the C API (big255 / big254)
    has_parallel: bool  # upstream exports *_vartime_parallel for this group

    @property
    def aff_bytes(self):
        return 2 * self.coord_bytes

    @property
    def jac_bytes(self):
        return 3 * self.coord_bytes


CURVES = {
    "bls12_381_g1": CurveInfo("bls12_381_g1", 0, "bls12_381_g1", 48, 255, True),
    "bls12_381_g2": CurveInfo("bls12_381_g2", 1, "bls12_381_g2", 96, 255, False),
    "bn254_snarks_g1": CurveInfo("bn254_snarks_g1", 2, "bn254_snarks_g1", 32, 254, True),
    "bn254_snarks_g2": CurveInfo("bn254_snarks_g2", 3, "bn254_snarks_g2", 64, 254, False),
    "pallas": CurveInfo("pallas", 4, "pallas_ec", 32, 255, True),
    "vesta": CurveInfo("vesta", 5, "vesta_ec", 32, 255, True),
}

OUT_AFF, OUT_JAC, OUT_PRJ = 0, 1, 2
COEF_BIG, COEF_FR = 0, 1
