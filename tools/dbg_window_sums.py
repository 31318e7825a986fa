import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from constantine_amd import multiScalarMul_vartime, _lib
from oracle import cref, pyoracle as po
L = _lib.lib()
for name in ("bn254_snarks_g1", "bls12_381_g1", "bn254_snarks_g2"):
    curve = po.CURVES[name]
    for n, sval in ((1, 1), (1, 2), (1, 3), (1, 5), (2, 1), (1, None), (50, None)):
        pts = cref.gen_points(name, 9, n)
        if sval is None:
            sc = cref.synth_scalars(10 + n, n, curve.scalar_bits)
        else:
            sc = curve.scalars_to_array([sval] * n)
        exp = bytes(cref.msm(name, sc, pts)[0])
        out = {}
        for hw in (0, 1):
            L.ctt_hip_msm_set_option(None, b"host_window_sums", hw)
            r = multiScalarMul_vartime(name, sc, pts, coord="jac")
            out[hw] = curve.jac_from_bytes(bytes(r)) == curve.aff_from_bytes(exp)
        print(name, n, sval, "device sums ok:", out[0], " host sums ok:", out[1])
