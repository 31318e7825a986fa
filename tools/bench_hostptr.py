#!/usr/bin/env python3
"""PCIe-inclusive rate of the Constantine-compatible symbol (host pointers, pageable memory), DESIGN.md section 4.
Also times the KZG commitment path (cached SRS, host scalars).  Never the headline `value` (inputs there are in HBM)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (device generation of the points)
from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402
from constantine_amd import _lib  # noqa: E402

name = "bls12_381_g1"
info = CURVES[name]
for lg in (16, 18, 20, 22):
    n = 1 << lg
    eng = DeviceMsm(0)
    d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(name, 5, n, d)
    pts = d.cpu().numpy()
    eng.close()
    sc = synth_scalars(6, n, 255)
    for chunks in (0, 1, 2, 3, 4, 5, 6):
        _lib.lib().ctt_hip_msm_set_option(None, b"chunks", chunks)
        multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")
        reps = 10
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        print(f"ctt_bls12_381_g1_jac_multi_scalar_mul_big_coefs_vartime_parallel, N=2^{lg}, host pointers (pageable), "
              f"upload in {chunks if chunks else 'auto'} slice(s): median {dt * 1e3:.3f} ms per MSM = {n / dt / 1e6:.1f} M pairs/s")
    _lib.lib().ctt_hip_msm_set_option(None, b"chunks", 0)
