#!/usr/bin/env python3
"""The Constantine symbol ctt_<curve>_jac_batch_affine on HOST arrays (the reference's call shape): ms per call.  Run twice for the A/B --
CTT_HIP_BATCH_AFFINE_SLICES=1 (one upload, the kernel, one download: rounds 2-3) and unset (slices, uploads on a helper thread: round 4)."""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from constantine_amd import CURVES, DeviceMsm, _lib  # noqa: E402
from constantine_amd.msm import _ptr  # noqa: E402

L = _lib.lib()
P = {"bls12_381_g1": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
     "bn254_snarks_g1": 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47}
for name in ("bls12_381_g1", "bn254_snarks_g1"):
    info = CURVES[name]
    cb = info.coord_bytes
    for lg in (16, 18, 20, 22):
        n = 1 << lg
        eng = DeviceMsm(0)
        d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
        eng.gen_points(name, 11, n, d)
        aff = d.cpu().numpy()
        eng.close()
        one = np.frombuffer(((1 << (8 * cb)) % P[name]).to_bytes(cb, "little"), dtype=np.uint8)      # Z = 1 as a Montgomery residue
        jac = np.zeros((n, 3 * cb), dtype=np.uint8)
        jac[:, :2 * cb] = aff
        jac[:, 2 * cb:] = one
        dst = np.zeros((n, info.aff_bytes), dtype=np.uint8)
        fn = getattr(L, f"ctt_{info.sym}_jac_batch_affine")
        for _ in range(2):
            fn(_ptr(dst), _ptr(jac), n)
        assert bytes(dst) == bytes(aff)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            fn(_ptr(dst), _ptr(jac), n)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"ctt_{info.sym}_jac_batch_affine, 2^{lg} points on the host: {statistics.median(ts):.3f} ms (min {min(ts):.3f}); "
              f"{n * (3 * cb + 2 * cb) / 1e6:.0f} MB over the link", flush=True)
