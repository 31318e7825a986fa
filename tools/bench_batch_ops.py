import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from constantine_amd import DeviceMsm, CURVES
from oracle import pyoracle as po
for name in ("bls12_381_g1", "bn254_snarks_g1", "bls12_381_g2"):
    info = CURVES[name]
    eng = DeviceMsm(0)
    for lg in (16, 20, 22):
        n = 1 << lg
        d_pts = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
        eng.gen_points(name, 77, n, d_pts)
        eng.sum_reduce(name, d_pts, n)
        t0 = time.perf_counter()
        for _ in range(5): eng.sum_reduce(name, d_pts, n)
        t_sum = (time.perf_counter() - t0) / 5
        F = po.CURVES[name].F
        one = np.frombuffer(F.to_mont_bytes(1 if F.degree == 1 else (1, 0)), dtype=np.uint8)
        d_jac = torch.empty((n, 3 * info.coord_bytes), dtype=torch.uint8, device="cuda")
        d_jac[:, :info.aff_bytes] = d_pts
        d_jac[:, info.aff_bytes:] = torch.from_numpy(one.copy()).cuda()
        d_out = torch.empty_like(d_pts)
        eng.batch_affine(name, d_out, d_jac, n)
        t0 = time.perf_counter()
        for _ in range(5): eng.batch_affine(name, d_out, d_jac, n)
        t_ba = (time.perf_counter() - t0) / 5
        print(f"{name} 2^{lg}: sum_reduce {t_sum*1e3:.3f} ms ({n/t_sum/1e6:.0f} M pts/s)  batch_affine {t_ba*1e3:.3f} ms ({n/t_ba/1e6:.0f} M pts/s)", flush=True)
        del d_pts, d_jac, d_out
    eng.close()
