import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from constantine_amd import DeviceMsm, CURVES
P = {"bls12_381": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
     "bn254_snarks": 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47}
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
        # Z = 1 in the C-API representation: the Montgomery residue R mod p, little-endian (c1 = 0 for the quadratic extension)
        p = P[name.rsplit("_", 1)[0]]
        nb = (p.bit_length() + 63) // 64 * 8
        one = np.frombuffer(((1 << (8 * nb)) % p).to_bytes(nb, "little") + bytes(info.coord_bytes - nb), dtype=np.uint8)
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
