#!/usr/bin/env python3
"""What several engine contexts on ONE GPU do to a host-pointer call: the in-library sharding
(ctt_hip_msm_set_devices) with the same device listed 1, 2, 3, 4 times -- each context has its own streams, workspace and host
thread doing pageable H2D copies at the same time, which is the contention an 8-GPU node's eight uploader threads see on the host
side (the PCIe link itself is shared here and per GPU there).  One line per (size, contexts): median ms per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel, set_devices, set_shard_min  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

name = "bls12_381_g1"
info = CURVES[name]
set_shard_min(1 << 12)
for lg in (18, 20, 22):
    n = 1 << lg
    eng = DeviceMsm(0)
    d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(name, 5, n, d)
    pts = d.cpu().numpy()
    eng.close()
    sc = synth_scalars(6, n, 255)
    ref = None
    for nctx in (1, 2, 3, 4, 8):
        set_devices([0] * nctx if nctx > 1 else [])
        r = multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")
        ref = ref if ref is not None else bytes(r)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            r = multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        print(f"N=2^{lg} host pointers, {nctx} context(s) on device 0: median {dt * 1e3:.3f} ms per call = {n / dt / 1e6:.1f} M pairs/s, "
              f"same result: {bytes(r) == ref}", flush=True)
    set_devices([])
