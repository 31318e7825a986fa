#!/usr/bin/env python3
"""Host-pointer call of the Constantine symbol (2^20 BLS12-381 G1 pairs, pageable arrays) on the shared chip and with the chip partitioned
($CTT_HIP_CU_TAIL = 4 / 8: profiles/cu_mask_r06.txt): median of 10 calls per configuration, each in a process of its own."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time
    sys.path.insert(0, ROOT)
    import torch
    from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel
    from constantine_amd.synth import synth_scalars
    name = "bls12_381_g1"
    info = CURVES[name]
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)      # (a partitioned context's streams are blocking streams: keep torch off the null stream)
    for lg in (18, 20):
        n = 1 << lg
        eng = DeviceMsm(0)
        d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
        eng.gen_points(name, 5, n, d)
        pts = d.cpu().numpy()
        eng.close()
        sc = synth_scalars(6, n, 255)
        ts = []
        for i in range(13):
            t0 = time.perf_counter()
            r = multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[3:])
        print(f"CTT_HIP_CU_TAIL={os.environ.get('CTT_HIP_CU_TAIL', '0')}: 2^{lg} pairs, host pointers: median {ts[len(ts) // 2]:.3f} ms per call (min {ts[0]:.3f})", flush=True)
else:
    for r in ("0", "4", "8", "0"):
        env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
        env.pop("CTT_HIP_CU_TAIL", None)
        if r != "0":
            env["CTT_HIP_CU_TAIL"] = r
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=env)
