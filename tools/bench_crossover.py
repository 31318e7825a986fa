#!/usr/bin/env python3
"""Where the GPU takes over from the CPU (INTEGRATION.md part B's CttHipMinPoints): for N = 2^6 .. 2^16 BLS12-381 G1 pairs, the host-pointer
call of the drop-in symbol (PCIe included, what a Constantine caller gets) and the device-resident blocking call, with Constantine's published
16-thread laptop figures beside them.  The CPU port's side of the comparison is `python bench.py --cpu-only --log2n K` (the one place outside tests/
that runs the oracle): tools/collect_round.sh writes both.  One JSON line per size."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

PUBLISHED_MS = {10: 1.660, 12: 4.542, 14: 14.727, 15: 25.009, 16: 51.280}   # Ryzen 7 7840U, 16 threads
name = "bls12_381_g1"
info = CURVES[name]
eng = DeviceMsm(0)


def med(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


for lg in (6, 8, 10, 11, 12, 13, 14, 16):
    n = 1 << lg
    d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(name, 5, n, d)
    pts = d.cpu().numpy()
    sc = synth_scalars(6, n, 255)
    d_s = torch.from_numpy(sc).cuda()
    host = med(lambda: multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac"), 30)
    dev = med(lambda: eng.msm(name, d_s, d, n, coord="aff"), 30)
    print(json.dumps({"log2n": lg, "gpu_hostptr_ms": round(host, 4), "gpu_device_resident_ms": round(dev, 4),
                      "constantine_published_16_threads_ms_other_hw": PUBLISHED_MS.get(lg)}), flush=True)
eng.close()
