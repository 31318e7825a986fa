#!/usr/bin/env python3
"""Per-step finish times of a pipelined loop with `depth` MSMs in flight (round 6: the occasional slow loop with three in flight):
    python tools/depth_trace.py <log2n> <depth> [loops=8] [steps=60]"""
import collections
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from constantine_amd import DeviceMsm  # noqa: E402
from constantine_amd.msm import CURVES  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

curve = "bls12_381_g1"
log2n, depth = int(sys.argv[1]), int(sys.argv[2])
loops = int(sys.argv[3]) if len(sys.argv) > 3 else 8
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
info = CURVES[curve]
n = 1 << log2n
eng = DeviceMsm(0)
d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
eng.gen_points(curve, 0x5EED0002, n, d_points)
d_scal = torch.from_numpy(synth_scalars(0x5EED0003, n, info.scalar_bits)).cuda()
torch.cuda.synchronize()
for loop in range(loops):
    pend, sub, stamps = collections.deque(), 0, []
    t0 = time.perf_counter()
    for _ in range(steps):
        while sub < steps and len(pend) < depth:
            pend.append(eng.submit(curve, d_scal, d_points, n))
            sub += 1
        eng.finish(pend.popleft(), coord="aff")
        stamps.append((time.perf_counter() - t0) * 1e3)
    eng.sync()
    d = [stamps[0]] + [b - a for a, b in zip(stamps, stamps[1:])]
    slow = [(i, round(x, 2)) for i, x in enumerate(d) if x > 2.0 * sorted(d)[len(d) // 2]]
    print(f"2^{log2n} depth {depth} loop {loop}: {stamps[-1] / steps:.4f} ms per MSM, median step {sorted(d)[len(d) // 2]:.3f}, max {max(d):.2f}; steps > 2 x median: {slow[:12]}", flush=True)
eng.close()
