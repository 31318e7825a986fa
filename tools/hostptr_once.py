#!/usr/bin/env python3
"""Four host-pointer calls of the Constantine symbol at 2^20 BLS12-381 G1 pairs (pageable arrays): the workload of tools/timeline_copies.py.
    (cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d OUT -o p -- python tools/hostptr_once.py); python tools/timeline_copies.py OUT/.../p_results.db"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel
from constantine_amd.synth import synth_scalars
name = "bls12_381_g1"; info = CURVES[name]; n = 1 << 20
eng = DeviceMsm(0)
d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
eng.gen_points(name, 5, n, d); pts = d.cpu().numpy(); eng.close()
sc = synth_scalars(6, n, 255)
for i in range(4):
    t0 = time.perf_counter(); multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac"); print("call", i, (time.perf_counter() - t0) * 1e3, "ms", flush=True)
