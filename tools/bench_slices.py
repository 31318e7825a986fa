#!/usr/bin/env python3
"""Does the accumulate kernel slow down at 2^22 / 2^24 pairs because its record gathers leave the 256 MiB Infinity Cache?
The host-pointer entry accumulates the pairs in `chunks` slices (MsmEngine::submit_host), every slice gathering from its own
part of the record array only: the summed accumulate stage time over the slices against one slice is the answer.
    python tools/bench_slices.py <log2n> [chunks ...]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel  # noqa: E402
from constantine_amd import _lib  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

name = "bls12_381_g1"
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
chunk_list = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
info = CURVES[name]
n = 1 << lg
eng = DeviceMsm(0)
d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
eng.gen_points(name, 5, n, d)
pts = d.cpu().numpy()
eng.close()
del d
sc = synth_scalars(6, n, 255)
L = _lib.lib()
L.ctt_hip_msm_set_option(None, b"timings", 1)
ms = (ctypes.c_float * 8)()
for chunks in chunk_list:
    L.ctt_hip_msm_set_option(None, b"chunks", chunks)
    multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")
    ts, stages = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")
        ts.append((time.perf_counter() - t0) * 1e3)
        k = L.ctt_hip_msm_last_timings(None, ms, 8)
        stages.append([round(float(ms[i]), 3) for i in range(k)])
    print(json.dumps({"log2n": lg, "chunks": chunks, "ms_per_call": round(sorted(ts)[1], 3),
                      "stage_ms_summed_over_slices[digits,sort,accumulate,merge,reduce,total]": stages[-1]}), flush=True)
L.ctt_hip_msm_set_option(None, b"chunks", 0)
