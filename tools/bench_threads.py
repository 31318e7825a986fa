#!/usr/bin/env python3
"""Concurrent callers of the Constantine host-pointer symbol (the reference's `_parallel` MSM is callable from several threads, each
with its own pool: include/constantine/core/threadpool.h:25-39, "can be nested" ec_multi_scalar_mul_parallel.nim:596):

    python tools/bench_threads.py [log2n=16] [calls=40]

T = 1, 2, 4, 8 threads, each issuing `calls` blocking calls on pageable host arrays; the library gives every calling thread a context of
its own up to $CTT_HIP_HOST_CONTEXTS (read per call).  One line per (contexts cap, threads): aggregate calls/s, speed-up over one thread,
and that every result equals the single-thread result."""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

name = "bls12_381_g1"
info = CURVES[name]
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = 1 << lg
eng = DeviceMsm(0)
d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
eng.gen_points(name, 5, n, d)
pts = d.cpu().numpy()
eng.close()
inputs = [synth_scalars(6 + t, n, 255) for t in range(8)]
want = [bytes(multiScalarMul_vartime_parallel(None, name, inputs[t], pts, coord="jac")) for t in range(8)]
base = None
for cap in (1, 2, 4, 8):
    os.environ["CTT_HIP_HOST_CONTEXTS"] = str(cap)
    for T in (1, 2, 4, 8):
        if T < cap and T != 1:
            continue
        ok = [True] * T

        def work(t):
            for _ in range(calls):
                r = multiScalarMul_vartime_parallel(None, name, inputs[t], pts, coord="jac")
                ok[t] = ok[t] and bytes(r) == want[t]
        for rep in range(2):     # (the first repetition opens the contexts and grows their workspaces)
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            dt = time.perf_counter() - t0
        rate = T * calls / dt
        if base is None:
            base = rate
        print(f"2^{lg} pairs per call, contexts cap {cap}, {T} thread(s): {rate:8.1f} calls/s aggregate ({dt / calls * 1e3:.3f} ms per call per thread), "
              f"{rate / base:.2f} x one thread, results {'ok' if all(ok) else 'DIFFER'}", flush=True)
