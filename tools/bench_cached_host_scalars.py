#!/usr/bin/env python3
"""Cached bases with HOST-resident coefficients (ctt_hip_msm_with_bases, coefs_on_device = 0 -- the Halo2-ZAL msm_with_cached_base shape):
ms per blocking call with the coefficients uploaded in one piece in front of the MSM (option chunks = 1: what rounds 2-3 did) and in slices
underneath the accumulation (chunks = 0, automatic: MsmEngine::submit_host with d_prepared, round 4); plain records and window table."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from constantine_amd import CURVES, CachedBases, DeviceMsm, _lib  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402


def med(fn, n=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), r


L = _lib.lib()
for name, lg, fr in (("bn254_snarks_g1", 22, True), ("bls12_381_g1", 20, False), ("bls12_381_g1", 22, False), ("bls12_381_g1", 16, False)):
    info = CURVES[name]
    n = 1 << lg
    eng = DeviceMsm(0)
    d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(name, 5, n, d)
    sc = synth_scalars(6, n, 253 if fr else info.scalar_bits)
    d_sc = torch.from_numpy(sc).cuda()
    for table in (False, True):
        bases = CachedBases(name, d, ctx=eng.ctx, on_device=True, table=table)
        dev_ms, r_dev = med(lambda: bases.msm(d_sc, coord="aff", fr_coefs=fr))
        out = {}
        for chunks in (1, 0, 2, 3, 4, 6):
            L.ctt_hip_msm_set_option(eng.ctx, b"chunks", chunks)
            ms, r = med(lambda: bases.msm(sc, coord="aff", fr_coefs=fr))
            assert bytes(r) == bytes(r_dev)
            out[chunks] = ms
        L.ctt_hip_msm_set_option(eng.ctx, b"chunks", 0)
        print(f"{name} 2^{lg} {'window table' if table else 'plain records'}: coefficients on the device {dev_ms:.3f} ms; on the host, "
              f"one piece {out[1]:.3f} ms, automatic slices {out[0]:.3f} ms; forced 2 / 3 / 4 / 6 slices {out[2]:.3f} / {out[3]:.3f} / {out[4]:.3f} / {out[6]:.3f}", flush=True)
        bases.close()
    eng.close()
