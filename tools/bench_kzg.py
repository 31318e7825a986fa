#!/usr/bin/env python3
"""Latency of the KZG symbols (ctt_eth_kzg_blob_to_kzg_commitment / _compute_kzg_proof / _compute_blob_kzg_proof) through the C ABI,
SRS cached as plain records and as a window table; median of 30 calls after 5 warm-ups, host pointers in and out."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from constantine_amd import kzg  # noqa: E402
from tests import _golden  # noqa: E402

raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
cases = [c for c in _golden.kzg4844_raw_cases() if c[2] is not None]
blob, com = max(cases, key=lambda c: len(set(c[1])))[1:]          # a random blob (dense scalars)
z = bytes(31) + b"\x05"


def med(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


for table in (False, True):
    t0 = time.perf_counter()
    ctx = kzg.EthereumKZGContext(raw, table=table)
    build = (time.perf_counter() - t0) * 1e3
    assert kzg.blob_to_kzg_commitment(ctx, blob) == com
    a = med(lambda: kzg.blob_to_kzg_commitment(ctx, blob))
    b = med(lambda: kzg.compute_kzg_proof(ctx, blob, z))
    c = med(lambda: kzg.compute_blob_kzg_proof(ctx, blob, com))
    print(f"SRS as {'window table' if table else 'plain records'}: context {build:.0f} ms (4096 square roots on the host + upload); "
          f"blob_to_kzg_commitment {a[0]:.3f} ms (min {a[1]:.3f}), compute_kzg_proof {b[0]:.3f} ms (min {b[1]:.3f}), "
          f"compute_blob_kzg_proof {c[0]:.3f} ms (min {c[1]:.3f})", flush=True)
    ctx.delete()
