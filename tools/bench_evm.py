#!/usr/bin/env python3
"""Latency of the EIP-2537 symbols ctt_eth_evm_bls12381_g1msm / _g2msm through the C ABI for k pairs (k = 1 .. 512): parsing and
field / curve checks on the host, subgroup checks (host up to 256 points, one GPU launch above), the MSM through the host-pointer
entry.  Inputs: the pairs of the reference's vectors, repeated.  Median of 20 calls after 3 warm-ups."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from constantine_amd import evm  # noqa: E402
from tests import _golden  # noqa: E402

DOC = json.load(open(os.path.join(_golden.HERE, "eip2537_multiexp.json")))


def med(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


for group, rec, fn in (("g1", 160, evm.eth_evm_bls12381_g1msm), ("g2", 288, evm.eth_evm_bls12381_g2msm)):
    pool = b"".join(bytes.fromhex(inp) for _, inp, _ in DOC[group])
    pairs = [pool[i:i + rec] for i in range(0, len(pool), rec)]
    pairs = [p for p in pairs if any(p[:rec - 32])]            # (skip the neutral: it passes every check at once)
    for k in (1, 4, 16, 64, 128, 256, 512):
        inp = b"".join(pairs[i % len(pairs)] for i in range(k))
        m = med(lambda: fn(inp))
        print(f"ctt_eth_evm_bls12381_{group}msm, {k} pairs: {m[0]:.3f} ms (min {m[1]:.3f})", flush=True)
