"""Loaders for tests/golden fixtures (made by tests/golden/make_golden.py from the reference's vectors)."""
import json
import os

from oracle import pyoracle as po

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _c(curve, v):
    if isinstance(v, list):
        return (int(v[0], 16), int(v[1], 16))
    return int(v, 16)


def scalar_mul_kats(name):
    doc = json.load(open(os.path.join(HERE, "scalar_mul_kat.json")))
    curve = po.CURVES[name]
    out = []
    for P, k, Q in doc[name]:
        out.append(((_c(curve, P[0]), _c(curve, P[1])), int(k, 16), (_c(curve, Q[0]), _c(curve, Q[1]))))
    return out


def eip2537(group):
    """-> list of (name, scalars, points, expected_affine_or_None). EIP-2537 wire format:
    G1 pair = 64B x | 64B y | 32B scalar (big-endian); G2 pair = x.c0|x.c1|y.c0|y.c1 (64B each) | 32B scalar
    (reference: constantine/ethereum_evm_precompiles.nim:905-975)."""
    doc = json.load(open(os.path.join(HERE, "eip2537_multiexp.json")))
    ncoord = 2 if group == "g1" else 4
    plen = 64 * ncoord + 32
    out = []
    for name, inp, exp in doc[group]:
        raw = bytes.fromhex(inp)
        assert len(raw) % plen == 0
        scalars, points = [], []
        for i in range(len(raw) // plen):
            rec = raw[i * plen:(i + 1) * plen]
            cs = [int.from_bytes(rec[64 * j:64 * (j + 1)], "big") for j in range(ncoord)]
            k = int.from_bytes(rec[64 * ncoord:], "big")
            if group == "g1":
                P = None if cs == [0, 0] else (cs[0], cs[1])
            else:
                P = None if cs == [0, 0, 0, 0] else ((cs[0], cs[1]), (cs[2], cs[3]))
            scalars.append(k)
            points.append(P)
        e = bytes.fromhex(exp)
        es = [int.from_bytes(e[64 * j:64 * (j + 1)], "big") for j in range(ncoord)]
        if all(v == 0 for v in es):
            E = None
        elif group == "g1":
            E = (es[0], es[1])
        else:
            E = ((es[0], es[1]), (es[2], es[3]))
        out.append((name, scalars, points, E))
    return out
