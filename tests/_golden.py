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


# ---- EIP-4844 blob -> KZG commitment (N = 4096 G1 MSM) ------------------------------------------------------
_BLS_P = po.BLS12_381_G1.F.p


def g1_decompress(b48: bytes):
    """ZCash BLS12-381 G1 encoding: bit 7 compression flag, bit 6 infinity, bit 5 = y is the larger root."""
    assert len(b48) == 48 and b48[0] & 0x80
    if b48[0] & 0x40:
        return None
    x = int.from_bytes(b48, "big") & ((1 << 381) - 1)
    y2 = (pow(x, 3, _BLS_P) + 4) % _BLS_P
    y = pow(y2, (_BLS_P + 1) // 4, _BLS_P)
    assert y * y % _BLS_P == y2, "not on the curve"
    if bool(b48[0] & 0x20) != (y > (_BLS_P - 1) // 2):
        y = _BLS_P - y
    return (x, y)


def g1_compress(P) -> bytes:
    if P is None:
        return bytes([0xC0]) + bytes(47)
    x, y = P
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if y > (_BLS_P - 1) // 2 else 0)
    return bytes(b)


def _bit_reverse_permutation(seq):
    n = len(seq)
    bits = n.bit_length() - 1
    return [seq[int(format(i, f"0{bits}b")[::-1], 2)] for i in range(n)]


def kzg4844_setup_points():
    """The 4096 Lagrange-form G1 points in the order commitments use them: bit-reversal permutation of the file
    order (EIP-4844 `bit_reversal_permutation(KZG_SETUP_G1_LAGRANGE)`; the reference applies it when it loads the
    setup, constantine/commitments_setups/ethereum_kzg_srs.nim)."""
    raw = open(os.path.join(HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    pts = [g1_decompress(raw[48 * i:48 * i + 48]) for i in range(4096)]
    return _bit_reverse_permutation(pts)


def kzg4844_cases():
    """-> [(name, [4096 scalars], commitment_bytes)]; blob = 4096 big-endian 32-byte field elements."""
    import base64
    import zlib
    out = []
    for name, blob, com in kzg4844_raw_cases():
        if com is None:
            continue
        assert len(blob) == 4096 * 32
        scalars = [int.from_bytes(blob[32 * i:32 * i + 32], "big") for i in range(4096)]
        out.append((name, scalars, com))
    return out


def kzg4844_raw_cases():
    """-> [(name, blob_bytes, commitment_bytes or None)]; None = the reference rejects the blob."""
    import base64
    import zlib
    out = []
    for name, z, com in json.load(open(os.path.join(HERE, "kzg4844_blob_to_commitment.json"))):
        if z.startswith("LEN:"):
            blob = bytes(int(z[4:]))          # wrong-length blob; its content never matters
        else:
            blob = zlib.decompress(base64.b64decode(z))
        out.append((name, blob, None if com is None else bytes.fromhex(com)))
    return out


def kzg4844_proof_cases():
    """-> {"compute_kzg_proof": [(case, blob_bytes, z_bytes, (proof, y) | None)],
           "compute_blob_kzg_proof": [(case, blob_bytes, commitment_bytes, proof | None)]}"""
    blobs = {name: blob for name, blob, _ in kzg4844_raw_cases()}
    doc = json.load(open(os.path.join(HERE, "kzg4844_proofs.json")))
    out = {"compute_kzg_proof": [], "compute_blob_kzg_proof": []}
    for case, ref, z, res in doc["compute_kzg_proof"]:
        out["compute_kzg_proof"].append((case, blobs[ref], bytes.fromhex(z),
                                         None if res is None else (bytes.fromhex(res[0]), bytes.fromhex(res[1]))))
    for case, ref, com, res in doc["compute_blob_kzg_proof"]:
        out["compute_blob_kzg_proof"].append((case, blobs[ref], bytes.fromhex(com), None if res is None else bytes.fromhex(res)))
    return out
