"""EIP-4844 blob_to_kzg_commitment known answers: a 4096-point BLS12-381 G1 MSM through the path
kzg_commit -> multiScalarMul_vartime (constantine/commitments/kzg.nim:186; vectors from
tests/protocol_ethereum_eip4844_deneb_kzg/blob_to_kzg_commitment/kzg-mainnet)."""
import os

import numpy as np
import pytest

from oracle import cref
from oracle import pyoracle as po
from tests import _golden

CURVE = po.BLS12_381_G1
NAME = "bls12_381_g1"


@pytest.fixture(scope="module")
def setup():
    pts = _golden.kzg4844_setup_points()
    assert all(CURVE.is_on_curve(P) for P in pts[:16])
    return pts, CURVE.points_to_array(pts)


def test_oracle_reproduces_commitments(setup):
    pts, pts_arr = setup
    cases = _golden.kzg4844_cases()
    assert len(cases) >= 5
    for name, scalars, commitment in cases:
        assert all(k < CURVE.order for k in scalars)
        sc = CURVE.scalars_to_array(scalars)
        out, _ = cref.msm(NAME, sc, pts_arr, nthreads=4)
        assert _golden.g1_compress(CURVE.aff_from_bytes(bytes(out))) == commitment, name
    # and once through the pure-Python bucket method, on the smallest non-trivial case
    name, scalars, commitment = min(cases, key=lambda c: sum(1 for k in c[1] if k))
    assert _golden.g1_compress(CURVE.msm_pippenger(scalars, pts, c=8)) == commitment


def test_emulated_kernels_reproduce_commitments(setup):
    from tests.emu import emu
    _, pts_arr = setup
    for name, scalars, commitment in _golden.kzg4844_cases()[:3]:
        sc = CURVE.scalars_to_array(scalars)
        out, _ = emu.msm(NAME, sc, pts_arr)
        assert _golden.g1_compress(CURVE.aff_from_bytes(bytes(out))) == commitment, name
        mont = CURVE.fr_scalars_to_array(scalars)     # kzg_commit hands Fr elements (fr_coefs entry)
        out, _ = emu.msm(NAME, mont, pts_arr, coef_is_fr=True)
        assert _golden.g1_compress(CURVE.aff_from_bytes(bytes(out))) == commitment, name


@pytest.mark.gpu
def test_gpu_reproduces_commitments(setup):
    from constantine_amd import CachedBases, multiScalarMul_vartime_parallel
    _, pts_arr = setup
    bases = CachedBases(NAME, pts_arr)      # the SRS is the textbook cached-base case
    try:
        for name, scalars, commitment in _golden.kzg4844_cases():
            mont = CURVE.fr_scalars_to_array(scalars)
            r = multiScalarMul_vartime_parallel(None, NAME, mont, pts_arr, coord="jac", fr_coefs=True)
            assert _golden.g1_compress(CURVE.jac_from_bytes(bytes(r))) == commitment, name
            r = bases.msm(CURVE.scalars_to_array(scalars), coord="prj")
            assert _golden.g1_compress(CURVE.prj_from_bytes(bytes(r))) == commitment, name
    finally:
        bases.close()


# ---- the protocol-level callers: the reference's C symbols (include/ctt_msm_hip.h part 3, csrc/protocols.hip) through
# ---- constantine_amd/kzg.py; the spec restatement they are checked against is oracle/kzg_spec.py ------------------------------
def test_spec_restatement_reproduces_the_reference_vectors():
    """oracle/kzg_spec.py (the checker of the C++ host logic) is itself pinned: y = p(z) of every valid compute_kzg_proof vector
    (incl. z = a root of unity and z = 0), the rejections decided before any MSM, and the codec on SRS points.  No GPU."""
    from oracle import kzg_spec as ks
    raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    for i in (0, 1, 77, 4095):
        c = raw[48 * i:48 * i + 48]
        pt = ks.deserialize_g1_compressed(c)
        assert pt == _golden.g1_decompress(c) and ks.serialize_g1_compressed(pt) == c
    cases = _golden.kzg4844_proof_cases()["compute_kzg_proof"]
    assert len(cases) == 52
    seen_root = False
    for case, blob, zb, res in cases:
        if res is None:
            with pytest.raises(ks.SpecError):
                ks.bytes_to_bls_field(zb)
                ks.blob_to_bigint_polynomial(blob)
                raise AssertionError(case + ": neither z nor the blob was rejected")
            continue
        z = ks.bytes_to_bls_field(zb)
        poly = [int.from_bytes(bytes(row), "little") for row in ks.blob_to_bigint_polynomial(blob)]
        q, y = ks.quotient_polynomial(poly, z)
        assert y.to_bytes(32, "big") == res[1], case
        dom = ks.domain_brp()
        seen_root |= z in dom
        for i in (0, 1, 777, 4095):   # q really is (p - y) / (X - z) on the domain
            assert q[i] * (dom[i] - z) % ks.R == (poly[i] - y) % ks.R or dom[i] == z, case
    assert seen_root, "the vectors include openings at a root of unity"


def test_host_logic_of_the_c_symbols():
    """The host-only pieces of the protocol symbols (C++, no GPU): SHA-256 against hashlib, the G1 codec, blob parsing with the
    reference's statuses, the Fiat-Shamir challenge and the quotient polynomial (both branches) against the spec restatement."""
    import hashlib
    import random
    from constantine_amd import kzg
    from oracle import kzg_spec as ks
    rng = random.Random(5)
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 1000, 131072 + 64):
        d = bytes(rng.randrange(256) for _ in range(n)) if n < 2000 else os.urandom(n)
        assert kzg.sha256(d) == hashlib.sha256(d).digest(), n
    raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    for i in list(range(0, 4096, 97)) + [4095]:
        c = raw[48 * i:48 * i + 48]
        aff = kzg.g1_decompress(c)
        assert aff == ks.aff_mont_bytes(_golden.g1_decompress(c)) and kzg.g1_compress(aff) == c
    assert kzg.g1_decompress(bytes([0xC0]) + bytes(47)) == bytes(96) and kzg.g1_compress(bytes(96)) == bytes([0xC0]) + bytes(47)
    for bad, want in ((bytes(48), "cttEthKzg_EccInvalidEncoding"),                                    # compression flag missing
                      (bytes([0xE0]) + bytes(47), "cttEthKzg_EccInvalidEncoding"),                   # infinity with the sign bit
                      (bytes([0xC0]) + bytes(46) + b"\x01", "cttEthKzg_EccInvalidEncoding"),         # infinity with payload
                      (bytes([0x9F]) + bytes([0xFF]) * 47, "cttEthKzg_EccCoordinateGreaterThanOrEqualModulus"),
                      (bytes([0x80]) + bytes(46) + b"\x02", "cttEthKzg_EccPointNotOnCurve")):         # x = 2: 12 is not a square
        with pytest.raises(kzg.KzgError) as e:
            kzg.g1_decompress(bad)
        assert e.value.status.name == want, bad.hex()
        with pytest.raises(ks.SpecError) as e2:
            ks.deserialize_g1_compressed(bad)
        assert e2.value.status == want
    n_bad = 0
    for name, blob, com in _golden.kzg4844_raw_cases():
        if com is None:
            n_bad += 1
            with pytest.raises(kzg.KzgError) as e:
                kzg.blob_to_bigint_polynomial(blob)
            want = (kzg.cttEthKzgStatus.cttEthKzg_InputsLengthsMismatch if len(blob) != kzg.BYTES_PER_BLOB
                    else kzg.cttEthKzgStatus.cttEthKzg_ScalarLargerThanCurveOrder)
            assert e.value.status == want, name
        else:
            poly = kzg.blob_to_bigint_polynomial(blob)
            assert bytes(poly) == bytes(ks.blob_to_bigint_polynomial(blob)), name
    assert n_bad == 4
    cases = _golden.kzg4844_proof_cases()
    for case, blob, com, res in cases["compute_blob_kzg_proof"]:
        if len(blob) == kzg.BYTES_PER_BLOB and len(com) == 48:
            assert int.from_bytes(kzg.compute_challenge(blob, com), "big") == ks.compute_challenge(blob, com), case
    checked = roots = 0
    for case, blob, zb, res in cases["compute_kzg_proof"]:
        if res is None:
            continue
        z = int.from_bytes(zb, "big")
        is_root = z in ks.domain_brp()
        if not is_root and checked >= 6:
            continue
        poly_le = kzg.blob_to_bigint_polynomial(blob)
        q, y = kzg.quotient_polynomial_host(poly_le, z)
        assert y.to_bytes(32, "big") == res[1], case
        want_q, want_y = ks.quotient_polynomial([int.from_bytes(bytes(r), "little") for r in poly_le], z)
        assert y == want_y and [int.from_bytes(bytes(r), "little") for r in q] == want_q, case
        checked += 1
        roots += is_root
    assert checked >= 7 and roots >= 1


@pytest.mark.gpu
def test_blob_to_kzg_commitment_on_gpu():
    """ctt_eth_kzg_blob_to_kzg_commitment (ethereum_eip4844_kzg.h:106) over the GPU MSM: all 11 reference vectors, with the SRS
    cached as plain records and as a window table."""
    from constantine_amd import kzg
    raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    for table in (False, True):
        ctx = kzg.EthereumKZGContext(raw, table=table)
        try:
            n = 0
            for name, blob, com in _golden.kzg4844_raw_cases():
                n += 1
                if com is None:
                    with pytest.raises(kzg.KzgError):
                        kzg.blob_to_kzg_commitment(ctx, blob)
                else:
                    assert kzg.blob_to_kzg_commitment(ctx, blob) == com, name
                    if table:   # the _parallel symbol (ethereum_eip4844_kzg_parallel.h:40): same path, thread pool unused
                        assert kzg.blob_to_kzg_commitment_parallel(None, ctx, blob) == com, name
            assert n == 11
        finally:
            ctx.delete()


@pytest.mark.gpu
def test_compute_kzg_proof_vectors_on_gpu():
    """ctt_eth_kzg_compute_kzg_proof / _compute_blob_kzg_proof (ethereum_eip4844_kzg.h:126,153): all 52 + 15 reference vectors."""
    from constantine_amd import kzg
    ctx = kzg.EthereumKZGContext(open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read())
    try:
        cases = _golden.kzg4844_proof_cases()
        assert len(cases["compute_kzg_proof"]) == 52 and len(cases["compute_blob_kzg_proof"]) == 15
        for case, blob, zb, res in cases["compute_kzg_proof"]:
            if res is None:
                with pytest.raises(kzg.KzgError):
                    kzg.compute_kzg_proof(ctx, blob, zb)
            else:
                assert kzg.compute_kzg_proof(ctx, blob, zb) == res, case
                assert kzg.compute_kzg_proof_parallel(None, ctx, blob, zb) == res, case
        for case, blob, com, res in cases["compute_blob_kzg_proof"]:
            if res is None:
                with pytest.raises(kzg.KzgError):
                    kzg.compute_blob_kzg_proof(ctx, blob, com)
            else:
                assert kzg.compute_blob_kzg_proof(ctx, blob, com) == res, case
                assert kzg.compute_blob_kzg_proof_parallel(None, ctx, blob, com) == res, case
    finally:
        ctx.delete()


@pytest.mark.gpu
def test_kzg_context_from_the_ceremony_text_file(tmp_path):
    """ctt_eth_kzg_context_new: the c-kzg text format ("4096\\n65\\n" + hex lines) written from the golden SRS; a missing file and
    a malformed one give the reference's trusted-setup statuses."""
    from constantine_amd import kzg
    raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    path = tmp_path / "setup.txt"
    # the G2 and monomial-G1 sections are not used by the commitment / proof functions, but the loader reads them like the
    # reference's (load_ckzg4844): length, hex digits, compression flag, coordinates below p.  The golden fixture holds the Lagrange
    # section only: the neutral element stands in for the 65 G2 points, the Lagrange lines for the 4096 monomial ones
    g2 = bytes([0xC0]) + bytes(95)
    g1_lines = [raw[48 * i:48 * i + 48].hex() for i in range(4096)]
    text = "4096\n65\n" + "\n".join(g1_lines) + "\n" + "\n".join([g2.hex()] * 65) + "\n" + "\n".join(g1_lines) + "\n"
    path.write_text(text)
    ctx = kzg.EthereumKZGContext.from_ckzg_text(path)
    try:
        name, blob, com = next(c for c in _golden.kzg4844_raw_cases() if c[2] is not None)
        assert kzg.blob_to_kzg_commitment(ctx, blob) == com
    finally:
        ctx.delete()
    ctx = kzg.EthereumKZGContext.from_ckzg_text(path, precompute=(256, 8))     # ctt_eth_kzg_context_new_with_precompute
    try:
        assert kzg.blob_to_kzg_commitment(ctx, blob) == com
    finally:
        ctx.delete()
    with pytest.raises(ValueError, match="cttEthTS_MissingOrInaccessibleFile"):
        kzg.EthereumKZGContext.from_ckzg_text(tmp_path / "nope.txt")
    # what load_ckzg4844 rejects: a character that is not a hex digit (sscanf's "%2x" took "+f" in round 4), a short or a long line, a
    # file that ends before its G2 or monomial section, counts other than 4096 / 65; CRLF line ends are fine
    bad = tmp_path / "bad.txt"
    cases = {
        "non-hex": text.replace(g1_lines[7], "+f" + g1_lines[7][2:], 1),
        "short line": text.replace(g1_lines[9], g1_lines[9][:-2], 1),
        "long line": text.replace(g1_lines[9], g1_lines[9] + "00", 1),
        "no monomial section": "4096\n65\n" + "\n".join(g1_lines) + "\n" + "\n".join([g2.hex()] * 65) + "\n",
        "truncated G2 section": "4096\n65\n" + "\n".join(g1_lines) + "\n" + "\n".join([g2.hex()] * 10) + "\n",
        "G2 without the compression flag": text.replace(g2.hex(), "00" * 96, 1),
        "wrong count": text.replace("4096\n65\n", "4095\n65\n", 1),
        "zz": "4096\n65\n" + raw[:48].hex() + "\nzz\n",
    }
    for what, body in cases.items():
        bad.write_text(body)
        with pytest.raises(ValueError, match="cttEthTS_InvalidFile"):
            kzg.EthereumKZGContext.from_ckzg_text(bad)
            raise AssertionError(what + " was accepted")
    crlf = tmp_path / "crlf.txt"
    crlf.write_bytes(text.replace("\n", "\r\n").encode())
    ctx = kzg.EthereumKZGContext.from_ckzg_text(crlf)
    try:
        assert kzg.blob_to_kzg_commitment(ctx, blob) == com
    finally:
        ctx.delete()


@pytest.mark.gpu
def test_two_kzg_contexts_on_two_threads():
    """Two contexts, each with its own streams and lock, used from two threads at once (no process-wide lock on the path):
    every commitment and proof equals the single-threaded answer."""
    import threading
    from constantine_amd import kzg
    raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    blobs = [(blob, com) for _, blob, com in _golden.kzg4844_raw_cases() if com is not None][:4]
    ctxs = [kzg.EthereumKZGContext(raw), kzg.EthereumKZGContext(raw, table=False)]
    errors = []

    def work(ctx):
        try:
            for _ in range(6):
                for blob, com in blobs:
                    assert kzg.blob_to_kzg_commitment(ctx, blob) == com
                    proof = kzg.compute_blob_kzg_proof(ctx, blob, com)
                    assert len(proof) == 48
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    try:
        ths = [threading.Thread(target=work, args=(c,)) for c in ctxs for _ in range(2)]   # two threads per context, two contexts
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errors, errors
    finally:
        for c in ctxs:
            c.delete()


def test_quotient_polynomial_bodies_match_the_host_formula():
    """ctt_hip_fr_quotient's per-lane bodies (msm_bodies.h fr_quotient_*_body, run here by tests/emu) against the spec formula
    quotient_polynomial: barycentric y = p(z) and q_i = (p_i - y)/(w_i - z) over the bit-reversed 4096-point domain, for several
    lane spans (one Montgomery-trick run and one inversion per lane)."""
    import random

    from oracle import kzg_spec as ks
    from tests.emu import emu
    r = ks.R
    n = ks.FIELD_ELEMENTS_PER_BLOB
    rng = random.Random(44)
    dom = ks.domain_brp()
    R = ks.FR_MONT
    dom_m = np.frombuffer(b"".join((w * R % r).to_bytes(32, "little") for w in dom), dtype=np.uint8).reshape(n, 32)
    for K in (8, 5, 64):
        poly = [rng.randrange(r) for _ in range(n)]
        poly[3] = 0
        poly[7] = r - 1
        z = rng.randrange(r)
        assert pow(z, n, r) != 1
        want_q, want_y = ks.quotient_polynomial(poly, z)
        poly_le = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in poly), dtype=np.uint8).reshape(n, 32)
        scale = (pow(z, n, r) - 1) * pow(n, -1, r) % r
        q, y = emu.fr_quotient("bls12_381_g1", poly_le, dom_m, np.frombuffer((z * R % r).to_bytes(32, "little"), dtype=np.uint8),
                               np.frombuffer((scale * R % r).to_bytes(32, "little"), dtype=np.uint8), K=K)
        assert int.from_bytes(bytes(y), "little") == want_y
        assert [int.from_bytes(bytes(row), "little") for row in q] == want_q


@pytest.mark.gpu
def test_quotient_polynomial_on_device_matches_the_host_formula():
    """The device path of the proofs (ctt_hip_fr_quotient) against the spec formula, and the branch it leaves to the host
    (z a root of unity: -2)."""
    import ctypes
    import random

    import torch
    from constantine_amd import _lib
    from oracle import kzg_spec as ks
    L = _lib.lib()
    r = ks.R
    n = ks.FIELD_ELEMENTS_PER_BLOB
    rng = random.Random(45)
    dom = ks.domain_brp()
    d_dom = torch.from_numpy(np.frombuffer(b"".join((w * ks.FR_MONT % r).to_bytes(32, "little") for w in dom), dtype=np.uint8).reshape(n, 32).copy()).cuda()
    d_q = torch.empty((n, 32), dtype=torch.uint8, device="cuda")

    def device_quotient(poly_le, z):
        d_poly = torch.from_numpy(np.ascontiguousarray(poly_le)).cuda()
        torch.cuda.synchronize()
        y = np.zeros(32, dtype=np.uint8)
        zb = np.frombuffer(z.to_bytes(32, "little"), dtype=np.uint8).copy()
        rc = L.ctt_hip_fr_quotient(None, 0, ctypes.c_void_p(d_q.data_ptr()), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_poly.data_ptr()),
                                   ctypes.c_void_p(d_dom.data_ptr()), zb.ctypes.data_as(ctypes.c_void_p), n)
        return rc, int.from_bytes(bytes(y), "little")

    for _ in range(3):
        poly = [rng.randrange(r) for _ in range(n)]
        z = rng.randrange(r)
        want_q, want_y = ks.quotient_polynomial(poly, z)
        poly_le = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in poly), dtype=np.uint8).reshape(n, 32)
        rc, y = device_quotient(poly_le, z)
        assert rc == 0 and y == want_y
        assert [int.from_bytes(bytes(row), "little") for row in d_q.cpu().numpy()] == want_q
    assert device_quotient(poly_le, dom[5])[0] == -2
