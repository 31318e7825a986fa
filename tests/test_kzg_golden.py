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


# ---- the protocol-level caller (constantine_amd/kzg.py) -----------------------------------------------------
def test_kzg_codec_and_validation_host_logic():
    """Host logic of the KZG layer needs no GPU: point codec round trip, blob parsing, rejection of bad blobs."""
    import os
    from constantine_amd import kzg
    raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    for i in (0, 1, 77, 4095):
        c = raw[48 * i:48 * i + 48]
        P = kzg.deserialize_g1_compressed(c)
        assert P == _golden.g1_decompress(c) and kzg.serialize_g1_compressed(P) == c
    assert kzg.deserialize_g1_compressed(bytes([0xC0]) + bytes(47)) is None
    with pytest.raises(kzg.KzgError):
        kzg.deserialize_g1_compressed(bytes(48))            # compression flag missing
    with pytest.raises(kzg.KzgError):
        kzg.deserialize_g1_compressed(bytes([0x9F]) + bytes([0xFF]) * 47)   # x >= p
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
            assert poly.shape == (4096, 32)
            assert int.from_bytes(bytes(poly[5]), "little") == int.from_bytes(blob[160:192], "big")
    assert n_bad == 4


@pytest.mark.gpu
def test_blob_to_kzg_commitment_on_gpu():
    """blob_to_kzg_commitment (ethereum_eip4844_kzg.nim:297-330) over the GPU MSM: all reference vectors."""
    import os
    from constantine_amd import kzg
    raw = open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    ctx = kzg.EthereumKZGContext(raw)
    try:
        for name, blob, com in _golden.kzg4844_raw_cases():
            if com is None:
                with pytest.raises(kzg.KzgError):
                    kzg.blob_to_kzg_commitment(ctx, blob)
            else:
                assert kzg.blob_to_kzg_commitment(ctx, blob) == com, name
    finally:
        ctx.delete()


# ---- proofs (compute_kzg_proof / compute_blob_kzg_proof) ---------------------------------------------------------
def test_quotient_polynomial_evaluations_match_the_reference_vectors():
    """Host part of kzg_prove: y = p(z) of every valid compute_kzg_proof vector (incl. z = a root of unity and z = 0),
    and the rejections that are decided before any MSM (z >= r, malformed blob).  No GPU."""
    from constantine_amd import kzg
    cases = _golden.kzg4844_proof_cases()["compute_kzg_proof"]
    seen_root = False
    for case, blob, zb, res in cases:
        if res is None:
            with pytest.raises(kzg.KzgError):
                z = kzg._bytes_to_bls_field(zb)
                kzg.blob_to_bigint_polynomial(blob)
                raise AssertionError(case + ": neither z nor the blob was rejected")
            continue
        z = kzg._bytes_to_bls_field(zb)
        poly = [int.from_bytes(bytes(row), "little") for row in kzg.blob_to_bigint_polynomial(blob)]
        q, y = kzg.quotient_polynomial(poly, z)
        assert y.to_bytes(32, "big") == res[1], case
        seen_root |= z in kzg._domain_brp()
        # q really is (p - y) / (X - z) on the domain
        dom = kzg._domain_brp()
        for i in (0, 1, 777, 4095):
            assert q[i] * (dom[i] - z) % kzg._R == (poly[i] - y) % kzg._R or dom[i] == z, case
    assert seen_root, "the vectors include openings at a root of unity"


@pytest.mark.gpu
def test_compute_kzg_proof_vectors_on_gpu():
    from constantine_amd import kzg
    ctx = kzg.EthereumKZGContext(open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read())
    try:
        cases = _golden.kzg4844_proof_cases()
        for case, blob, zb, res in cases["compute_kzg_proof"]:
            if res is None:
                with pytest.raises(kzg.KzgError):
                    kzg.compute_kzg_proof(ctx, blob, zb)
            else:
                assert kzg.compute_kzg_proof(ctx, blob, zb) == res, case
        for case, blob, com, res in cases["compute_blob_kzg_proof"]:
            if res is None:
                with pytest.raises(kzg.KzgError):
                    kzg.compute_blob_kzg_proof(ctx, blob, com)
            else:
                assert kzg.compute_blob_kzg_proof(ctx, blob, com) == res, case
    finally:
        ctx.delete()


def test_quotient_polynomial_bodies_match_the_host_formula():
    """ctt_hip_fr_quotient's per-lane bodies (msm_bodies.h fr_quotient_*_body, run here by tests/emu) against the host formula
    quotient_polynomial: barycentric y = p(z) and q_i = (p_i - y)/(w_i - z) over the bit-reversed 4096-point domain, for several
    lane spans (one Montgomery-trick run and one inversion per lane)."""
    import random

    from constantine_amd import kzg
    from tests.emu import emu
    r = kzg._R
    n = kzg.FIELD_ELEMENTS_PER_BLOB
    rng = random.Random(44)
    dom = kzg._domain_brp()
    R = 1 << 256
    dom_m = np.frombuffer(b"".join((w * R % r).to_bytes(32, "little") for w in dom), dtype=np.uint8).reshape(n, 32)
    for K in (8, 5, 64):
        poly = [rng.randrange(r) for _ in range(n)]
        poly[3] = 0
        poly[7] = r - 1
        z = rng.randrange(r)
        assert pow(z, n, r) != 1
        want_q, want_y = kzg.quotient_polynomial(poly, z)
        poly_le = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in poly), dtype=np.uint8).reshape(n, 32)
        scale = (pow(z, n, r) - 1) * pow(n, -1, r) % r
        q, y = emu.fr_quotient("bls12_381_g1", poly_le, dom_m, np.frombuffer((z * R % r).to_bytes(32, "little"), dtype=np.uint8),
                               np.frombuffer((scale * R % r).to_bytes(32, "little"), dtype=np.uint8), K=K)
        assert int.from_bytes(bytes(y), "little") == want_y
        assert [int.from_bytes(bytes(row), "little") for row in q] == want_q


@pytest.mark.gpu
def test_quotient_polynomial_on_device_matches_the_host_formula():
    """The device path of the proofs (ctt_hip_fr_quotient) against quotient_polynomial, and the branch it leaves to the host
    (z a root of unity: -2)."""
    import random

    import os

    from constantine_amd import kzg
    ctx = kzg.EthereumKZGContext(open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read())
    r = kzg._R
    n = kzg.FIELD_ELEMENTS_PER_BLOB
    rng = random.Random(45)
    for _ in range(3):
        poly = [rng.randrange(r) for _ in range(n)]
        z = rng.randrange(r)
        want_q, want_y = kzg.quotient_polynomial(poly, z)
        poly_le = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in poly), dtype=np.uint8).reshape(n, 32)
        d_q, y = kzg.quotient_polynomial_device(ctx, poly_le, z)
        assert y == want_y
        assert [int.from_bytes(bytes(row), "little") for row in d_q.cpu().numpy()] == want_q
    assert kzg.quotient_polynomial_device(ctx, poly_le, kzg._domain_brp()[5]) is None
    ctx.delete()
