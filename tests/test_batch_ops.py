"""
Batch conversion to affine and sum-reduce: batchAffine(_vartime) and sum_reduce_vartime
(ec_shortweierstrass_batch_ops.nim:44-345, :649-663).  CPU: the kernel bodies through tests/emu; GPU: the C ABI.
Affine coordinates are unique, so every comparison is byte-exact against the oracle.
"""
import random

import numpy as np
import pytest

from oracle import cref
from oracle import pyoracle as po
from tests.emu import emu

ALL = list(po.CURVES)


def _rand_fe(F, rng, allow_zero=False):
    base = F if F.degree == 1 else F.base
    while True:
        v = rng.randrange(base.p) if F.degree == 1 else (rng.randrange(base.p), rng.randrange(base.p))
        if allow_zero or not F.is_zero(v):
            return v


def make_nonaffine(name, aff_bytes_rows, kind, rng, neutral_at=()):
    """Blow affine points up to Jacobian (kind 'jac': X = x z^2, Y = y z^3) or projective ('prj': X = x z, Y = y z)
    coordinates with random z; rows listed in neutral_at (and affine neutrals) become Z = 0 with junk X, Y."""
    curve = po.CURVES[name]
    F = curve.F
    rows, expect = [], []
    for i, b in enumerate(aff_bytes_rows):
        P = curve.aff_from_bytes(bytes(b))
        if i in neutral_at or P is None:
            x, y, z = _rand_fe(F, rng), _rand_fe(F, rng), F.from_int(0)
            P = None
        else:
            z = _rand_fe(F, rng)
            if kind == "jac":
                z2 = F.sqr(z)
                x, y = F.mul(P[0], z2), F.mul(P[1], F.mul(z2, z))
            else:
                x, y = F.mul(P[0], z), F.mul(P[1], z)
        rows.append(F.to_mont_bytes(x) + F.to_mont_bytes(y) + F.to_mont_bytes(z))
        expect.append(curve.aff_to_bytes(P))
    n = len(rows)
    return (np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(n, -1).copy(),
            np.frombuffer(b"".join(expect), dtype=np.uint8).reshape(n, -1).copy())


def oracle_sum(name, pts):
    curve = po.CURVES[name]
    acc = None
    for b in pts:
        acc = curve.add(acc, curve.aff_from_bytes(bytes(b)))
    return acc


# ---- CPU: kernel bodies through the emulator ----------------------------------------------------------------
@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("kind", ["jac", "prj"])
def test_emu_batch_affine(name, kind):
    rng = random.Random(11)
    n = 37
    aff = cref.gen_points(name, 21, n)
    src, expect = make_nonaffine(name, aff, kind, rng, neutral_at={0, 5, 6, n - 1})
    for K in (1, 8, 64):
        got = emu.batch_affine(name, src, 1 if kind == "jac" else 2, K=K)
        assert bytes(got) == bytes(expect), (name, kind, K)
    # all-neutral lane, single point
    src1, exp1 = make_nonaffine(name, aff[:3], kind, rng, neutral_at={0, 1, 2})
    assert bytes(emu.batch_affine(name, src1, 1 if kind == "jac" else 2, K=8)) == bytes(exp1)


@pytest.mark.parametrize("name", ALL)
def test_emu_sum_reduce(name):
    curve = po.CURVES[name]
    pts = cref.gen_points(name, 33, 300)
    for n, K in ((1, 0), (2, 0), (15, 4), (64, 16), (300, 7), (300, 0)):
        got = emu.sum_reduce(name, pts[:n], K=K)
        assert curve.aff_from_bytes(bytes(got)) == oracle_sum(name, pts[:n]), (name, n, K)
    # P + P (doubling inside a lane and across lanes), P + (-P), neutrals in the list, empty input
    P = curve.aff_from_bytes(bytes(pts[0]))
    mix = [P, P, P, None, curve.neg(P), P, None, P, curve.neg(P)]
    arr = curve.points_to_array(mix)
    assert curve.aff_from_bytes(bytes(emu.sum_reduce(name, arr, K=2))) == curve.scalar_mul(3, P)
    assert curve.jac_from_bytes(bytes(emu.sum_reduce(name, arr, out_kind=1, K=4))) == curve.scalar_mul(3, P)
    assert curve.aff_from_bytes(bytes(emu.sum_reduce(name, arr[:0]))) is None


# ---- GPU: through the C ABI -----------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("kind", ["jac", "prj"])
def test_gpu_batch_affine_constantine_symbol(name, kind):
    from constantine_amd import batchAffine_vartime
    rng = random.Random(12)
    n = 301
    aff = cref.gen_points(name, 22, n)
    src, expect = make_nonaffine(name, aff, kind, rng, neutral_at={0, 7, 8, 9, 150, n - 1})
    assert bytes(batchAffine_vartime(name, src, coord=kind)) == bytes(expect)
    assert batchAffine_vartime(name, src[:0], coord=kind).shape[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL)
def test_gpu_sum_reduce_vs_oracle(name):
    from constantine_amd import sum_reduce_vartime
    curve = po.CURVES[name]
    pts = cref.gen_points(name, 34, 1000)
    for n in (1, 2, 17, 1000):
        got = sum_reduce_vartime(name, pts[:n], coord="aff")
        assert curve.aff_from_bytes(bytes(got)) == oracle_sum(name, pts[:n]), (name, n)
    P = curve.aff_from_bytes(bytes(pts[0]))
    arr = curve.points_to_array([P, P, None, curve.neg(P), P, P] * 50)
    assert curve.jac_from_bytes(bytes(sum_reduce_vartime(name, arr, coord="jac"))) == curve.scalar_mul(150, P)
    assert curve.prj_from_bytes(bytes(sum_reduce_vartime(name, arr[:0], coord="prj"))) is None


@pytest.mark.gpu
def test_gpu_device_resident_primitives_at_scale():
    """2^20 points in HBM: sum_reduce against the ORACLE (an MSM of the port with every scalar 1, which the engine's own MSM must
    equal too); batch_affine of 2^16 Jacobian and projective points with random Z (built on the host with the big-integer field,
    neutrals sprinkled in) against the affine points they were made from, and the Z = 1 / Z = 0 rows at 2^20."""
    import os
    import torch
    from constantine_amd import DeviceMsm, CURVES
    name = "bls12_381_g1"
    info = CURVES[name]
    n = 1 << 20
    nt = max(1, min(32, os.cpu_count() or 1))
    eng = DeviceMsm(0)
    d_pts = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(name, 77, n, d_pts)
    pts = d_pts.cpu().numpy()
    ones = np.zeros((n, 32), dtype=np.uint8)
    ones[:, 0] = 1
    want = bytes(cref.msm(name, ones, pts, nthreads=nt)[0])          # the oracle's sum of the 2^20 points
    assert bytes(eng.sum_reduce(name, d_pts, n, coord="aff")) == want
    assert bytes(eng.msm(name, torch.from_numpy(ones).cuda(), d_pts, n, coord="aff")) == want
    # random Z: 2^16 points blown up on the host, converted back on the device
    m = 1 << 16
    rng = random.Random(99)
    for kind in ("jac", "prj"):
        src, expect = make_nonaffine(name, pts[:m], kind, rng, neutral_at=set(range(0, m, 997)) | {m - 1})
        d_src = torch.from_numpy(src).cuda()
        d_out = torch.empty((m, info.aff_bytes), dtype=torch.uint8, device="cuda")
        eng.batch_affine(name, d_out, d_src, m, src_coord=kind)
        torch.cuda.synchronize()
        assert bytes(d_out.cpu().numpy()) == bytes(expect), kind
        # the Constantine symbol on HOST arrays large enough for its slices (4 from 2^19 points, 8 from 2^21: uploads on a helper thread,
        # kernel and download per slice -- ctt_hip_batch_affine): the same rows repeated, ragged lengths
        from constantine_amd import batchAffine_vartime
        for big in ((1 << 19) + 333, (1 << 21) + 5) if kind == "jac" else ((1 << 19) + 64,):
            reps = big // m + 1
            got = batchAffine_vartime(name, np.tile(src, (reps, 1))[:big], coord=kind)
            assert bytes(got) == bytes(np.tile(expect, (reps, 1))[:big]), (kind, big)
    # Jacobian with Z = 1 (Montgomery one) must come back unchanged; Z = 0 rows become the neutral
    one = np.frombuffer(po.CURVES[name].F.to_mont_bytes(1), dtype=np.uint8)
    d_jac = torch.empty((n, 3 * info.coord_bytes), dtype=torch.uint8, device="cuda")
    d_jac[:, :info.aff_bytes] = d_pts
    d_jac[:, info.aff_bytes:] = torch.from_numpy(one.copy()).cuda()
    d_jac[::1000, info.aff_bytes:] = 0
    d_out = torch.empty_like(d_pts)
    eng.batch_affine(name, d_out, d_jac, n, src_coord="jac")
    torch.cuda.synchronize()
    exp = d_pts.clone()
    exp[::1000] = 0
    assert torch.equal(d_out, exp)
    eng.close()
