"""C++ CPU restatement (oracle/msm_ref.cpp) vs the golden-pinned Python oracle."""
import numpy as np
import pytest

from oracle import cref
from oracle import pyoracle as po
from tests import _golden

ALL = list(po.CURVES)


def _aff(curve, b):
    return curve.aff_from_bytes(bytes(b))


@pytest.mark.parametrize("name", ALL)
def test_scalar_mul_kats(name):
    curve = po.CURVES[name]
    kats = _golden.scalar_mul_kats(name)
    for P, k, Q in kats[-40:] + kats[:8]:
        out = cref.scalar_mul(name, curve.scalars_to_array([k])[0], curve.points_to_array([P])[0])
        assert _aff(curve, out) == Q


@pytest.mark.parametrize("name", ALL)
def test_gen_points_matches_python_definition(name):
    curve = po.CURVES[name]
    pts = cref.gen_points(name, 1234, 6, first=3)
    for i in range(6 if curve.F.degree == 1 else 2):
        assert _aff(curve, pts[i]) == po.synth_point(curve, 1234, 3 + i)


def test_synth_scalars_vectorised():
    a = cref.synth_scalars(99, 17, 255, first=5)
    for i in range(17):
        assert int.from_bytes(bytes(a[i]), "little") == po.synth_scalar(99, 5 + i, 255)
    a = cref.synth_scalars(99, 4, 254)
    assert all(int.from_bytes(bytes(r), "little") < (1 << 254) for r in a)


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("n", [1, 2, 3, 8, 33])
def test_msm_small_vs_python(name, n):
    curve = po.CURVES[name]
    if curve.F.degree == 2 and n > 8:
        pytest.skip("python Fp2 too slow")
    pts = cref.gen_points(name, 7, n)
    sc = cref.synth_scalars(11, n, curve.scalar_bits)
    ks = [int.from_bytes(bytes(r), "little") for r in sc]
    Ps = [_aff(curve, p) for p in pts]
    expect = curve.msm_naive(ks, Ps)
    for nthreads, c in [(1, 0), (1, 4), (3, 0), (2, 5)]:
        out, _ = cref.msm(name, sc, pts, nthreads=nthreads, c=c)
        assert _aff(curve, out) == expect


@pytest.mark.parametrize("group,cname", [("g1", "bls12_381_g1"), ("g2", "bls12_381_g2")])
def test_eip2537(group, cname):
    curve = po.CURVES[cname]
    for name, scalars, points, expected in _golden.eip2537(group):
        sc = curve.scalars_to_array([k % curve.order for k in scalars])
        pts = curve.points_to_array(points)
        out, _ = cref.msm(cname, sc, pts)
        assert _aff(curve, out) == expected, name


def test_msm_kat_sum_all_window_sizes():
    curve = po.BLS12_381_G1
    kats = _golden.scalar_mul_kats("bls12_381_g1")[-40:]
    expect = None
    for _, _, Q in kats:
        expect = curve.add(expect, Q)
    sc = curve.scalars_to_array([k for _, k, _ in kats])
    pts = curve.points_to_array([P for P, _, _ in kats])
    for c in range(2, 17):
        out, used = cref.msm("bls12_381_g1", sc, pts, c=c)
        assert used == c and _aff(curve, out) == expect


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "pallas"])
def test_serial_equals_parallel_medium(name):
    curve = po.CURVES[name]
    n = 3000
    pts = cref.gen_points(name, 21, n)
    sc = cref.synth_scalars(22, n, curve.scalar_bits)
    a, _ = cref.msm(name, sc, pts, nthreads=1)
    b, _ = cref.msm(name, sc, pts, nthreads=4)
    c, _ = cref.msm(name, sc, pts, nthreads=1, c=7)
    assert bytes(a) == bytes(b) == bytes(c)


@pytest.mark.parametrize("name,lg", [("bls12_381_g1", 17), ("bls12_381_g2", 14), ("bn254_snarks_g1", 17), ("bn254_snarks_g2", 14),
                                     ("pallas", 16), ("vesta", 16)])
def test_port_against_the_discrete_log_identity(name, lg):
    """The C++ port at sizes pyoracle's bucket method cannot reach (the GPU tests compare with the port at 2^20 .. 2^24): the synthetic
    points are [s_i]G with known s_i, so sum a_i P_i = [sum a_i s_i mod r]G -- big-integer arithmetic and ONE scalar multiplication
    of the Python oracle, which the reference's own vectors pin (test_scalar_mul_kats).  Endomorphism path and plain path, threads."""
    curve = po.CURVES[name]
    n = (1 << lg) + 77
    pts = cref.gen_points(name, 9000 + lg, n)
    sc = cref.synth_scalars(9100 + lg, n, curve.scalar_bits)
    sc[3] = 255                                     # a scalar above the bit width (all 256 bits set)
    sc[4] = 0
    want = cref.msm_by_discrete_logs(name, 9000 + lg, sc)
    for nthreads in (1, 8):
        out, _ = cref.msm(name, sc, pts, nthreads=nthreads)
        assert _aff(curve, out) == want, (name, nthreads)
    # a slice of the sequence (first > 0): what a rank of a sharded run generates
    h = n // 3
    out, _ = cref.msm(name, sc[h:], pts[h:])
    assert _aff(curve, out) == cref.msm_by_discrete_logs(name, 9000 + lg, sc[h:], first=h)


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "vesta"])
def test_fr_roundtrip_and_value(name):
    curve = po.CURVES[name]
    ks = [po.synth_scalar(5, i, 250) % curve.Fr.p for i in range(8)]
    can = curve.scalars_to_array(ks)
    mont = cref.fr_to_mont(name, can)
    assert bytes(mont.tobytes()) == bytes(curve.fr_scalars_to_array(ks).tobytes())
    assert bytes(cref.fr_from_mont(name, mont).tobytes()) == bytes(can.tobytes())


def test_edge_infinity_and_repeats():
    curve = po.BLS12_381_G1
    name = "bls12_381_g1"
    G = curve.gen
    pts = curve.points_to_array([G, None, G, curve.neg(G), G])
    sc = curve.scalars_to_array([5, 77, 5, 3, 0])
    out, _ = cref.msm(name, sc, pts, c=3)
    assert _aff(curve, out) == curve.scalar_mul(7, G)
    # everything cancels
    pts = curve.points_to_array([G, curve.neg(G)])
    sc = curve.scalars_to_array([9, 9])
    out, _ = cref.msm(name, sc, pts)
    assert _aff(curve, out) is None
