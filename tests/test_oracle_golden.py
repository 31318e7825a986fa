"""Pin the CPU oracle (oracle/pyoracle.py) to the reference's golden vectors."""
import pytest

from oracle import pyoracle as po
from tests import _golden


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "pallas", "vesta"])
def test_scalar_mul_kats_g1(name):
    curve = po.CURVES[name]
    kats = _golden.scalar_mul_kats(name)
    assert len(kats) >= 40
    for P, k, Q in kats:
        assert curve.is_on_curve(P) and curve.is_on_curve(Q)
        assert curve.scalar_mul(k, P) == Q


@pytest.mark.parametrize("name", ["bls12_381_g2", "bn254_snarks_g2"])
def test_scalar_mul_kats_g2(name):
    curve = po.CURVES[name]
    kats = _golden.scalar_mul_kats(name)
    # G2 arithmetic in pure Python is slow: 255/254-bit vectors + a slice of the short ones
    for P, k, Q in kats[-40:] + kats[:10]:
        assert curve.is_on_curve(P) and curve.is_on_curve(Q)
        assert curve.scalar_mul(k, P) == Q


@pytest.mark.parametrize("name", list(po.CURVES))
def test_generators_on_curve_and_in_subgroup(name):
    curve = po.CURVES[name]
    assert curve.is_on_curve(curve.gen)
    assert curve.scalar_mul(curve.order, curve.gen) is None


@pytest.mark.parametrize("group,cname", [("g1", "bls12_381_g1"), ("g2", "bls12_381_g2")])
def test_eip2537_multiexp(group, cname):
    curve = po.CURVES[cname]
    cases = _golden.eip2537(group)
    assert len(cases) >= 13
    for name, scalars, points, expected in cases:
        assert curve.msm_naive(scalars, points) == expected, name
        # the restated bucket method must agree too, at several window sizes
        for c in (2, 3, 5):
            # scalars in these vectors may exceed 255 bits? they are 256-bit words; reduce as the
            # reference does (it parses into Fr, ethereum_evm_precompiles.nim:940-960)
            red = [k % curve.order for k in scalars]
            assert curve.msm_pippenger(red, points, c=c) == expected, (name, c)


def test_msm_of_kats_equals_sum_of_products():
    """MSM(k_i, P_i) == sum Q_i for the 40 255-bit BLS12-381 G1 vectors, every window size 2..16."""
    curve = po.BLS12_381_G1
    kats = _golden.scalar_mul_kats("bls12_381_g1")[-40:]
    expect = None
    for _, _, Q in kats:
        expect = curve.add(expect, Q)
    scalars = [k for _, k, _ in kats]
    points = [P for P, _, _ in kats]
    for c in (2, 4, 5, 8, 15, 16, 17):   # 15 and 17 divide 255: the extra-window rule
        assert curve.msm_pippenger(scalars, points, c=c) == expect, c


def test_booth_recoding_identity():
    """sum_w digit_w * 2^(w*c) == k for unreduced k < 2^bits."""
    import random
    rng = random.Random(7)
    for bits in (254, 255):
        for c in range(2, 18):
            W = bits // c + 1
            for _ in range(50):
                k = rng.getrandbits(bits)
                tot = 0
                for w in range(W):
                    v, neg = po.booth_digit(k, w, c)
                    assert 0 <= v <= 1 << (c - 1)
                    tot += (-v if neg else v) << (w * c)
                assert tot == k
            for k in (0, 1, (1 << bits) - 1, 1 << (bits - 1)):
                tot = sum((-v if neg else v) << (w * c)
                          for w in range(W) for v, neg in [po.booth_digit(k, w, c)])
                assert tot == k


def test_best_bucket_bit_size_table():
    """Values SURVEY.md §8(a5) derived by restating scheduler.nim:172-223."""
    expect = {10: 9, 12: 10, 14: 12, 16: 13, 18: 14, 20: 15, 22: 16, 24: 17}
    for lg, c in expect.items():
        assert po.best_bucket_bit_size(1 << lg, 255) == c
