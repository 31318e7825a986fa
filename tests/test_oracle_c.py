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


@pytest.mark.parametrize("name", ["bls12_381_g2", "bn254_snarks_g2"])
def test_g2_endomorphism_split_of_the_port(name):
    """Round 5 (SURVEY 8 row a16 on G2): the port applies the reference's M = 4 pre-split -- psi, the 4 x 4 lattice, Babai rounding
    -- where the reference's dispatch does.  Pinned piece by piece against the big-integer oracle: psi(G) = [p mod r]G (the
    eigenvalue the lattices are built for), sum_j +-m_j lambda^j = k (mod r) with 65-bit mini-scalars for edge and random k, and
    the MSM with the split (automatic window: serial c <= 13) = the MSM without it (explicit window: the split is off) = pyoracle."""
    curve = po.CURVES[name]
    p, r = curve.F.p, curve.order
    lam = p % r
    G = curve.gen
    Garr = curve.points_to_array([G])[0]
    assert curve.aff_from_bytes(bytes(cref.psi_g2(name, Garr))) == curve.scalar_mul(lam, G)
    Q = curve.scalar_mul(0xDEADBEEFCAFE, G)
    assert curve.aff_from_bytes(bytes(cref.psi_g2(name, curve.points_to_array([Q])[0]))) == curve.scalar_mul(lam, Q)
    rng = np.random.default_rng(8)
    ks = [0, 1, 2, r - 1, r - 2, r, r + 1, (1 << curve.scalar_bits) - 1, 1 << (curve.scalar_bits - 1), lam, lam * lam % r, r // 2]
    ks += [int.from_bytes(rng.bytes(32), "little") % (1 << curve.scalar_bits) for _ in range(200)]
    for k in ks:
        mini, neg = cref.decompose_g2(name, k)
        assert all(m < (1 << 65) for m in mini), hex(k)
        assert sum((-m if s else m) * pow(lam, j, r) for j, (m, s) in enumerate(zip(mini, neg))) % r == k % r, hex(k)
    for n, seed in ((1, 3), (2, 4), (37, 5), (300, 6)):
        pts = cref.gen_points(name, 9000 + seed, n)
        sc = cref.synth_scalars(9100 + seed, n, curve.scalar_bits)
        sc[0] = np.frombuffer((r - 1).to_bytes(32, "little"), dtype=np.uint8)
        if n > 2:
            sc[1] = 0
            pts[2] = 0                                                  # a neutral point among the inputs
        auto, c_auto = cref.msm(name, sc, pts, nthreads=1)              # serial, automatic c (<= 13 here): the split applies
        assert c_auto <= 13
        plain, _ = cref.msm(name, sc, pts, nthreads=1, c=c_auto)        # explicit c: no split
        assert bytes(auto) == bytes(plain), n
        par, _ = cref.msm(name, sc, pts, nthreads=4)                    # parallel dispatch (split for c in {2..6, 9, 10})
        assert bytes(par) == bytes(plain), n
        if n <= 37:
            want = curve.msm_naive([int.from_bytes(bytes(x), "little") for x in sc], [curve.aff_from_bytes(bytes(x)) for x in pts])
            assert curve.aff_from_bytes(bytes(auto)) == want, n


def _bug366_inputs():
    """The inputs of tests/math_elliptic_curves/t_ec_shortw_jac_g2_msm_bug_366.nim:17-43, rebuilt with the restated PRNG
    (oracle/refprng.py): seed 1234, 22529 copies of the BN254-Snarks G2 generator, 22529 Fr scalars from random_long01Seq."""
    from oracle import refprng
    curve = po.CURVES["bn254_snarks_g2"]
    rng = refprng.RngState(1234)
    n = 22529
    ks = [rng.random_long01seq_field(curve.order, 4) for _ in range(n)]
    return curve, n, ks


def test_bug366_regression_on_its_real_input():
    """https://github.com/mratsim/constantine/issues/366: N = 22529 gives c = 13, BN254 G2 splits a 254-bit scalar into four 65-bit
    mini-scalars, and 13 | 65 hit an off-by-one in the window count.  The port runs the reference's configuration (automatic c, the
    M = 4 split) on the reference's input construction; all points are the generator, so the answer is [sum k_i mod r]G -- one
    scalar multiplication of the big-integer oracle, nothing shared with the port's bucket method."""
    curve, n, ks = _bug366_inputs()
    assert cref.lib().oracle_best_bucket_bit_size(n, 254) == 13                    # the c of the issue
    assert sum(1 for k in ks if k.bit_length() > 200) > n // 4                      # long runs of ones and zeros, reduced mod r
    sc = curve.scalars_to_array(ks)
    pts = np.tile(curve.points_to_array([curve.gen]), (n, 1))
    want = curve.scalar_mul(sum(ks) % curve.order, curve.gen)
    out, c_used = cref.msm("bn254_snarks_g2", sc, pts, nthreads=1)                  # multiScalarMul_vartime: c = 13, M = 4 split
    assert c_used == 13 and curve.aff_from_bytes(bytes(out)) == want
    out, _ = cref.msm("bn254_snarks_g2", sc, pts, nthreads=1, c=13)                 # no split: the 254-bit scalars themselves
    assert curve.aff_from_bytes(bytes(out)) == want
    out, _ = cref.msm("bn254_snarks_g2", sc, pts, nthreads=8)
    assert curve.aff_from_bytes(bytes(out)) == want


@pytest.mark.parametrize("name", ALL)
def test_points_with_unknown_logs_are_subgroup_points(name):
    """cref.gen_points_unknown_log (random x, square root, cofactor clearing: the reference's bench inputs, helpers/prng_unsafe.nim:306-316,
    bench_elliptic_parallel_template.nim:78-102) against the big-integer oracle, every curve of the path: on the curve, distinct, [r]P
    neutral -- and the port's MSM over them equals the oracle's naive sum (Fp square roots by exponentiation / Tonelli-Shanks, Fp2 by the
    complex method, the G2 cofactors 0x5d54...38e5 and 2p - r: all derived in the port, none shared with the kernels)."""
    curve = po.CURVES[name]
    n = 24
    pts = cref.gen_points_unknown_log(name, 0xABCDEF, n, nthreads=2)
    again = cref.gen_points_unknown_log(name, 0xABCDEF, 5, first=7, nthreads=1)
    assert bytes(again) == bytes(pts[7:12])                                   # a slice of the sequence is the sequence
    P = [curve.aff_from_bytes(bytes(p)) for p in pts]
    assert len(set(P)) == n and None not in P
    for p in P:
        assert curve.is_on_curve(p)
    for p in P[:4]:
        assert curve.scalar_mul(curve.order, p) is None
    sc = cref.synth_scalars(0xABCDF0, n, curve.scalar_bits)
    ks = [int.from_bytes(bytes(s), "little") for s in sc]
    want = curve.msm_naive(ks, P)
    for nt in (1, 3):
        assert curve.aff_from_bytes(bytes(cref.msm(name, sc, pts, nthreads=nt)[0])) == want
