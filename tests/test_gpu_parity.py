"""
GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the C ABI
(include/ctt_msm_hip.h), against the CPU oracle on the same inputs -- byte-exact on the canonical affine
image (the reference compares group elements, never raw (X,Y,Z): tests/math_elliptic_curves/t_ec_template.nim:1440-1483).

Nothing here reads /root/reference; golden vectors come from tests/golden/.
"""
import os
import random

import numpy as np
import pytest

from oracle import cref
from oracle import pyoracle as po
from tests import _golden

pytestmark = pytest.mark.gpu

ALL = list(po.CURVES)
CURVE_COORD_BYTES = {"bls12_381_g1": 48, "bls12_381_g2": 96, "bn254_snarks_g1": 32, "bn254_snarks_g2": 64, "pallas": 32, "vesta": 32}
G1S = ["bls12_381_g1", "bn254_snarks_g1", "pallas", "vesta"]
NT = max(1, min(32, os.cpu_count() or 1))  # the GPU box grants a 16-CPU quota


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def dev(torch_cuda):
    from constantine_amd import DeviceMsm
    d = DeviceMsm(0)
    yield d
    d.close()


def _aff(curve, b):
    return curve.aff_from_bytes(bytes(b))


def _decode(curve, coord, r):
    return {"jac": curve.jac_from_bytes, "prj": curve.prj_from_bytes, "aff": curve.aff_from_bytes}[coord](bytes(r))


def _to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ----------------------------------------------------------------------------------------------
# kernels below the MSM: Montgomery arithmetic and the input generator
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ALL)
def test_field_ops_on_device(name, dev, torch_cuda):
    torch = torch_cuda
    curve = po.CURVES[name]
    F = curve.F
    base = F if F.degree == 1 else F.base
    p = base.p
    rng = random.Random(11)
    n = 2048
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 1 << (p.bit_length() - 1), base.R % p, (base.R * base.R) % p]

    def rnd():
        v = rng.choice(edge) if rng.random() < 0.15 else rng.randrange(p)
        if F.degree == 1:
            return v
        return (v, rng.choice(edge) if rng.random() < 0.15 else rng.randrange(p))

    A = [rnd() for _ in range(n)]
    B = [rnd() for _ in range(n)]
    a = np.frombuffer(b"".join(F.to_mont_bytes(x) for x in A), dtype=np.uint8).reshape(n, -1)
    b = np.frombuffer(b"".join(F.to_mont_bytes(x) for x in B), dtype=np.uint8).reshape(n, -1)
    da, db = _to_dev(torch, a), _to_dev(torch, b)
    dr = torch.empty_like(da)
    ops = {0: F.mul, 1: lambda x, y: F.sqr(x), 2: F.add, 3: F.sub, 4: lambda x, y: F.neg(x)}
    for op, fn in ops.items():
        dev.field_op(name, op, da, db, dr, n)
        out = dr.cpu().numpy()
        for i in range(n):
            assert F.from_mont_bytes(bytes(out[i])) == fn(A[i], B[i]), (name, op, i)


DEV_FIELD = {"bls12_381_g1": (28, 14, 1), "bls12_381_g2": (28, 14, 2), "bn254_snarks_g1": (29, 9, 1),
             "pallas": (29, 9, 1), "vesta": (29, 9, 1)}   # (limb bits, limbs per base element, degree), fpu.h


@pytest.mark.parametrize("name", list(DEV_FIELD))
def test_carry_free_field_on_device(name, dev, torch_cuda):
    """The field the kernels compute in (fpu.h): x*R' mod p in LB-bit limbs, normalised, bounded by a few p."""
    torch = torch_cuda
    lb, nl, deg = DEV_FIELD[name]
    curve = po.CURVES[name]
    F = curve.F
    base = F if deg == 1 else F.base
    p = base.p
    Rp_inv = pow(1 << (lb * nl), -1, p)
    rng = random.Random(13)
    n = 1024
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, base.R % p, (1 << (lb * nl)) % p]

    def rnd():
        c = [rng.choice(edge) if rng.random() < 0.15 else rng.randrange(p) for _ in range(deg)]
        return c[0] if deg == 1 else tuple(c)

    A = [rnd() for _ in range(n)]
    B = [rnd() for _ in range(n)]
    a = np.frombuffer(b"".join(F.to_mont_bytes(x) for x in A), dtype=np.uint8).reshape(n, -1).copy()
    b = np.frombuffer(b"".join(F.to_mont_bytes(x) for x in B), dtype=np.uint8).reshape(n, -1).copy()
    da, db = _to_dev(torch, a), _to_dev(torch, b)
    dr = torch.zeros((n, nl * deg), dtype=torch.int32, device="cuda")
    ops = {0: (F.mul, 2), 1: (lambda x, y: F.sqr(x), 2), 2: (F.add, 4), 3: (F.sub, 4), 4: (lambda x, y: x, 2),
           5: (F.mul, 2), 6: (lambda x, y: F.sqr(x), 2), 7: (lambda x, y: F.sub(F.mul(x, y), F.sqr(x)), 4)}
    # 5-7: operands at the largest bounds, and in the lazy forms, that ec.h feeds into products (dev_field_probe)
    for op, (fn, bound) in ops.items():
        dev.field_op(name, 16 + op, da, db, dr, n)
        out = dr.cpu().numpy().astype(np.uint32)
        for i in range(n):
            comps = []
            for k in range(deg):
                limbs = [int(x) for x in out[i, k * nl:(k + 1) * nl]]
                assert all(x < (1 << lb) for x in limbs[:-1])
                v = sum(x << (lb * j) for j, x in enumerate(limbs))
                assert v < bound * p
                comps.append(v * Rp_inv % p)
            got = comps[0] if deg == 1 else tuple(comps)
            assert got == fn(A[i], B[i]), (name, op, i)


@pytest.mark.parametrize("name", ALL)
def test_gen_points_matches_oracle(name, dev, torch_cuda):
    torch = torch_cuda
    info_bytes = cref.AFF_BYTES[name]
    n = 300 if po.CURVES[name].F.degree == 1 else 64
    d = torch.empty((n, info_bytes), dtype=torch.uint8, device="cuda")
    dev.gen_points(name, 4242, n, d, first=17)
    expect = cref.gen_points(name, 4242, n, first=17)
    assert bytes(d.cpu().numpy().tobytes()) == bytes(expect.tobytes())


# ----------------------------------------------------------------------------------------------
# MSM through the Constantine-compatible C symbols (host pointers)
# ----------------------------------------------------------------------------------------------
SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 16, 32, 64, 128, 1024, 2048, 16384]  # t_ec_template.nim:1459-1483, parallel template :170-192


@pytest.mark.parametrize("name", G1S)
@pytest.mark.parametrize("n", SIZES)
def test_msm_host_symbols_vs_oracle(name, n):
    from constantine_amd import multiScalarMul_vartime, multiScalarMul_vartime_parallel
    curve = po.CURVES[name]
    pts = cref.gen_points(name, 100 + n, n)
    sc = cref.synth_scalars(200 + n, n, curve.scalar_bits)   # uniform < 2^bits, NOT reduced mod r
    expect, _ = cref.msm(name, sc, pts, nthreads=NT)
    expect = _aff(curve, expect)
    assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts, coord="jac")) == expect
    assert _decode(curve, "prj", multiScalarMul_vartime_parallel(None, name, sc, pts, coord="prj")) == expect


@pytest.mark.parametrize("name", ["bls12_381_g2", "bn254_snarks_g2"])
@pytest.mark.parametrize("n", [1, 2, 7, 64, 1024])
def test_msm_g2_vs_oracle(name, n):
    from constantine_amd import multiScalarMul_vartime
    curve = po.CURVES[name]
    pts = cref.gen_points(name, 300 + n, n)
    sc = cref.synth_scalars(400 + n, n, curve.scalar_bits)
    expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
    assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts, coord="jac")) == expect
    assert _decode(curve, "prj", multiScalarMul_vartime(name, sc, pts, coord="prj")) == expect


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "pallas", "vesta", "bls12_381_g2"])
def test_fr_coefs_symbols(name):
    from constantine_amd import multiScalarMul_vartime
    curve = po.CURVES[name]
    n = 500 if curve.F.degree == 1 else 40
    pts = cref.gen_points(name, 61, n)
    ks = [po.synth_scalar(62, i, 256) % curve.Fr.p for i in range(n)]
    can = curve.scalars_to_array(ks)
    mont = curve.fr_scalars_to_array(ks)
    expect = _aff(curve, cref.msm(name, can, pts, nthreads=NT)[0])
    assert _decode(curve, "jac", multiScalarMul_vartime(name, mont, pts, coord="jac", fr_coefs=True)) == expect


def test_halo2_zal_engine_entry():
    """CttEngine.msm -> ctt_bn254_snarks_g1_prj_multi_scalar_mul_fr_coefs_vartime_parallel (lib.rs:42-58);
    sizes 2^3..2^14 like t_zal_msm_accel.rs:28-69."""
    from constantine_amd import CttEngine
    name = "bn254_snarks_g1"
    curve = po.CURVES[name]
    eng = CttEngine(0)
    for k in (3, 8, 14):
        n = 1 << k
        pts = cref.gen_points(name, 70 + k, n)
        ks = [po.synth_scalar(71 + k, i, 256) % curve.Fr.p for i in range(n)]
        can, mont = curve.scalars_to_array(ks), curve.fr_scalars_to_array(ks)
        expect = _aff(curve, cref.msm(name, can, pts, nthreads=NT)[0])
        assert curve.prj_from_bytes(bytes(eng.msm(mont, pts))) == expect
        for table in (True, False):                   # with and without the window table (CachedBases)
            desc = eng.get_base_descriptor(pts, table=table)   # uploads + converts the bases once (lib.rs:68-71)
            assert (desc.window_bits > 0) == table
            assert curve.prj_from_bytes(bytes(eng.msm_with_cached_base(mont, desc))) == expect
            assert curve.prj_from_bytes(bytes(eng.msm_with_cached_base(mont, desc))) == expect
            desc.close()


@pytest.mark.parametrize("group,cname", [("g1", "bls12_381_g1"), ("g2", "bls12_381_g2")])
def test_eip2537_golden_vectors(group, cname):
    from constantine_amd import multiScalarMul_vartime
    curve = po.CURVES[cname]
    for name, scalars, points, expected in _golden.eip2537(group):
        sc = curve.scalars_to_array([k % curve.order for k in scalars])
        pts = curve.points_to_array(points)
        assert _decode(curve, "jac", multiScalarMul_vartime(cname, sc, pts, coord="jac")) == expected, name


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "pallas", "vesta"])
def test_scalar_mul_kat_sum(name):
    """MSM(k_i, P_i) over the reference's 40 full-width [k]P=Q vectors == sum Q_i."""
    from constantine_amd import multiScalarMul_vartime
    curve = po.CURVES[name]
    kats = _golden.scalar_mul_kats(name)[-40:]
    expect = None
    for _, _, Q in kats:
        expect = curve.add(expect, Q)
    sc = curve.scalars_to_array([k for _, k, _ in kats])
    pts = curve.points_to_array([P for P, _, _ in kats])
    assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts)) == expect


# ----------------------------------------------------------------------------------------------
# edge cases
# ----------------------------------------------------------------------------------------------
def test_edge_cases_infinity_cancellation_empty():
    from constantine_amd import multiScalarMul_vartime
    name = "bls12_381_g1"
    curve = po.CURVES[name]
    G = curve.gen
    pts = curve.points_to_array([G, None, G, curve.neg(G), G, None])
    sc = curve.scalars_to_array([5, 77, 5, 3, 0, 0])
    assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts)) == curve.scalar_mul(7, G)
    pts = curve.points_to_array([G, curve.neg(G)])
    sc = curve.scalars_to_array([9, 9])
    r = multiScalarMul_vartime(name, sc, pts, coord="jac")
    assert curve.jac_from_bytes(bytes(r)) is None
    one = curve.F.to_mont_bytes(1)
    assert bytes(r) == one + one + bytes(48)           # EC_ShortW_Jac neutral (1,1,0)
    r = multiScalarMul_vartime(name, sc, pts, coord="prj")
    assert bytes(r) == bytes(48) + one + bytes(48)     # EC_ShortW_Prj neutral (0,1,0)
    r = multiScalarMul_vartime(name, np.zeros((0, 32), np.uint8), np.zeros((0, 96), np.uint8))
    assert curve.jac_from_bytes(bytes(r)) is None      # len == 0 -> neutral
    r = multiScalarMul_vartime(name, curve.scalars_to_array([0, 0, 0]), curve.points_to_array([G, G, G]))
    assert curve.jac_from_bytes(bytes(r)) is None


def test_scalars_beyond_the_bit_width_do_not_corrupt_memory(dev, torch_cuda):
    """Scalars >= 2^bits are outside the API contract: the result is unspecified, but the sort must keep
    every record inside its arrays (the top window's narrower bucket groups clamp) and the next call must be exact."""
    torch = torch_cuda
    name = "bn254_snarks_g1"          # 254-bit scalars: bits 254, 255 set -> top-window digits beyond 2^14 buckets
    n = 50000
    pts = cref.gen_points(name, 511, n)
    bad = np.full((n, 32), 0xFF, dtype=np.uint8)
    bad[::3] = cref.synth_scalars(512, n, 254)[::3]
    dp = _to_dev(torch, pts)
    for c in (0, 16, 13):
        dev.set_option("c", c)
        dev.msm(name, _to_dev(torch, bad), dp, n, coord="aff")     # must return
    dev.set_option("c", 0)
    sc = cref.synth_scalars(513, n, 254)
    assert bytes(dev.msm(name, _to_dev(torch, sc), dp, n, coord="aff")) == bytes(cref.msm(name, sc, pts, nthreads=NT)[0])


def test_all_equal_scalars_one_bucket_per_window():
    """Adversarial distribution: every pair lands in the same bucket -> long head/tail chains + merge tree."""
    from constantine_amd import multiScalarMul_vartime
    name = "bls12_381_g1"
    curve = po.CURVES[name]
    n = 20000
    pts = cref.gen_points(name, 41, n)
    sc = np.tile(cref.synth_scalars(42, 1, 255), (n, 1))
    expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
    assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts)) == expect


def test_all_equal_points_bug366_style():
    """t_ec_shortw_jac_g2_msm_bug_366.nim: N = 22529, all points equal (P == Q additions everywhere)."""
    from constantine_amd import multiScalarMul_vartime
    for name, n in (("bn254_snarks_g1", 22529), ("bn254_snarks_g2", 2049)):
        curve = po.CURVES[name]
        pts = np.tile(cref.gen_points(name, 51, 1), (n, 1))
        sc = cref.synth_scalars(52, n, curve.scalar_bits)
        expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
        assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts)) == expect
        sc2 = np.tile(sc[:1], (n, 1))
        expect = _aff(curve, cref.msm(name, sc2, pts, nthreads=NT)[0])
        assert _decode(curve, "jac", multiScalarMul_vartime(name, sc2, pts)) == expect


def test_bug366_regression_on_its_real_input_gpu():
    """The reference's regression test itself (t_ec_shortw_jac_g2_msm_bug_366.nim:17-43): BN254-Snarks G2, N = 22529 (the N whose
    window size 13 divides the 65-bit mini-scalars of the reference's G2 split), every point the generator, Fr scalars from
    random_long01Seq of xoshiro512** seeded with 1234 (oracle/refprng.py restates the PRNG).  The reference compares its reference
    and optimised MSMs; here the HIP result, through the fr_coefs and the big_coefs symbols, is compared with [sum k_i mod r]G
    (one scalar multiplication of the big-integer oracle) and with the port running the reference's configuration."""
    from constantine_amd import multiScalarMul_vartime
    from tests.test_oracle_c import _bug366_inputs
    name = "bn254_snarks_g2"
    curve, n, ks = _bug366_inputs()
    sc = curve.scalars_to_array(ks)
    mont = curve.fr_scalars_to_array(ks)                      # cs: seq[Fr[BN254_Snarks]] -- the fr_coefs overload
    pts = np.tile(curve.points_to_array([curve.gen]), (n, 1))
    want = curve.scalar_mul(sum(ks) % curve.order, curve.gen)
    assert _decode(curve, "jac", multiScalarMul_vartime(name, mont, pts, coord="jac", fr_coefs=True)) == want
    assert _decode(curve, "prj", multiScalarMul_vartime(name, sc, pts, coord="prj")) == want
    assert _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0]) == want


@pytest.mark.parametrize("name,lg", [("bls12_381_g1", 18), ("bn254_snarks_g1", 18), ("pallas", 18), ("vesta", 18),
                                     ("bls12_381_g2", 16), ("bn254_snarks_g2", 16), ("bls12_381_g1", 20)])
def test_points_with_unknown_discrete_logs(name, lg, dev, torch_cuda):
    """Every other full-size test feeds points [s_i]G with known s_i (that is what lets the discrete-log identity check them without
    the port).  Here the points come the way the reference's benches make theirs -- random x, square root, cofactor clearing
    (helpers/prng_unsafe.nim:185-190,306-316, bench_elliptic_parallel_template.nim:78-102): nobody knows their logarithms, no structure for a
    bug to hide behind.  Every curve of the path (round 6; round 5 had the two G1 curves at 2^18): 2^18 for the G1 curves, 2^16 for the G2
    curves (a 512-bit cofactor multiplication per point on the host), and BLS12-381 G1 at the metric's own 2^20.  The HIP result
    (device-resident, and through the Constantine symbol on host arrays) against the port."""
    from constantine_amd import multiScalarMul_vartime, multiScalarMul_vartime_parallel
    from constantine_amd import CURVES as INFO
    torch = torch_cuda
    curve = po.CURVES[name]
    n = 1 << lg
    pts = cref.gen_points_unknown_log(name, 0xC0FFEE, n)
    for i in (0, 1, n // 2, n - 1):
        assert curve.is_on_curve(curve.aff_from_bytes(bytes(pts[i])))
    assert curve.scalar_mul(curve.order, curve.aff_from_bytes(bytes(pts[7]))) is None
    sc = cref.synth_scalars(0xC0FFEF, n, curve.scalar_bits)
    expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
    dp, ds = _to_dev(torch, pts), _to_dev(torch, sc)
    assert _aff(curve, dev.msm(name, ds, dp, n, coord="aff")) == expect
    if INFO[name].has_parallel:
        assert _decode(curve, "jac", multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")) == expect
    else:
        assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts, coord="jac")) == expect
    # a ragged prefix too (not a power of two, sorted / merged differently)
    m = n - 12345
    expect = _aff(curve, cref.msm(name, sc[:m], pts[:m], nthreads=NT)[0])
    assert _aff(curve, dev.msm(name, ds[:m], dp[:m], m, coord="aff")) == expect


def test_window_sizes_and_lane_spans(dev, torch_cuda):
    """Same answer for every plan: window bits (incl. divisors of the scalar width), entries per lane, sort slices."""
    torch = torch_cuda
    name = "bls12_381_g1"
    curve = po.CURVES[name]
    n = 5000
    pts = cref.gen_points(name, 91, n)
    sc = cref.synth_scalars(92, n, 255)
    sc[:50, 24:31] = 0xFF
    sc[:50, 31] |= 0x7F
    expect = bytes(cref.msm(name, sc, pts, nthreads=NT)[0])
    dp, ds = _to_dev(torch, pts), _to_dev(torch, sc)
    try:
        for c, K, S in ((3, 4, 0), (5, 8, 64), (8, 16, 1000), (13, 0, 0), (15, 12, 3000), (16, 0, 0), (0, 0, 0)):
            dev.set_option("c", c)
            dev.set_option("K", K)
            dev.set_option("S", S)
            assert bytes(dev.msm(name, ds, dp, n, coord="aff")) == expect, (c, K, S, dev.last_plan())
    finally:
        dev.set_option("c", 0)
        dev.set_option("K", 0)
        dev.set_option("S", 0)


def test_horner_groups_and_merge_without_host_wait(dev, torch_cuda):
    """Round 3: the bit Horner of a window is cut into groups of hb bits (one quad of lanes each, joined by the host) and
    the head merge decides on the device how far its tree goes.  Same element for every group size, for uniform digits and
    for the adversarial inputs whose head chains are far longer than the steps the plan enqueues (all scalars equal; a
    quarter of them equal)."""
    torch = torch_cuda
    name = "bls12_381_g1"
    n = 60000
    pts = cref.gen_points(name, 191, n)
    sc = cref.synth_scalars(192, n, 255)
    sc_eq = np.tile(sc[:1], (n, 1))
    sc_q = sc.copy()
    sc_q[::4] = sc[1]
    dp = _to_dev(torch, pts)
    try:
        for label, s in (("uniform", sc), ("all equal", sc_eq), ("quarter equal", sc_q)):
            expect = bytes(cref.msm(name, s, pts, nthreads=NT)[0])
            ds = _to_dev(torch, s)
            for c, hb, K in ((0, 0, 0), (13, 1, 0), (13, 3, 8), (16, 2, 0), (16, 15, 0), (9, 4, 4), (11, 8, 0), (15, 5, 12)):
                dev.set_option("c", c)
                dev.set_option("K", K)
                dev.set_option("horner_bits", hb)
                assert bytes(dev.msm(name, ds, dp, n, coord="aff")) == expect, (label, c, hb, K, dev.last_plan())
        # more than 16 groups per window (horner_bits = 1, or the legacy host_window_sums = 1, with c >= 18 gives
        # 17 .. 19 groups = 68 .. 76 lanes: the groups beyond a workgroup's 16 quads run in further blocks)
        expect = bytes(cref.msm(name, sc, pts, nthreads=NT)[0])
        ds = _to_dev(torch, sc)
        for c, hb, hws in ((18, 1, 0), (19, 1, 0), (20, 1, 0), (20, 0, 1), (18, 0, 1), (20, 2, 0)):
            dev.set_option("c", c)
            dev.set_option("K", 0)
            dev.set_option("horner_bits", hb)
            dev.set_option("host_window_sums", hws)
            assert bytes(dev.msm(name, ds, dp, n, coord="aff")) == expect, (c, hb, hws, dev.last_plan())
    finally:
        for k in ("c", "K", "horner_bits", "host_window_sums"):
            dev.set_option(k, 0)


def test_repeated_lone_calls_on_one_context(torch_cuda):
    """Blocking calls on a fresh context, the way a caller without a pipeline uses it: the same buffers again and again, the same
    buffers with new contents, a larger MSM in between (the grow-only workspace is reallocated), another plan for the same inputs,
    two MSMs in flight in between, stage timings on and off, prefixes of cached bases, another curve on the same context.
    (Round 4 wrote it for the HIP-graph replay of lone MSMs -- measured no gain, profiles/graph_lone_msm_r04.txt, code removed --
    and it stays as the test of exactly the state that replay would have had to get right.)"""
    torch = torch_cuda
    from constantine_amd import CachedBases, DeviceMsm
    name = "bls12_381_g1"
    eng = DeviceMsm(0)     # a fresh context: its workspace grows inside this test
    try:
        n = 20000
        pts = cref.gen_points(name, 901, n)
        sc1, sc2 = cref.synth_scalars(902, n, 255), cref.synth_scalars(903, n, 255)
        exp1, exp2 = bytes(cref.msm(name, sc1, pts, nthreads=NT)[0]), bytes(cref.msm(name, sc2, pts, nthreads=NT)[0])
        d_p, d_s = _to_dev(torch, pts), _to_dev(torch, sc1)
        for _ in range(5):
            assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp1
        d_s.copy_(torch.from_numpy(sc2))                              # same buffer, new contents
        assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp2
        n2 = 300000                                                   # a larger MSM reallocates the workspace
        pts2, sc3 = cref.gen_points(name, 904, n2), cref.synth_scalars(905, n2, 255)
        exp3 = bytes(cref.msm(name, sc3, pts2, nthreads=NT)[0])
        d_p2, d_s3 = _to_dev(torch, pts2), _to_dev(torch, sc3)
        assert bytes(eng.msm(name, d_s3, d_p2, n2, coord="aff")) == exp3
        for _ in range(4):
            assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp2
            assert bytes(eng.msm(name, d_s3, d_p2, n2, coord="aff")) == exp3
        eng.set_option("c", 11)                                       # another plan for the same inputs
        for _ in range(3):
            assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp2
        eng.set_option("c", 0)
        a, b = eng.submit(name, d_s, d_p, n), eng.submit(name, d_s3, d_p2, n2)   # two in flight: tail stream
        assert bytes(eng.finish(a)) == exp2 and bytes(eng.finish(b)) == exp3
        for _ in range(3):
            assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp2
        eng.enable_timings(True)
        assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp2 and eng.last_timings()["total"] > 0
        eng.enable_timings(False)
        assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp2
        # prefixes of cached bases, host scalars staged through the context's buffer (the KZG commitment's call shape)
        bases = CachedBases(name, pts, ctx=eng.ctx)
        try:
            for m, sc in ((n, sc1), (n, sc2), (n // 2, sc1[:n // 2]), (n, sc1), (n // 2, sc2[:n // 2]), (n, sc2), (n // 2, sc1[:n // 2])):
                want = bytes(cref.msm(name, sc, pts[:m], nthreads=NT)[0])
                assert bytes(bases.msm(sc, coord="aff")) == want, m
        finally:
            bases.close()
        # another curve on the same context
        pv = cref.gen_points("vesta", 906, 3000)
        sv = cref.synth_scalars(907, 3000, 255)
        ev = bytes(cref.msm("vesta", sv, pv, nthreads=NT)[0])
        d_pv, d_sv = _to_dev(torch, pv), _to_dev(torch, sv)
        for _ in range(4):
            assert bytes(eng.msm("vesta", d_sv, d_pv, 3000, coord="aff")) == ev
            assert bytes(eng.msm(name, d_s, d_p, n, coord="aff")) == exp2
    finally:
        eng.close()


def test_tickets_finished_out_of_order(dev, torch_cuda):
    """with one ticket outstanding, blocking calls must keep working (whichever slot is free is taken)."""
    torch = torch_cuda
    name = "vesta"
    n = 3000
    pts = cref.gen_points(name, 195, n)
    sc = cref.synth_scalars(196, n, 255)
    expect = bytes(cref.msm(name, sc, pts, nthreads=NT)[0])
    ds, dp = _to_dev(torch, sc), _to_dev(torch, pts)
    a = dev.submit(name, ds, dp, n)
    for _ in range(3):
        assert bytes(dev.msm(name, ds, dp, n)) == expect      # B, C, D while A is outstanding
    b = dev.submit(name, ds, dp, n)
    assert bytes(dev.finish(b)) == expect                      # newer ticket first
    assert bytes(dev.msm(name, ds, dp, n)) == expect
    assert bytes(dev.finish(a)) == expect


@pytest.mark.parametrize("n", [16385, 40000, 100003, (1 << 20) + 7])
def test_sizes_around_the_sort_group_boundaries(dev, torch_cuda, n):
    """Pair counts that are not powers of two (ragged last partition block, groups of uneven size)."""
    torch = torch_cuda
    name = "pallas"
    pts = cref.gen_points(name, 411, n)
    sc = cref.synth_scalars(412, n, 255)
    expect = bytes(cref.msm(name, sc, pts, nthreads=NT)[0])
    assert bytes(dev.msm(name, _to_dev(torch, sc), _to_dev(torch, pts), n, coord="aff")) == expect


def test_sort_under_skewed_digit_distributions(dev, torch_cuda):
    """The two-pass bucket sort (partition by bucket group, LDS sort per group) must not depend on the digits being
    uniform: giant buckets (bypass the LDS image), groups several tiles long, medium buckets straddling a tile end."""
    torch = torch_cuda
    name = "bn254_snarks_g1"
    n = 1 << 18
    pts = cref.gen_points(name, 311, n)
    dp = _to_dev(torch, pts)
    rng = np.random.default_rng(7)
    uni = cref.synth_scalars(312, n, 254)
    cases = {"uniform": uni}
    few = uni[:5]
    cases["five distinct scalars"] = few[rng.integers(0, 5, n)]
    some = uni[:100]
    cases["hundred distinct scalars"] = some[rng.integers(0, 100, n)]
    low = uni.copy()
    low[:, 1::2] &= 0x1F          # every 16-bit chunk < 2^13: three quarters of the bucket range stay empty
    cases["low quarter of the bucket range"] = low
    mix = uni.copy()
    mix[: n // 2] = some[rng.integers(0, 100, n // 2)]
    cases["half uniform, half repeated"] = mix
    try:
        for label, sc in cases.items():
            sc = np.ascontiguousarray(sc)
            expect = bytes(cref.msm(name, sc, pts, nthreads=NT)[0])
            for c in ((0, 11, 12, 13, 14, 15) if label == "uniform" else (0, 12)):   # 11..14: short top windows
                dev.set_option("c", c)
                assert bytes(dev.msm(name, _to_dev(torch, sc), dp, n, coord="aff")) == expect, (label, c)
            # round 4: the partition pass has two forms (records staged through LDS / one store per record) and two block-to-slice
            # mappings (XCD-aware / plain); the older forms still serve more than 1024 groups per window: every combination here
            for staged, xcd in ((0, 0), (0, 1), (2, 0), (2, 1)):
                dev.set_option("sort_staged", staged)
                dev.set_option("sort_xcd", xcd)
                for c in (0, 13):
                    dev.set_option("c", c)
                    assert bytes(dev.msm(name, _to_dev(torch, sc), dp, n, coord="aff")) == expect, (label, c, staged, xcd)
            dev.set_option("sort_staged", 1)
            dev.set_option("sort_xcd", 1)
    finally:
        dev.set_option("c", 0)
        dev.set_option("sort_staged", 1)
        dev.set_option("sort_xcd", 1)


# ----------------------------------------------------------------------------------------------
# BASELINE.json sizes: every config is byte-compared with the oracle on its FULL input
# ----------------------------------------------------------------------------------------------
def _full_size(name, lg, dev, torch, seed=None):
    """Oracle on the whole input (oracle/msm_ref.cpp: ~1.2 M pairs/s on the box's 16 CPUs), then two size-independent
    properties the reference's tests cannot use: the sum of the half-MSMs, and a different plan giving the same element."""
    from constantine_amd.msm import ec_sum_affine
    curve = po.CURVES[name]
    n = 1 << lg
    info_bytes = cref.AFF_BYTES[name]
    dp = torch.empty((n, info_bytes), dtype=torch.uint8, device="cuda")
    dev.gen_points(name, seed or (1000 + lg), n, dp)
    sc = cref.synth_scalars(2000 + lg, n, curve.scalar_bits)
    ds = _to_dev(torch, sc)
    full = dev.msm(name, ds, dp, n, coord="aff")
    expect, _ = cref.msm(name, sc, dp.cpu().numpy(), nthreads=NT)
    assert bytes(expect) == bytes(full)
    # independent of the C++ port: the points are [s_i]G with known s_i, so the MSM is [sum a_i s_i mod r]G -- one scalar
    # multiplication of the big-integer oracle (pinned by the reference's vectors), no bucket method anywhere
    assert curve.aff_from_bytes(bytes(full)) == cref.msm_by_discrete_logs(name, seed or (1000 + lg), sc)
    h = n // 2
    a = dev.msm(name, ds[:h], dp[:h], h, coord="aff")
    b = dev.msm(name, ds[h:], dp[h:], n - h, coord="aff")
    assert bytes(ec_sum_affine(name, np.stack([a, b]))) == bytes(full)
    dev.set_option("c", 13)
    dev.set_option("K", 36)
    try:
        assert bytes(dev.msm(name, ds, dp, n, coord="aff")) == bytes(full)
    finally:
        dev.set_option("c", 0)
        dev.set_option("K", 0)


@pytest.mark.parametrize("name,n", [("bls12_381_g1", (1 << 17) + 777), ("bls12_381_g1", 3 * (1 << 16) + 1), ("bls12_381_g1", 1000003),
                                    ("bn254_snarks_g1", (1 << 18) + 12345), ("pallas", 299999)])
def test_sizes_that_are_not_powers_of_two(name, n, dev, torch_cuda):
    """The plan's roundings (entries per lane so that the accumulate grid fits the resident wave slots, equal partition
    blocks, bucket-group count) at sizes that are not powers of two, against the oracle; with the window table as well."""
    from constantine_amd import CachedBases
    torch = torch_cuda
    curve = po.CURVES[name]
    dp = torch.empty((n, cref.AFF_BYTES[name]), dtype=torch.uint8, device="cuda")
    dev.gen_points(name, 31337 + n, n, dp)
    sc = cref.synth_scalars(4000 + n, n, curve.scalar_bits)
    ds = _to_dev(torch, sc)
    expect, _ = cref.msm(name, sc, dp.cpu().numpy(), nthreads=NT)
    assert bytes(dev.msm(name, ds, dp, n, coord="aff")) == bytes(expect)
    bases = CachedBases(name, dp, ctx=dev.ctx, on_device=True, table=True)
    try:
        assert bytes(bases.msm(ds, coord="aff")) == bytes(expect)
    finally:
        bases.close()


def test_bls12_381_g1_2pow20(dev, torch_cuda):
    """BASELINE config 2: BLS12-381 G1, 2^20 pairs, bit-exact vs the CPU path."""
    _full_size("bls12_381_g1", 20, dev, torch_cuda)


def test_bls12_381_g1_2pow24_full_oracle(dev, torch_cuda):
    """BASELINE config 4 (2^24 pairs; 8 GPUs there, one here): the oracle on all 2^24 pairs (~15 s of CPU), and the
    sharded-sum identity over the 8 balanced slices the 8-GPU run computes (MSM = sum of the per-rank MSMs)."""
    torch = torch_cuda
    from constantine_amd.msm import ec_sum_affine
    from constantine_amd.parallel import shard_bounds
    name = "bls12_381_g1"
    n = 1 << 24
    dp = torch.empty((n, 96), dtype=torch.uint8, device="cuda")
    dev.gen_points(name, 2424, n, dp)
    sc = cref.synth_scalars(2425, n, 255)
    ds = _to_dev(torch, sc)
    full = dev.msm(name, ds, dp, n, coord="aff")
    parts = []
    for r in range(8):
        s0, ln = shard_bounds(n, 8, r)
        parts.append(dev.msm(name, ds[s0:s0 + ln], dp[s0:s0 + ln], ln, coord="aff"))
    assert bytes(ec_sum_affine(name, np.stack(parts))) == bytes(full)
    expect, _ = cref.msm(name, sc, dp.cpu().numpy(), nthreads=NT)
    assert bytes(expect) == bytes(full)
    assert po.CURVES[name].aff_from_bytes(bytes(full)) == cref.msm_by_discrete_logs(name, 2424, sc)   # (no port involved: _full_size)


def test_bn254_g1_2pow22_zal_entry_full_oracle(dev, torch_cuda):
    """BASELINE config 3: BN254-Snarks G1, 2^22 pairs THROUGH THE HALO2-ZAL ENTRY
    (CttEngine.msm -> ctt_bn254_snarks_g1_prj_multi_scalar_mul_fr_coefs_vartime_parallel, lib.rs:42-58): Montgomery Fr
    coefficients on host pointers, projective result, against the oracle on all 2^22 pairs."""
    torch = torch_cuda
    from constantine_amd import CttEngine
    name = "bn254_snarks_g1"
    curve = po.CURVES[name]
    n = 1 << 22
    dp = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    dev.gen_points(name, 2222, n, dp)
    pts = dp.cpu().numpy()
    mont = cref.synth_scalars(2223, n, 253)          # any 253-bit pattern is a valid Fr Montgomery residue (r > 2^253)
    can = cref.fr_from_mont(name, mont)              # the canonical scalars the entry point computes with
    expect = _aff(curve, cref.msm(name, can, pts, nthreads=NT)[0])
    assert expect == cref.msm_by_discrete_logs(name, 2222, can)      # the port against the discrete-log identity (no bucket method)
    assert curve.prj_from_bytes(bytes(CttEngine(0).msm(mont, pts))) == expect
    # the device-resident path on the same pairs (what bench.py --curve bn254_snarks_g1 --log2n 22 times)
    assert _aff(curve, dev.msm(name, _to_dev(torch, mont), dp, n, coord="aff", fr_coefs=True)) == expect
    _full_size(name, 22, dev, torch)


def test_pasta_and_g2_2pow20_full_oracle(dev, torch_cuda):
    """BASELINE config 5: BLS12-381 G2, Pallas and Vesta at 2^20 pairs, each against the oracle on the full input."""
    _full_size("pallas", 20, dev, torch_cuda)
    _full_size("vesta", 20, dev, torch_cuda)
    _full_size("bls12_381_g2", 20, dev, torch_cuda)


def test_two_msms_in_flight(dev, torch_cuda):
    """submit/finish split: results of pipelined MSMs (different sizes, shared workspace, the tail of one running under
    the first kernels of the next) are independent."""
    torch = torch_cuda
    name = "bls12_381_g1"
    sizes = [3000, 70000, 1, 4096, 33333, 1 << 17, 5]
    data = []
    for i, n in enumerate(sizes):
        pts = cref.gen_points(name, 700 + i, n)
        sc = cref.synth_scalars(800 + i, n, 255)
        data.append((n, _to_dev(torch, sc), _to_dev(torch, pts), bytes(cref.msm(name, sc, pts, nthreads=NT)[0])))
    pending = dev.submit(name, data[0][1], data[0][2], data[0][0])
    for i in range(len(data)):
        nxt = dev.submit(name, data[i + 1][1], data[i + 1][2], data[i + 1][0]) if i + 1 < len(data) else None
        assert bytes(dev.finish(pending, coord="aff")) == data[i][3], sizes[i]
        pending = nxt


def test_host_symbols_upload_in_slices():
    """The host-pointer symbols upload the pairs in slices underneath the accumulation (MsmEngine::submit_host; round 4: ONE bucket set,
    k_accum<FD, INTO> continues the stored sums; from three slices on the copies come from a helper thread): same element for 1, 2,
    3, 4, 8 slices and for the automatic choice, ragged sizes -- against the port and against the discrete-log identity."""
    from constantine_amd import _lib, multiScalarMul_vartime, multiScalarMul_vartime_parallel
    L = _lib.lib()
    try:
        for name, n in (("bls12_381_g1", (1 << 20) + 3), ("bn254_snarks_g1", 300001), ("bls12_381_g2", 40000), ("bls12_381_g2", (1 << 17) + 5),
                        ("pallas", 7)):
            curve = po.CURVES[name]
            pts = cref.gen_points(name, 600 + n, n)
            sc = cref.synth_scalars(601 + n, n, curve.scalar_bits)
            expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
            assert expect == cref.msm_by_discrete_logs(name, 600 + n, sc)
            for chunks in ((0, 1, 2, 3, 4, 8) if n > 100000 else (0, 2, 3)):
                assert L.ctt_hip_msm_set_option(None, b"chunks", chunks) == 0
                assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts, coord="jac")) == expect, (name, chunks)
            if name == "bn254_snarks_g1":
                mont = cref.synth_scalars(602, n, 253)
                exp2 = _aff(curve, cref.msm(name, cref.fr_from_mont(name, mont), pts, nthreads=NT)[0])
                L.ctt_hip_msm_set_option(None, b"chunks", 3)
                assert _decode(curve, "prj", multiScalarMul_vartime_parallel(None, name, mont, pts, coord="prj", fr_coefs=True)) == exp2
    finally:
        L.ctt_hip_msm_set_option(None, b"chunks", 0)


def test_stage_timings_are_opt_in(torch_cuda):
    """ctt_hip_msm_last_timings reads HIP events that are only recorded after set_option("timings", 1)."""
    torch = torch_cuda
    from constantine_amd import DeviceMsm
    name = "pallas"
    n = 30000
    ds = _to_dev(torch, cref.synth_scalars(5, n, 255))
    dp = _to_dev(torch, cref.gen_points(name, 6, n))
    d = DeviceMsm(0)
    try:
        r0 = bytes(d.msm(name, ds, dp, n))
        assert all(v == 0.0 for v in d.last_timings().values())
        d.enable_timings()
        assert bytes(d.msm(name, ds, dp, n)) == r0
        t = d.last_timings()
        assert t["accumulate"] > 0.0 and t["total"] >= t["accumulate"]
        # "timings_every" k: only every k-th MSM records its events, the others report zeros (bench.py below 2^20 pairs);
        # mode 2 records the accumulate stage and the total only
        d.enable_timings(2)
        d.set_option("timings_every", 3)
        seen = []
        for _ in range(6):
            assert bytes(d.msm(name, ds, dp, n)) == r0
            t = d.last_timings()
            seen.append(t["total"] > 0.0)
            assert t["sort"] == 0.0 and (t["accumulate"] > 0.0) == seen[-1]
        assert seen == [True, False, False, True, False, False]
    finally:
        d.close()


def test_api_misuse_returns_error_codes(dev, torch_cuda):
    """Recoverable misuse of the device-resident interface is an error code (RuntimeError here), not an abort: a fourth
    ticket on a curve (three slots per engine since round 6), finishing a ticket twice, a blocking call while three tickets are
    outstanding, cached bases used with another context."""
    torch = torch_cuda
    from constantine_amd import CachedBases, DeviceMsm
    name = "pallas"
    n = 2000
    pts = cref.gen_points(name, 31, n)
    sc = cref.synth_scalars(32, n, 255)
    expect = bytes(cref.msm(name, sc, pts, nthreads=NT)[0])
    ds, dp = _to_dev(torch, sc), _to_dev(torch, pts)
    t1 = dev.submit(name, ds, dp, n)
    t2 = dev.submit(name, ds, dp, n)
    t3 = dev.submit(name, ds, dp, n)
    from constantine_amd import _lib
    L = _lib.lib()
    with pytest.raises(RuntimeError):
        dev.submit(name, ds, dp, n)              # fourth ticket
    assert L.ctt_hip_last_error() == -5          # ERR_BUSY (round 6; -1 before): "finish a ticket, or retry" -- not "bad arguments"
    with pytest.raises(RuntimeError):
        dev.msm(name, ds, dp, n)                 # blocking call needs a free slot
    assert L.ctt_hip_last_error() == -5
    assert bytes(dev.finish(t1)) == expect
    with pytest.raises(RuntimeError):
        dev.finish(t1)                           # finished already
    assert L.ctt_hip_last_error() == -1          # a bare refusal is "bad arguments", and no stale -5 survives into it
    assert bytes(dev.finish(t3)) == expect       # (any order)
    assert bytes(dev.finish(t2)) == expect
    assert bytes(dev.msm(name, ds, dp, n)) == expect
    with pytest.raises(KeyError):
        dev.set_option("lanes", 2)               # removed option
    other = DeviceMsm(0)
    try:
        bases = CachedBases(name, pts, ctx=other.ctx)
        assert bytes(bases.msm(sc, coord="aff")) == expect
        bases.ctx = dev.ctx                       # wrong context: refused, not a wild pointer
        with pytest.raises(RuntimeError):
            bases.msm(sc, coord="aff")
        bases.ctx = other.ctx
        bases.close()
    finally:
        other.close()


def test_engine_stream_is_ordered_after_torch(dev, torch_cuda):
    """Inputs produced by torch kernels that may still be running when the engine is called: the Python
    mirror orders the engine's stream after torch's current stream (ctt_hip_msm_wait_stream)."""
    torch = torch_cuda
    name = "bn254_snarks_g1"
    n = 1 << 16
    pts = cref.gen_points(name, 41, n)
    sc = cref.synth_scalars(42, n, 254)
    expect = bytes(cref.msm(name, sc, pts, nthreads=NT)[0])
    junk = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    good = _to_dev(torch, sc)
    dp = _to_dev(torch, pts)
    big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        ds = junk.clone()
        for _ in range(4):
            big.add_(1)          # keep torch's stream busy in front of the copy that makes the real input
        ds.copy_(good)
        assert bytes(dev.msm(name, ds, dp, n)) == expect


def test_host_symbols_shard_over_distinct_devices(torch_cuda):
    """The in-library sharding on a node with more than one GPU: every context on its own device (skipped on the 1-GPU boxes
    of the pool, where test_host_symbols_shard_over_contexts runs the same path with the contexts sharing device 0)."""
    torch = torch_cuda
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs at least two GPUs")
    from constantine_amd import multiScalarMul_vartime_parallel, set_devices, set_shard_min
    name = "bls12_381_g1"
    n = 100003
    pts = cref.gen_points(name, 61, n, nthreads=NT)
    sc = cref.synth_scalars(62, n, 255)
    expect = _aff(po.CURVES[name], cref.msm(name, sc, pts, nthreads=NT)[0])
    try:
        set_shard_min(1000)
        for g in sorted({2, min(4, ndev), ndev}):
            set_devices(list(range(g)))
            assert _decode(po.CURVES[name], "jac", multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")) == expect, g
    finally:
        set_devices([])
        set_shard_min(1 << 15)


def test_host_symbols_shard_over_contexts():
    """In-library multi-GPU path on this single-GPU box: two (three) contexts on device 0, the Constantine symbols cut the
    call into balanced slices (partitioners.nim:44-77), one host thread per context, host sum of the partials -- against
    the oracle on the whole input."""
    from constantine_amd import multiScalarMul_vartime, multiScalarMul_vartime_parallel, set_devices, set_shard_min
    try:
        set_shard_min(1000)
        for devices, name, n in (([0, 0], "bls12_381_g1", 70001), ([0, 0, 0], "bn254_snarks_g1", 50000),
                                 ([0, 0], "bls12_381_g2", 5000), ([0, 0], "pallas", 1500)):   # 1500 < 2 x shard_min: one GPU
            set_devices(devices)
            curve = po.CURVES[name]
            pts = cref.gen_points(name, 900 + n, n)
            sc = cref.synth_scalars(901 + n, n, curve.scalar_bits)
            expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
            assert _decode(curve, "jac", multiScalarMul_vartime(name, sc, pts, coord="jac")) == expect
            if name != "bls12_381_g2":   # the fr_coefs / projective / parallel symbol (the ZAL entry's shape) on a shorter input
                m = 4000 if n >= 4000 else n
                mont = cref.synth_scalars(902 + n, m, 253)     # < 2^253 < r: valid Montgomery residues
                exp2 = _aff(curve, cref.msm(name, cref.fr_from_mont(name, mont), pts[:m], nthreads=NT)[0])
                got = multiScalarMul_vartime_parallel(None, name, mont, pts[:m], coord="prj", fr_coefs=True)
                assert _decode(curve, "prj", got) == exp2
    finally:
        set_devices([])
        set_shard_min(1 << 15)


@pytest.mark.parametrize("name", ["bls12_381_g1", "pallas", "bls12_381_g2", "bn254_snarks_g2"])
def test_cached_bases_prefix_and_reuse(name):
    """ctt_hip_msm_bases_*: bases converted once, reused with different coefficient vectors and prefixes."""
    from constantine_amd import CachedBases
    curve = po.CURVES[name]
    n = 3000 if curve.F.degree == 1 else 300
    pts = cref.gen_points(name, 901, n)
    bases = CachedBases(name, pts)
    try:
        for seed, m in ((1, n), (2, n), (3, n // 3), (4, 1)):
            sc = cref.synth_scalars(seed, m, curve.scalar_bits)
            expect = _aff(curve, cref.msm(name, sc, pts[:m], nthreads=NT)[0])
            assert _decode(curve, "jac", bases.msm(sc, coord="jac")) == expect
    finally:
        bases.close()


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "pallas", "vesta", "bls12_381_g2", "bn254_snarks_g2"])
def test_window_table_cached_bases_vs_oracle(name):
    """ctt_hip_msm_bases_create_table: the multiples 2^(c*w) * P of the bases are resident, every digit window selects a
    table row, all windows share one bucket set (the merged form of the sort), no window combine.  Oracle parity for the
    automatic c and explicit ones (small c: many windows and long bucket chains; c dividing the scalar width: the extra
    window), prefixes of the bases, a neutral base, a repeated pair, Fr Montgomery coefficients."""
    from constantine_amd import CachedBases
    curve = po.CURVES[name]
    n = 20000 if curve.F.degree == 1 else 1500
    pts = cref.gen_points(name, 911, n)
    pts[11] = 0
    pts[13] = pts[12]
    from constantine_amd import _lib
    set_default = lambda k, v: _lib.lib().ctt_hip_msm_set_option(None, k.encode(), int(v))   # noqa: E731  (the default context's options)
    for wb in (0, 5, 12, 15) if curve.F.degree == 1 else (0, 9):
        bases = CachedBases(name, pts, table=True, window_bits=wb)
        try:
            assert bases.window_bits == wb or (wb == 0 and bases.window_bits > 0)
            for seed, m in ((1, n), (3, n // 3), (4, 1)):
                sc = cref.synth_scalars(seed, m, curve.scalar_bits)
                if m > 13:
                    sc[13] = sc[12]
                expect = _aff(curve, cref.msm(name, sc, pts[:m], nthreads=NT)[0])
                assert _decode(curve, "jac", bases.msm(sc, coord="jac")) == expect, (name, wb, m)
                # both forms of the partition pass's second sweep on the merged (one bucket set, 64-bit records) sort (round 4)
                try:
                    for staged in (0, 2):
                        set_default("sort_staged", staged)
                        assert _decode(curve, "jac", bases.msm(sc, coord="jac")) == expect, (name, wb, m, staged)
                finally:
                    set_default("sort_staged", 1)
            if curve.F.degree == 1:
                mont = cref.synth_scalars(5, n, 250)
                expect = _aff(curve, cref.msm(name, cref.fr_from_mont(name, mont), pts, nthreads=NT)[0])
                assert _decode(curve, "prj", bases.msm(mont, coord="prj", fr_coefs=True)) == expect
            assert _decode(curve, "aff", bases.msm(np.zeros((n, 32), np.uint8), coord="aff")) is None
        finally:
            bases.close()


@pytest.mark.parametrize("name,log2n", [("bls12_381_g1", 20), ("bn254_snarks_g1", 20), ("bn254_snarks_g1", 22),
                                        ("bls12_381_g2", 20)])
def test_window_table_full_size_vs_oracle(name, log2n):
    """The window-table form at a BASELINE size against the oracle (the automatic c: 20 at 2^20 bases, one set of 2^19
    buckets, 13 table rows per base), device-resident coefficients, two MSMs in flight."""
    import torch
    from constantine_amd import CachedBases, DeviceMsm
    from constantine_amd.msm import CURVES
    from constantine_amd.synth import synth_scalars
    curve = po.CURVES[name]
    info = CURVES[name]
    n = 1 << log2n
    eng = DeviceMsm(0)
    try:
        d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
        eng.gen_points(name, 0x7AB1E, n, d_points)
        bases = CachedBases(name, d_points, ctx=eng.ctx, on_device=True, table=True)
        try:
            assert bases.window_bits >= 17
            sc_a = synth_scalars(21, n, info.scalar_bits)
            sc_b = synth_scalars(22, n, info.scalar_bits)
            d_a, d_b = torch.from_numpy(sc_a).cuda(), torch.from_numpy(sc_b).cuda()
            ta = bases.submit(d_a, n)
            tb = bases.submit(d_b, n)
            ra, rb = bases.finish(ta, coord="aff"), bases.finish(tb, coord="aff")
            pts = d_points.cpu().numpy()
            assert _decode(curve, "aff", ra) == _aff(curve, cref.msm(name, sc_a, pts, nthreads=NT)[0])
            assert _decode(curve, "aff", rb) == _aff(curve, cref.msm(name, sc_b, pts, nthreads=NT)[0])
            # and the table-less engine on the same inputs
            assert bytes(eng.msm(name, d_a, d_points, n, coord="aff")) == bytes(ra)
        finally:
            bases.close()
    finally:
        eng.close()


@pytest.mark.parametrize("name", ALL)
def test_neutral_host_pointer_symbols(name):
    """Header part 1c (round 4): ctt_hip_msm_<curve>_<coord>_<coefs> and ctt_hip_msm_host give the Constantine symbols' result and a status
    instead of an abort -- what a binding inside libconstantine imports (INTEGRATION.md part B)."""
    from constantine_amd import msm_available, msm_host, multiScalarMul_vartime
    from constantine_amd import _lib
    import ctypes
    assert msm_available()
    curve = po.CURVES[name]
    n = 777
    pts = cref.gen_points(name, 3100, n)
    sc = cref.synth_scalars(3101, n, curve.scalar_bits)
    expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
    for coord in ("jac", "prj"):
        assert _decode(curve, coord, msm_host(name, sc, pts, coord=coord)) == expect
        assert _decode(curve, coord, msm_host(name, sc, pts, coord=coord, typed=False)) == expect
        assert bytes(msm_host(name, sc, pts, coord=coord)) == bytes(multiScalarMul_vartime(name, sc, pts, coord=coord))
    assert _decode(curve, "aff", msm_host(name, sc, pts, coord="aff")) == expect
    mont = cref.fr_to_mont(name, sc) if hasattr(cref, "fr_to_mont") else None
    if mont is not None:
        red = cref.fr_from_mont(name, mont)
        exp_fr = _aff(curve, cref.msm(name, red, pts, nthreads=NT)[0])
        assert _decode(curve, "jac", msm_host(name, mont, pts, coord="jac", fr_coefs=True)) == exp_fr
    # refusals come back as a status, r untouched
    L = _lib.lib()
    r = np.full(3 * CURVE_COORD_BYTES[name], 0xAB, dtype=np.uint8)
    for args in ((17, 0, 1), (0, 5, 1), (0, 0, 9)):
        assert L.ctt_hip_msm_host(args[0], args[1], args[2], r.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p),
                                  pts.ctypes.data_as(ctypes.c_void_p), n) == -1
    assert (r == 0xAB).all()


def test_c_program_through_the_header(tmp_path):
    """A plain C program including include/ctt_msm_hip.h, linked against libctt_msm_hip.so, gets the oracle's answer."""
    import subprocess
    from constantine_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "t_msm_c_abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_api", "t_msm_c_abi.c"), "-L", libdir, "-lctt_msm_hip",
                           f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    name = "bls12_381_g1"
    curve = po.CURVES[name]
    n = 1500
    pts = cref.gen_points(name, 77, n)
    sc = cref.synth_scalars(78, n, 255)
    import json
    doc = json.load(open(os.path.join(_golden.HERE, "eip2537_multiexp.json")))
    evm_name, evm_in, evm_exp = max(doc["g1"], key=lambda c: len(c[1]))          # the longest of the reference's G1MSM vectors
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.uint64(n).tobytes())
        f.write(sc.tobytes())
        f.write(pts.tobytes())
        f.write(np.uint64(len(evm_in) // 2).tobytes())
        f.write(bytes.fromhex(evm_in))
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    out = open(tmp_path / "out.bin", "rb").read()
    assert len(out) == 6 * 144 + n + 128
    expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0])
    assert curve.jac_from_bytes(out[:144]) == expect
    assert curve.prj_from_bytes(out[144:288]) == expect
    assert curve.jac_from_bytes(out[288:432]) == expect          # sharded over two contexts
    assert curve.jac_from_bytes(out[432:576]) == expect          # cached bases with a window table
    assert curve.jac_from_bytes(out[576:720]) == expect          # neutral typed symbol (what the Nim binding imports)
    assert curve.prj_from_bytes(out[720:864]) == expect          # neutral generic symbol
    assert out[864:864 + n] == b"\x01" * n                        # every generated point is in the subgroup
    assert out[864 + n:] == bytes.fromhex(evm_exp), evm_name       # the precompile symbol called from C


def test_zal_cached_base_descriptor_from_c(tmp_path):
    """The Halo2-ZAL caching hooks (get_base_descriptor / msm_with_cached_base / Drop; lib.rs:60-95, written out for Rust in
    INTEGRATION.md part D2) as a C program over ctt_hip_msm_bases_create_table / ctt_hip_msm_with_bases: ONE descriptor over
    BN254-Snarks G1 bases reused across 8 vectors of Montgomery Fr coefficients -- every result equals the oracle's, the
    un-cached ZAL entry's, and a prefix of the bases works."""
    import subprocess
    from constantine_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "t_zal_cached_base"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_api", "t_zal_cached_base.c"), "-L", libdir, "-lctt_msm_hip",
                           f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    name = "bn254_snarks_g1"
    curve = po.CURVES[name]
    n, m = 3000, 8
    pts = cref.gen_points(name, 4100, n)
    monts = [cref.synth_scalars(4200 + v, n, 250) for v in range(m)]       # Fr elements in Montgomery form, as the ZAL hands them over
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.uint64(n).tobytes())
        f.write(np.uint64(m).tobytes())
        f.write(pts.tobytes())
        for mont in monts:
            f.write(mont.tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    out = open(tmp_path / "out.bin", "rb").read()
    assert len(out) == (2 * m + 1) * 96 + 4
    for v, mont in enumerate(monts):
        expect = _aff(curve, cref.msm(name, cref.fr_from_mont(name, mont), pts, nthreads=NT)[0])
        assert curve.prj_from_bytes(out[96 * v:96 * v + 96]) == expect, v                       # msm_with_cached_base
        assert curve.prj_from_bytes(out[96 * (m + v):96 * (m + v) + 96]) == expect, v           # msm
    half = n // 2
    expect = _aff(curve, cref.msm(name, cref.fr_from_mont(name, monts[0][:half]), pts[:half], nthreads=NT)[0])
    assert curve.prj_from_bytes(out[96 * 2 * m:96 * 2 * m + 96]) == expect
    assert int.from_bytes(out[-4:], "little", signed=True) > 0                                # the descriptor is a window table


def test_concurrent_callers_overlap_on_their_own_contexts():
    """Round 6 (review item 8): the reference's `_parallel` MSM may be called from several threads at once, each with its own pool
    (include/constantine/core/threadpool.h:25-39; "can be nested", ec_multi_scalar_mul_parallel.nim:596).  Here every calling thread gets a
    context of its own (up to $CTT_HIP_HOST_CONTEXTS, default 4) instead of taking turns on one: a blocking 2^16-pair call keeps the chip
    busy for about a third of its 0.9 ms, so the chains of several callers overlap.  2 and 4 threads x 2^16 pairs through the Constantine
    symbol: every result against the oracle, and the aggregate rate against one thread's (measured on the builder's box: 1.62 x with two
    threads, 2.11 x with four -- profiles/concurrent_callers_r06.txt; the bars here leave room for a noisy box)."""
    import threading
    import time
    from constantine_amd import multiScalarMul_vartime_parallel
    name = "bls12_381_g1"
    curve = po.CURVES[name]
    n = 1 << 16
    pts = cref.gen_points(name, 2100, n)
    inputs, expect = [], []
    for t in range(4):
        sc = cref.synth_scalars(2200 + t, n, 255)
        inputs.append(sc)
        expect.append(_aff(curve, cref.msm(name, sc, pts, nthreads=NT)[0]))
    calls = 30
    old = os.environ.get("CTT_HIP_HOST_CONTEXTS")
    os.environ["CTT_HIP_HOST_CONTEXTS"] = "4"

    def run(T):
        got = [None] * T

        def work(t):
            for _ in range(calls):
                got[t] = multiScalarMul_vartime_parallel(None, name, inputs[t], pts, coord="jac")
        rate = 0.0
        for _ in range(2):          # (the first repetition opens the contexts and grows their workspaces)
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            rate = T * calls / (time.perf_counter() - t0)
        for t in range(T):
            assert _decode(curve, "jac", got[t]) == expect[t], (T, t)
        return rate
    try:
        r1, r2, r4 = run(1), run(2), run(4)
    finally:
        if old is None:
            os.environ.pop("CTT_HIP_HOST_CONTEXTS", None)
        else:
            os.environ["CTT_HIP_HOST_CONTEXTS"] = old
    print(f"concurrent callers, 2^16 pairs per call: {r1:.0f} / {r2:.0f} / {r4:.0f} calls/s with 1 / 2 / 4 threads ({r2 / r1:.2f} x, {r4 / r1:.2f} x)")
    assert r2 >= 1.3 * r1, (r1, r2)
    assert r4 >= 1.5 * r1, (r1, r4)


def test_concurrent_callers_are_serialised():
    """Several host threads calling the Constantine symbols at once (the reference allows MSM calls from inside
    pool tasks, ec_multi_scalar_mul_parallel.nim:596; KZG batch verification issues three at a time): the engine
    serialises them on its context and every caller gets its own answer."""
    import threading
    from constantine_amd import multiScalarMul_vartime_parallel
    jobs = []
    # (the two large ones go up in three or more slices: a helper thread copies while the caller's thread enqueues -- round 4)
    for i, (name, n) in enumerate([("bls12_381_g1", 3000), ("bn254_snarks_g1", 5000), ("pallas", 700),
                                   ("bls12_381_g1", 64), ("vesta", 2048), ("bn254_snarks_g1", 1),
                                   ("bls12_381_g1", (1 << 19) + 9), ("pallas", (1 << 20) + 1)]):
        curve = po.CURVES[name]
        pts = cref.gen_points(name, 1300 + i, n)
        sc = cref.synth_scalars(1400 + i, n, curve.scalar_bits)
        expect = _aff(curve, cref.msm(name, sc, pts, nthreads=NT if n > 100000 else 4)[0])
        if n > 100000:
            assert expect == cref.msm_by_discrete_logs(name, 1300 + i, sc)
        jobs.append((name, curve, sc, pts, expect))
    results = [None] * len(jobs)

    def work(k):
        name, curve, sc, pts, _ = jobs[k]
        for _ in range(3):
            results[k] = curve.jac_from_bytes(bytes(multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac")))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k, job in enumerate(jobs):
        assert results[k] == job[4], job[0]
