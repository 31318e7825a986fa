"""
Logic check of the HIP pipeline on the CPU: tests/emu runs the SAME per-thread bodies and host
orchestration as the GPU engine (constantine_amd/csrc/msm_bodies.h, msm_pipeline.h) with kernel launches
replaced by loops, and is compared with the oracle.  This is not the product path (which is GPU-only);
the GPU parity tests proper are tests/test_gpu_parity.py.
"""
import random

import numpy as np
import pytest

from oracle import cref
from oracle import pyoracle as po
from tests import _golden
from tests.emu import emu

ALL = list(po.CURVES)
G1S = ["bls12_381_g1", "bn254_snarks_g1", "pallas", "vesta"]


def _aff(curve, b):
    return curve.aff_from_bytes(bytes(b))


@pytest.mark.parametrize("name", ALL)
def test_field_ops_vs_python(name):
    curve = po.CURVES[name]
    F = curve.F
    rng = random.Random(5)
    base = F if F.degree == 1 else F.base
    p = base.p

    def rnd():
        pool = [0, 1, p - 1, p - 2, 2, (1 << (p.bit_length() - 1)), rng.randrange(p), rng.randrange(p), rng.randrange(p)]
        v = rng.choice(pool)
        return v if F.degree == 1 else (v, rng.choice(pool))

    def enc(v):
        return np.frombuffer(F.to_mont_bytes(v), dtype=np.uint8)

    for _ in range(60):
        a, b = rnd(), rnd()
        assert F.from_mont_bytes(bytes(emu.field_op(name, 0, enc(a), enc(b)))) == F.mul(a, b)
        assert F.from_mont_bytes(bytes(emu.field_op(name, 1, enc(a)))) == F.sqr(a)
        assert F.from_mont_bytes(bytes(emu.field_op(name, 2, enc(a), enc(b)))) == F.add(a, b)
        assert F.from_mont_bytes(bytes(emu.field_op(name, 3, enc(a), enc(b)))) == F.sub(a, b)
        assert F.from_mont_bytes(bytes(emu.field_op(name, 4, enc(a)))) == F.neg(a)
    # inversion by division steps (modinv.h) against Python, and against a^(p-2) (op 6, rounds 1-2's inversion); inv(0) = 0
    specials = [1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << (p.bit_length() - 1), base.R % p, pow(base.R, -1, p)]
    for v in specials + [rng.randrange(1, p) for _ in range(60)]:
        a = v if F.degree == 1 else (v, rng.choice([0, 1, p - 1, rng.randrange(p)]))
        got = F.from_mont_bytes(bytes(emu.field_op(name, 5, enc(a))))
        assert got == F.inv(a), (name, a)
        assert F.from_mont_bytes(bytes(emu.field_op(name, 6, enc(a)))) == got
    zero = F.from_int(0)
    assert F.is_zero(F.from_mont_bytes(bytes(emu.field_op(name, 5, enc(zero)))))


@pytest.mark.parametrize("name", ALL)
def test_carry_free_device_field_vs_python(name):
    """fpu.h: values are x*R' mod p in LB-bit limbs, only bounded by a small multiple of p (per Fp2 component)."""
    info = emu.dev_field_info(name)
    if info is None:
        pytest.skip("curve computes in the canonical field")
    lb, nl_total = info
    curve = po.CURVES[name]
    F = curve.F
    deg = F.degree
    nl = nl_total // deg
    base = F if deg == 1 else F.base
    p = base.p
    Rp = 1 << (lb * nl)
    rng = random.Random(9)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 1 << (p.bit_length() - 1), base.R % p, Rp % p]

    def rnd():
        c = [rng.choice(edge) if rng.random() < 0.2 else rng.randrange(p) for _ in range(deg)]
        return c[0] if deg == 1 else tuple(c)

    def enc(v):
        return np.frombuffer(F.to_mont_bytes(v), dtype=np.uint8)

    def dec(limbs, bound):
        comps = []
        for k in range(deg):
            part = limbs[k * nl:(k + 1) * nl]
            assert all(int(x) < (1 << lb) for x in part[:-1]), "limbs must come back normalised"
            v = sum(int(x) << (lb * i) for i, x in enumerate(part))
            assert v < bound * p, "value bound"
            comps.append(v * pow(Rp, -1, p) % p)
        return comps[0] if deg == 1 else tuple(comps)

    for _ in range(200 if deg == 1 else 60):
        a, b = rnd(), rnd()
        for op, want, bound in ((0, F.mul(a, b), 2), (1, F.sqr(a), 2), (2, F.add(a, b), 4), (3, F.sub(a, b), 4), (4, a, 2),
                                 (5, F.mul(a, b), 2), (6, F.sqr(a), 2), (7, F.sub(F.mul(a, b), F.sqr(a)), 4)):
            # 5-7: operands at the largest bounds, and in the lazy forms, that ec.h feeds into products
            assert dec(emu.field_op_dev(name, op, enc(a), enc(b)), bound) == want, (name, op, a, b)
    # raw zero in -> raw zero out (neutral flags rely on it)
    z = F.from_int(0)
    assert not emu.field_op_dev(name, 4, enc(z)).any()
    assert not emu.field_op_dev(name, 0, enc(z), enc(rnd())).any()


@pytest.mark.parametrize("name", ALL)
def test_gen_points_same_definition(name):
    a = emu.gen_points(name, 77, 5, first=2)
    b = cref.gen_points(name, 77, 5, first=2)
    assert bytes(a.tobytes()) == bytes(b.tobytes())


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 64, 300])
def test_msm_vs_oracle(name, n):
    curve = po.CURVES[name]
    if curve.F.degree == 2 and n > 64:
        pytest.skip("kept small for Fp2")
    pts = cref.gen_points(name, 3, n)
    sc = cref.synth_scalars(4, n, curve.scalar_bits)
    expect, _ = cref.msm(name, sc, pts)
    # K < 0: the emulator harness routes the call through prepare_bases + submit(prepared) (cached-base path)
    for kw in (dict(), dict(c=3, K=4), dict(c=5, K=8), dict(c=7, K=4, S=1), dict(c=6, K=-8)):
        out, plan = emu.msm(name, sc, pts, **kw)
        assert bytes(out) == bytes(expect), (kw, plan)


@pytest.mark.parametrize("name", G1S)
def test_msm_medium_default_plan(name):
    curve = po.CURVES[name]
    n = 5000
    pts = cref.gen_points(name, 13, n)
    sc = cref.synth_scalars(14, n, curve.scalar_bits)
    expect, _ = cref.msm(name, sc, pts, nthreads=4)
    out, plan = emu.msm(name, sc, pts)
    assert bytes(out) == bytes(expect), plan
    # and without the port: the points are [s_i]G with known s_i (cref.msm_by_discrete_logs, the big-integer oracle only)
    assert curve.aff_from_bytes(bytes(out)) == cref.msm_by_discrete_logs(name, 13, sc)
    out, plan = emu.msm(name, sc, pts, S=3, K=12)
    assert bytes(out) == bytes(expect), plan


def test_window_sizes_including_divisors_of_bits():
    """c | bits needs the extra top window (ec_multi_scalar_mul.nim:278-289); 255 = 3*5*17, 254 = 2*127."""
    for name, cs in (("bls12_381_g1", (3, 5, 15, 16)), ("bn254_snarks_g1", (2, 13, 16))):
        curve = po.CURVES[name]
        n = 200
        pts = cref.gen_points(name, 31, n)
        sc = cref.synth_scalars(32, n, curve.scalar_bits)
        # force the top bits on for some scalars
        sc[:20, 31] |= 0x7F if curve.scalar_bits == 255 else 0x3F
        sc[:20, 24:31] = 0xFF
        expect, _ = cref.msm(name, sc, pts)
        for c in cs:
            out, plan = emu.msm(name, sc, pts, c=c, K=8)
            assert bytes(out) == bytes(expect), (name, c)


def test_all_equal_scalars_long_chains():
    """Every point lands in the same bucket per window: exercises tail/head chains and the merge tree."""
    name = "bls12_381_g1"
    curve = po.CURVES[name]
    n = 700
    pts = cref.gen_points(name, 41, n)
    sc = np.tile(cref.synth_scalars(42, 1, 255), (n, 1))
    expect, _ = cref.msm(name, sc, pts)
    for K in (4, 8, 28, 64):
        out, plan = emu.msm(name, sc, pts, c=6, K=K)
        assert bytes(out) == bytes(expect), K


def test_all_equal_points_and_scalars_doubling_paths():
    """bug-366 style (t_ec_shortw_jac_g2_msm_bug_366.nim): all points equal -> P == Q adds everywhere."""
    name = "bn254_snarks_g1"
    curve = po.CURVES[name]
    n = 257
    pts = np.tile(cref.gen_points(name, 51, 1), (n, 1))
    sc = cref.synth_scalars(52, n, 254)
    expect, _ = cref.msm(name, sc, pts)
    out, _ = emu.msm(name, sc, pts, c=4, K=4)
    assert bytes(out) == bytes(expect)
    sc2 = np.tile(sc[:1], (n, 1))
    expect, _ = cref.msm(name, sc2, pts)
    out, _ = emu.msm(name, sc2, pts, c=5, K=8)
    assert bytes(out) == bytes(expect)


def test_infinity_inputs_cancellation_and_zero_scalars():
    name = "bls12_381_g1"
    curve = po.CURVES[name]
    G = curve.gen
    pts = curve.points_to_array([G, None, G, curve.neg(G), G, None])
    sc = curve.scalars_to_array([5, 77, 5, 3, 0, 0])
    out, _ = emu.msm(name, sc, pts, c=3, K=4)
    assert _aff(curve, out) == curve.scalar_mul(7, G)
    pts = curve.points_to_array([G, curve.neg(G)])
    sc = curve.scalars_to_array([9, 9])
    out, _ = emu.msm(name, sc, pts, out_kind=0)
    assert _aff(curve, out) is None
    jac, _ = emu.msm(name, sc, pts, out_kind=1)
    assert curve.jac_from_bytes(bytes(jac)) is None
    prj, _ = emu.msm(name, sc, pts, out_kind=2)
    assert curve.prj_from_bytes(bytes(prj)) is None
    # all-zero scalars, n == 0
    out, _ = emu.msm(name, curve.scalars_to_array([0, 0, 0]), curve.points_to_array([G, G, G]))
    assert _aff(curve, out) is None
    out, _ = emu.msm(name, np.zeros((0, 32), np.uint8), np.zeros((0, 96), np.uint8))
    assert _aff(curve, out) is None


@pytest.mark.parametrize("group,cname", [("g1", "bls12_381_g1"), ("g2", "bls12_381_g2")])
def test_eip2537_golden(group, cname):
    curve = po.CURVES[cname]
    for name, scalars, points, expected in _golden.eip2537(group):
        sc = curve.scalars_to_array([k % curve.order for k in scalars])
        pts = curve.points_to_array(points)
        for ok, dec in ((1, curve.jac_from_bytes), (2, curve.prj_from_bytes)):
            out, _ = emu.msm(cname, sc, pts, out_kind=ok)
            assert dec(bytes(out)) == expected, name


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "pallas"])
def test_fr_coefs_entry(name):
    curve = po.CURVES[name]
    n = 50
    pts = cref.gen_points(name, 61, n)
    ks = [po.synth_scalar(62, i, 256) % curve.Fr.p for i in range(n)]
    can = curve.scalars_to_array(ks)
    mont = curve.fr_scalars_to_array(ks)
    expect, _ = cref.msm(name, can, pts)
    out, _ = emu.msm(name, mont, pts, coef_is_fr=True, c=4)
    assert bytes(out) == bytes(expect)


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_snarks_g1", "bls12_381_g2"])
def test_host_pointer_form_uploads_in_slices(name):
    """MsmEngine::submit_host: the pairs arrive in slices, every slice is sorted on its own and accumulated INTO the one bucket set
    (accum_body<F, INTO = true>: a run continues the sum the earlier slices stored) -- same element for any number of slices (1, 2,
    3, 8; ragged slice sizes; slices that leave whole buckets empty; a pair that meets its own copy, and one that meets its
    negative, in a later slice: the stored sum is doubled / becomes the neutral; Fr Montgomery coefficients)."""
    curve = po.CURVES[name]
    n = 1201 if curve.F.degree == 1 else 150
    pts = cref.gen_points(name, 501, n)
    sc = cref.synth_scalars(502, n, curve.scalar_bits)
    sc[:40] = sc[0]                      # one heavy bucket per window that lives in the first slice only
    pts[7] = 0                           # a neutral point
    # the same (scalar, point) again in the last slice, and a (scalar, -point): every window's bucket of pair 45 is re-entered with the
    # stored P and doubled; pair 46's holds P, then P - P
    sc[n - 3], pts[n - 3] = sc[45], pts[45]
    sc[n - 2] = sc[46]
    pts[n - 2] = curve.points_to_array([curve.neg(curve.aff_from_bytes(bytes(pts[46])))])[0]
    expect, _ = cref.msm(name, sc, pts)
    for chunks, c in ((1, 0), (2, 0), (3, 5), (8, 0), (2, 11)):
        out, used = emu.msm_host(name, sc, pts, c=c, chunks=chunks)
        assert bytes(out) == bytes(expect), (name, chunks, c)
        assert used == chunks
    mont = cref.synth_scalars(503, n, 250)
    expect, _ = cref.msm(name, cref.fr_from_mont(name, mont), pts)
    out, _ = emu.msm_host(name, mont, pts, coef_is_fr=True, chunks=3)
    assert bytes(out) == bytes(expect)


def test_head_chains_of_every_length_class():
    """Buckets that span several accumulate lanes leave a chain of partial sums (heads) that the merge tree sums in
    log2(chain) steps: chains of 3 ... 250 heads next to ordinary buckets."""
    name = "bn254_snarks_g1"
    K = 4
    rng = np.random.default_rng(5)
    for n_equal in (9, 130, 255, 258, 262, 1000):       # chains of 3, 33, 64, 65, 66, 250 heads
        n = n_equal + 300
        pts = cref.gen_points(name, 700 + n_equal, n)
        sc = cref.synth_scalars(701 + n_equal, n, 254)
        idx = rng.permutation(n)[:n_equal]
        sc[idx] = sc[idx[0]]                              # n_equal pairs share every bucket; the rest is spread out
        expect, _ = cref.msm(name, sc, pts)
        for c in (5, 9):
            out, plan = emu.msm(name, sc, pts, c=c, K=K)
            assert bytes(out) == bytes(expect), (n_equal, c)


def test_head_merge_chain_form_and_tree_agree(monkeypatch):
    """Round 5: the head merge has a chain form (one lane adds up the heads of its bucket; chains above `lmax` heads go to the
    one-workgroup tree, merge_long_body) next to the launch-per-level tree.  Both forms, every lmax class (every chain long,
    some long, none long), on inputs whose chains hold 1 ... 250 heads, a quarter-equal and an all-equal input, G1 and G2."""
    K = 4
    rng = np.random.default_rng(11)
    cases = []
    name = "bn254_snarks_g1"
    for n_equal in (0, 9, 40, 258):
        n = n_equal + 200
        pts = cref.gen_points(name, 1700 + n_equal, n)
        sc = cref.synth_scalars(1701 + n_equal, n, 254)
        if n_equal:
            idx = rng.permutation(n)[:n_equal]
            sc[idx] = sc[idx[0]]
        cases.append((name, sc, pts))
    pts = cref.gen_points(name, 1801, 300)
    cases.append((name, np.tile(cref.synth_scalars(1802, 1, 254), (300, 1)), pts))     # all equal: one chain per window
    g2 = "bls12_381_g2"
    pts2 = cref.gen_points(g2, 1901, 90)
    sc2 = cref.synth_scalars(1902, 90, 255)
    sc2[:40] = sc2[0]
    cases.append((g2, sc2, pts2))
    for cname, sc, pts in cases:
        expect, _ = cref.msm(cname, sc, pts)
        for mode, lmax in ((2, 0), (1, 1), (1, 2), (1, 3), (1, 8), (1, 1000), (0, 0)):
            monkeypatch.setenv("EMU_MERGE_CHAIN", str(mode))
            monkeypatch.setenv("EMU_MERGE_LMAX", str(lmax))
            for c in (5, 9):
                out, _ = emu.msm(cname, sc, pts, c=c, K=K)
                assert bytes(out) == bytes(expect), (cname, len(sc), mode, lmax, c)
            # the host-pointer form merges once per slice, the later slices into the stored sums
            out, _ = emu.msm_host(cname, sc, pts, chunks=3)
            assert bytes(out) == bytes(expect), (cname, len(sc), mode, lmax, "slices")


@pytest.mark.parametrize("name", ALL)
def test_window_table_for_cached_bases(name):
    """MsmEngine::prepare_table + the window-table plan: T[w][j] = 2^(c*w) * P_j, every digit window selects a table row,
    all windows share one bucket set and there is no window combine.  Same element as the oracle for every c (including
    divisors of the scalar width: the extra window), for a prefix of the cached bases, with neutral points among the
    bases, equal points and scalars (doubling paths inside the shared buckets) and Fr Montgomery coefficients."""
    curve = po.CURVES[name]
    n = 120 if curve.F.degree == 1 else 24
    pts = cref.gen_points(name, 801, n)
    sc = cref.synth_scalars(802, n, curve.scalar_bits)
    pts[5] = 0                          # a neutral base: every table row of it stays neutral
    pts[9] = pts[8]
    sc[9] = sc[8]                       # the same pair twice: the shared bucket doubles
    expect, _ = cref.msm(name, sc, pts)
    for c, K in ((0, 0), (3, 4), (5, 8), (15, 4)) if name == "bls12_381_g1" else ((0, 0), (5, 4)):
        out, cu = emu.msm_table(name, sc, pts, c=c, K=K)
        assert cu == c or c == 0
        assert cu > 0
        assert bytes(out) == bytes(expect), (name, c, cu)
    # host-resident coefficients over the cached bases, uploaded in slices (MsmEngine::submit_host with d_prepared; round 4): a slice
    # addresses its column of every row block of the table (row w * ntab + j), and its records when the bases are plain (c = -1)
    for c, chunks in ((0, 1), (5, 2), (6, 3), (-1, 1), (-1, 3)):
        out, cu = emu.msm_table(name, sc, pts, c=c, K=4, chunks=chunks)
        assert bytes(out) == bytes(expect), (name, c, chunks)
        assert (cu == 0) == (c < 0)
    # a prefix of the cached bases (table rows stay ntab apart)
    m = n // 3
    expect, _ = cref.msm(name, sc[:m], pts[:m])
    out, _ = emu.msm_table(name, sc[:m], pts, c=6, K=4)
    assert bytes(out) == bytes(expect)
    out, _ = emu.msm_table(name, sc[:m], pts, c=6, K=4, chunks=2)
    assert bytes(out) == bytes(expect)
    if curve.F.degree == 1:
        mont = cref.synth_scalars(803, n, 250)
        expect, _ = cref.msm(name, cref.fr_from_mont(name, mont), pts)
        out, _ = emu.msm_table(name, mont, pts, coef_is_fr=True, c=7)
        assert bytes(out) == bytes(expect)
        # all-zero scalars -> neutral
        out, _ = emu.msm_table(name, np.zeros((n, 32), np.uint8), pts, c=4)
        assert _aff(curve, out) is None


def test_horner_groups(monkeypatch):
    """The bit Horner cut into groups of 1, 2, 3, 4, 7 bits or a single group per window (the legacy host_window_sums
    spellings included): the device returns ngrp partial sums per window, the host joins them -- the same element whatever
    the group size."""
    name = "bn254_snarks_g1"
    n = 400
    pts = cref.gen_points(name, 901, n)
    sc = cref.synth_scalars(902, n, 254)
    expect, _ = cref.msm(name, sc, pts)
    for c in (2, 4, 9, 10, 11, 12, 14):
        for hb in (0, 1, 2, 3, 7, 30):
            monkeypatch.setenv("EMU_HORNER_BITS", str(hb))
            out, plan = emu.msm(name, sc, pts, c=c, K=8)
            assert bytes(out) == bytes(expect), (c, hb)
    monkeypatch.delenv("EMU_HORNER_BITS")
    for hws in (1, 2):
        monkeypatch.setenv("EMU_HOST_WINDOW_SUMS", str(hws))
        out, _ = emu.msm(name, sc, pts, c=11, K=8)
        assert bytes(out) == bytes(expect), hws


def test_tickets_finished_out_of_order():
    """Three slots per engine (round 6; two before): with ticket A outstanding, blocking calls B and C must both be served (a free
    slot is taken whichever it is), three tickets may be outstanding together, a fourth is refused, and A still finishes correctly
    afterwards."""
    name = "bls12_381_g1"
    n = 90
    pts = cref.gen_points(name, 911, n)
    sc = cref.synth_scalars(912, n, 255)
    expect, _ = cref.msm(name, sc, pts)
    out, refused = emu.msm_slots(name, sc, pts)
    assert refused == 0
    for i in range(3):
        assert bytes(out[i]) == bytes(expect), i


def test_window_table_falls_back_when_memory_or_window_bits_do_not_fit(monkeypatch):
    """a window table that does not fit the device comes back as "no table" (plain records, same result) instead of
    an abort, and a caller's window_bits outside 2..22 is replaced by the automatic choice."""
    name = "pallas"
    curve = po.CURVES[name]
    n = 64
    pts = cref.gen_points(name, 921, n)
    sc = cref.synth_scalars(922, n, 255)
    expect, _ = cref.msm(name, sc, pts)
    for c in (1, 23, 31, 40):
        out, cu = emu.msm_table(name, sc, pts, c=c)
        assert 4 <= cu <= 22 and bytes(out) == bytes(expect), (c, cu)
    n = 2000
    pts = cref.gen_points(name, 923, n)
    sc = cref.synth_scalars(924, n, 255)
    expect, _ = cref.msm(name, sc, pts)
    monkeypatch.setenv("EMU_ALLOC_LIMIT", str(3 << 20))      # the table (>= 12 levels of n records: > 3 MiB) does not fit, the rest does
    out, cu = emu.msm_table(name, sc, pts, c=0)
    assert cu == 0 and bytes(out) == bytes(expect)


def test_plan_fits_the_gpu_for_any_size():
    """The plan's roundings (msm_pipeline.h): the accumulate grid -- W rows of ceil(G/64) one-wave workgroups -- never exceeds
    the resident wave slots (one workgroup more means a second round of a single wave: measured +28 % on BN254 2^22 at c = 15),
    the partition blocks are at most 512 (2048 from 2^23 pairs on) + a rounding remainder and all of one size, every entry has a lane; for sizes that are
    and are not powers of two, both scalar widths, both occupancies, and the window-table form."""
    rng = random.Random(99)
    sizes = [1, 2, 63, 64, 65, 1000, 4096, 65536, 65537, 100000, (1 << 17) + 777, 3 << 16, 1000003, 1 << 20, (1 << 20) + 12345,
             (1 << 22) + 77777, 1 << 24, (1 << 24) + 1, 5 << 22] + [rng.randrange(1, 1 << 25) for _ in range(200)]
    for n in sizes:
        for bits, lanes in ((255, 131072), (254, 262144), (255, 65536)):
            p = emu.plan(n, bits, lanes)
            # balanced windows over bits + 1 bits: r of cb + 1 bits, the others cb; c is the widest
            assert p["cb"] * p["W"] + p["r"] == bits + 1 and 0 <= p["r"] < p["W"] and p["c"] == p["cb"] + (1 if p["r"] else 0)
            assert p["Wd"] == p["W"] and p["nent"] == n and p["B"] == 1 << (p["c"] - 1)
            assert p["G"] == -(-n // p["K"]) and p["K"] >= 4
            assert p["W"] * -(-p["G"] // 64) <= max(lanes // 64, p["W"]), (n, bits, lanes, p)
            assert p["S"] == -(-n // p["slice"]) and p["S"] <= (2080 if n >= 1 << 23 else 520), (n, p)
            assert p["NG"] <= 4096 and (p["B"] >> p["gshift"]) == p["NG"] and p["B"] // p["NG"] <= 1024
            c = emu.table_window_bits(n, bits)
            assert 4 <= c <= 22
            t = emu.plan(n, bits, lanes, table_c=c, ntab=n)
            assert t["W"] == 1 and t["Wd"] == -(-(bits + 1) // c) and t["nent"] == t["Wd"] * n
            assert t["cb"] * t["Wd"] + t["r"] == bits + 1 and t["c"] <= c
            assert -(-t["G"] // 64) <= max(lanes // 64, 1), (n, bits, lanes, t)
            assert t["G"] == -(-t["nent"] // t["K"])
            assert t["NG"] <= 16384 and t["B"] // t["NG"] <= 1024
