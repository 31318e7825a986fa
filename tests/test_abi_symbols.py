"""The C-ABI library loads and exports every symbol include/ctt_msm_hip.h declares (no GPU, no compute)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    """Expand the declaration macros of the header with the C preprocessor and pull the function names."""
    out = subprocess.check_output(["gcc", "-E", "-P", os.path.join(ROOT, "include", "ctt_msm_hip.h")], text=True)
    return sorted(set(re.findall(r"\b(ctt_[a-z0-9_]+)\s*\(", out)))


def test_header_compiles_as_c_and_layouts():
    src = r'''
    #include "ctt_msm_hip.h"
    _Static_assert(sizeof(big255) == 32 && sizeof(big254) == 32, "scalar");
    _Static_assert(sizeof(bls12_381_g1_aff) == 96 && sizeof(bls12_381_g1_jac) == 144, "bls g1");
    _Static_assert(sizeof(bls12_381_g2_aff) == 192 && sizeof(bls12_381_g2_prj) == 288, "bls g2");
    _Static_assert(sizeof(bn254_snarks_g1_aff) == 64 && sizeof(bn254_snarks_g1_prj) == 96, "bn g1");
    _Static_assert(sizeof(bn254_snarks_g2_aff) == 128, "bn g2");
    _Static_assert(sizeof(pallas_ec_aff) == 64 && sizeof(vesta_ec_jac) == 96, "pasta");
    int main(void) { return 0; }
    '''
    p = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-",
                        "-fsyntax-only"], input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr


def test_library_exports_every_declared_symbol():
    from constantine_amd import _lib
    declared = _declared_symbols()
    assert len([s for s in declared if "multi_scalar_mul" in s]) == 40
    assert sorted(_lib.exported_symbols()) == declared
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} not built; run __graft_entry__.build()")
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert len([s for s in declared if s.endswith('_batch_affine') and 'hip' not in s]) == 12
    assert len([s for s in declared if re.fullmatch(r"ctt_hip_msm_[a-z0-9_]+_(jac|prj)_(big|fr)", s)]) == 24   # neutral set
    assert L.ctt_hip_msm_abi_version() == _lib.ABI_VERSION


def test_library_exports_nothing_but_the_c_abi():
    """the .so is built with -fvisibility=hidden; `nm -D --defined-only` shows the declared ctt_* symbols and
    nothing else (no C++ symbols of the engine, no device stubs, no per-curve operation tables)."""
    from constantine_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    names = sorted({line.split()[-1] for line in out.splitlines() if line.strip()})
    declared = set(_declared_symbols())
    extra = [s for s in names if s not in declared]
    assert not extra, extra[:20]
    assert set(names) == declared


def test_host_only_point_sum_matches_oracle():
    """ctt_hip_ec_sum_affine is host code inside the product library (no GPU needed)."""
    from constantine_amd.msm import ec_sum_affine
    from oracle import cref
    from oracle import pyoracle as po
    for name in ("bls12_381_g1", "bn254_snarks_g1", "bls12_381_g2"):
        curve = po.CURVES[name]
        pts = cref.gen_points(name, 5, 4)
        expect = None
        for p in pts:
            expect = curve.add(expect, curve.aff_from_bytes(bytes(p)))
        assert curve.aff_from_bytes(bytes(ec_sum_affine(name, pts))) == expect
        assert curve.jac_from_bytes(bytes(ec_sum_affine(name, pts, coord="jac"))) == expect
        assert curve.prj_from_bytes(bytes(ec_sum_affine(name, pts, coord="prj"))) == expect
        # P + (-P) = neutral
        P = curve.aff_from_bytes(bytes(pts[0]))
        two = curve.points_to_array([P, curve.neg(P)])
        assert curve.aff_from_bytes(bytes(ec_sum_affine(name, two))) is None


def test_host_mirror_argument_checks():
    from constantine_amd import msm
    with pytest.raises(ValueError):
        msm.multiScalarMul_vartime("bls12_381_g1", np.zeros((2, 32), np.uint8), np.zeros((3, 96), np.uint8))
    with pytest.raises(ValueError):
        msm.multiScalarMul_vartime("bls12_381_g1", np.zeros((2, 31), np.uint8), np.zeros((2, 96), np.uint8))
    with pytest.raises(AttributeError):
        msm.multiScalarMul_vartime_parallel(None, "bls12_381_g2", np.zeros((1, 32), np.uint8), np.zeros((1, 192), np.uint8))
    with pytest.raises(AssertionError):
        msm.CttEngine().msm(np.zeros((2, 32), np.uint8), np.zeros((1, 64), np.uint8))


def test_subgroup_check_of_a_few_host_points_runs_on_the_host():
    """ctt_hip_subgroup_check with <= 64 host-resident points (256 for BLS12-381) never touches the GPU (one GPU lane needs 6.5 ms for [r]P): the generic
    [r]P = neutral on the host for every curve, and for BLS12-381 the reference's endomorphism tests phi(P) = [-x^2]P for G1 and
    psi(P) = [x]P for G2 (bls12_381_subgroups.nim:170-207).  Against the big-integer oracle on subgroup points, curve points outside the subgroup, a point
    of order 3 and the neutral."""
    import random
    from constantine_amd.msm import subgroup_check
    from oracle import cref
    from oracle import pyoracle as po
    rng = random.Random(17)
    for name in ("bls12_381_g1", "bn254_snarks_g1", "bls12_381_g2"):
        curve = po.CURVES[name]
        F = curve.F
        inside = [curve.aff_from_bytes(bytes(b)) for b in cref.gen_points(name, 9, 6)]
        outside = []
        if name == "bls12_381_g1":
            p = F.p
            while len(outside) < 6:            # random points of the curve: the cofactor is ~2^126, so almost never in the subgroup
                x = rng.randrange(p)
                y2 = (x * x * x + 4) % p
                y = pow(y2, (p + 1) // 4, p)
                if y * y % p == y2:
                    outside.append((x, y))
            outside.append((0, 2))             # x = 0: a point of order 3
        if name == "bls12_381_g2":
            # a point of the twist outside G2: the reference's EIP-2537 failure vector, its multiples, and its sums with G2 points
            import json
            from tests import _golden
            doc = json.load(open(os.path.join(_golden.HERE, "eip2537_multiexp.json")))
            raw = bytes.fromhex(next(inp for _, inp, err in doc["g2_fail"] if "subgroup" in err))[:256]
            c = [int.from_bytes(raw[64 * i:64 * i + 64], "big") for i in range(4)]
            Q = ((c[0], c[1]), (c[2], c[3]))
            assert curve.is_on_curve(Q)
            outside = [Q, curve.scalar_mul(2, Q), curve.scalar_mul(5, Q), curve.add(Q, inside[0]), curve.add(curve.scalar_mul(7, Q), inside[1])]
        pts = inside + outside + [None]
        want = [curve.scalar_mul(curve.order, P) is None for P in pts]
        assert want[:6] == [True] * 6 and want[-1] is True
        if outside:
            assert not any(want[6:-1])
        got = subgroup_check(name, curve.points_to_array(pts))
        assert [bool(v) for v in got] == want, name
        # the 16-thread split
        many = curve.points_to_array((pts * 5)[:60])
        assert [bool(v) for v in subgroup_check(name, many)] == (want * 5)[:60], name


def test_without_a_device_calls_are_refused_not_fatal():
    """Round 5: no symbol with a return value terminates the process.  On a box without a HIP device -- this
    container -- the neutral MSM symbol returns -1, the protocol symbols return CTT_HIP_STATUS_GPU_UNAVAILABLE (0xF0, outside the
    reference's status enums), outputs stay untouched and ctt_hip_last_error() says -3 (no usable device).  Host-only work that
    comes first still yields the reference's statuses.  (The Constantine-named void MSM symbols have no channel and abort.)"""
    from constantine_amd import _lib, evm, kzg
    from oracle import pyoracle as po
    L = _lib.lib()
    if L.ctt_hip_msm_available() == 1:
        pytest.skip("a HIP device is present: the refusal path of a device-less box cannot be shown here")
    curve = po.CURVES["bls12_381_g1"]
    G = curve.gen
    pts = curve.points_to_array([G, G])
    sc = curve.scalars_to_array([3, 4])
    r = (ctypes.c_uint8 * 144)(*([0xAB] * 144))
    L.ctt_hip_clear_last_error()
    assert L.ctt_hip_last_error() == 0
    rc = L.ctt_hip_msm_host(0, 0, 1, r, sc.ctypes.data_as(ctypes.c_void_p), pts.ctypes.data_as(ctypes.c_void_p), 2)
    assert rc == -1 and L.ctt_hip_last_error() == -3 and b"no HIP device" in L.ctt_hip_last_error_message()
    assert bytes(r) == bytes([0xAB]) * 144                       # r untouched
    assert L.ctt_hip_msm_ctx_create(0) is None
    # EIP-2537: a well-formed pair is refused with the sentinel; a malformed input still gets the reference's status first
    x, y = G
    pair = x.to_bytes(64, "big") + y.to_bytes(64, "big") + (5).to_bytes(32, "big")
    with pytest.raises(_lib.GpuUnavailable) as e:
        evm.eth_evm_bls12381_g1msm(pair)
    assert e.value.code == -3
    with pytest.raises(evm.EvmError, match="cttEVM_InvalidInputSize"):
        evm.eth_evm_bls12381_g1msm(pair[:-1])
    with pytest.raises(evm.EvmError, match="cttEVM_PointNotOnCurve"):
        evm.eth_evm_bls12381_g1msm(x.to_bytes(64, "big") + (y + 1).to_bytes(64, "big") + (5).to_bytes(32, "big"))
    # EIP-4844: the context cannot be built without a device -- reported, not fatal; a malformed SRS is still cttEthTS_InvalidFile
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "kzg4844_srs_g1_lagrange.bin"), "rb").read()
    with pytest.raises(_lib.GpuUnavailable):
        kzg.EthereumKZGContext(raw)
    with pytest.raises(ValueError, match="cttEthTS_InvalidFile"):
        kzg.EthereumKZGContext(bytes(48) + raw[48:])             # an all-zero line: no compression flag
