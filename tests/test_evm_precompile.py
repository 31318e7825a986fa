"""EIP-2537 G1MSM / G2MSM: the reference's C symbols ctt_eth_evm_bls12381_g1msm / _g2msm (include/ctt_msm_hip.h part 3,
csrc/protocols.hip; constantine_amd/evm.py is their ctypes caller) against the reference's pass and fail vectors
(tests/protocol_ethereum_evm_precompiles/eip-2537/{,fail-}multiexp_G{1,2}_bls.json)."""
import json
import os

import pytest

from tests import _golden

DOC = json.load(open(os.path.join(_golden.HERE, "eip2537_multiexp.json")))
HOST_ONLY_ERRORS = {
    "invalid input length": "cttEVM_InvalidInputSize",
    "invalid fp.Element encoding": "cttEVM_IntLargerThanModulus",
    "invalid field element top bytes": "cttEVM_IntLargerThanModulus",
    "invalid point: not on curve": "cttEVM_PointNotOnCurve",
}


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_malformed_inputs_are_rejected_on_the_host(group):
    """Length, field-element and on-curve failures are decided before anything reaches the GPU."""
    from constantine_amd import evm
    fn = evm.eth_evm_bls12381_g1msm if group == "g1" else evm.eth_evm_bls12381_g2msm
    seen = 0
    for name, inp, err in DOC[group + "_fail"]:
        if err not in HOST_ONLY_ERRORS:
            continue
        seen += 1
        with pytest.raises(evm.EvmError) as e:
            fn(bytes.fromhex(inp))
        assert e.value.status.name == HOST_ONLY_ERRORS[err], name
    assert seen == 6
    # the output buffer's length is part of the contract (ethereum_evm_precompiles.nim:925-929): checked after the input length
    with pytest.raises(evm.EvmError) as e:
        fn(bytes(160 if group == "g1" else 288), r_len=64)
    assert e.value.status == evm.CttEVMStatus.cttEVM_InvalidOutputSize
    with pytest.raises(evm.EvmError) as e:
        fn(b"", r_len=64)
    assert e.value.status == evm.CttEVMStatus.cttEVM_InvalidInputSize


@pytest.mark.gpu
@pytest.mark.parametrize("group", ["g1", "g2"])
def test_precompile_vectors_on_gpu(group):
    from constantine_amd import evm
    fn = evm.eth_evm_bls12381_g1msm if group == "g1" else evm.eth_evm_bls12381_g2msm
    for name, inp, exp in DOC[group]:
        assert fn(bytes.fromhex(inp)) == bytes.fromhex(exp), name
    for name, inp, err in DOC[group + "_fail"]:
        with pytest.raises(evm.EvmError) as e:
            fn(bytes.fromhex(inp))
        if "subgroup" in err:
            assert e.value.status == evm.CttEVMStatus.cttEVM_PointNotInSubgroup, name


@pytest.mark.gpu
def test_status_is_that_of_the_first_offending_point():
    """The reference validates pair by pair (fromRawCoords: on the curve, then in the subgroup): with several bad points the
    status is the first one's.  All subgroup checks of a call run as one GPU launch (ctt_hip_subgroup_check)."""
    from constantine_amd import evm
    from constantine_amd.msm import subgroup_check
    from oracle import cref
    sub = next(inp for _, inp, err in DOC["g1_fail"] if "subgroup" in err)
    off_subgroup = bytes.fromhex(sub)[:128]                      # on the curve, outside the subgroup
    off_curve = (1).to_bytes(64, "big") + (1).to_bytes(64, "big")
    good = bytes.fromhex(DOC["g1"][0][1])[:128]
    k = (5).to_bytes(32, "big")
    too_large = b"\x00" * 16 + b"\xff" * 48 + (1).to_bytes(64, "big")     # x >= p: parsed in pair order like the rest
    for pts, want in (((off_subgroup, off_curve), "cttEVM_PointNotInSubgroup"),
                      ((good, off_curve, too_large), "cttEVM_PointNotOnCurve"),
                      ((good, off_subgroup, too_large), "cttEVM_PointNotInSubgroup"),
                      ((good, too_large, off_curve, off_subgroup), "cttEVM_IntLargerThanModulus"),
                      ((good, off_curve, off_subgroup), "cttEVM_PointNotOnCurve"),
                      ((good, good, off_subgroup, off_curve), "cttEVM_PointNotInSubgroup"),
                      ((off_curve, off_subgroup), "cttEVM_PointNotOnCurve")):
        with pytest.raises(evm.EvmError) as e:
            evm.eth_evm_bls12381_g1msm(b"".join(p + k for p in pts))
        assert e.value.status.name == want
    # the batched check itself, all six curves: subgroup points pass, the neutral passes (65 points: the GPU launch for the curves
    # without an endomorphism test, 300: for BLS12-381 too)
    for name in cref.AFF_BYTES:
        pts = cref.gen_points(name, 3, 300 if name.startswith("bls12_381") else 65)
        pts[7] = 0
        assert subgroup_check(name, pts).all(), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bls12_381_g1", "bls12_381_g2"])
def test_subgroup_check_kernel_rejects_points_outside_the_subgroup(name):
    """The GPU form of the reference's endomorphism tests (k_subgroup_check -> bls12_381_g1/g2_in_subgroup, msm_bodies.h;
    bls12_381_subgroups.nim:170-207) on more points than the host path takes: points of the curve outside the subgroup (the
    reference's failure vectors, their multiples, their sums with subgroup points), a point of order 3, the neutral, subgroup
    points -- against [r]P = neutral by the big-integer oracle."""
    from constantine_amd.msm import subgroup_check
    from oracle import cref
    from oracle import pyoracle as po
    curve = po.CURVES[name]
    group = name[-2:]
    rec = 128 if group == "g1" else 256
    raw = bytes.fromhex(next(inp for _, inp, err in DOC[group + "_fail"] if "subgroup" in err))[:rec]
    c = [int.from_bytes(raw[64 * i:64 * i + 64], "big") for i in range(rec // 64)]
    Q = (c[0], c[1]) if group == "g1" else ((c[0], c[1]), (c[2], c[3]))
    assert curve.is_on_curve(Q)
    inside = [curve.aff_from_bytes(bytes(b)) for b in cref.gen_points(name, 21, 5)]
    distinct = inside + [Q, curve.scalar_mul(2, Q), curve.scalar_mul(5, Q), curve.add(Q, inside[0]),
                         curve.add(curve.scalar_mul(7, Q), inside[1]), None]
    if group == "g1":
        distinct.append((0, 2))                                   # x = 0: order 3
    want = [curve.scalar_mul(curve.order, P) is None for P in distinct]
    assert want[:5] == [True] * 5 and not any(want[5:10]) and want[10] is True
    reps = 400 // len(distinct) + 1                               # > 256 points: the launch, not the host path
    got = subgroup_check(name, curve.points_to_array(distinct * reps))
    assert [bool(v) for v in got] == want * reps


@pytest.mark.gpu
def test_points_of_small_order_go_through_the_msm_and_are_rejected():
    """up to 256 pairs the MSM is issued while the subgroup checks still run on host threads, so curve points OUTSIDE the
    subgroup -- of order 3, of order 11, of full cofactor order -- reach the sort and the accumulation with scalars reduced mod r
    before the call is rejected.  The kernels must take any curve point (their additions handle P = +-Q and the neutral; nothing
    assumes the order), and the status must be cttEVM_PointNotInSubgroup whatever position the point has."""
    from constantine_amd import evm
    from oracle import pyoracle as po
    curve = po.BLS12_381_G1
    p, r = curve.F.p, curve.order
    h = 0x396c8c005555e1568c00aaab0000aaab                    # cofactor of E(Fp) (config_fields_and_curves.nim:269-287) = 3 * 11^2 * ...
    assert h % 3 == 0 and h % 121 == 0
    off = []
    x = 1
    while len(off) < 3:
        x += 1
        y2 = (x * x * x + 4) % p
        y = pow(y2, (p + 1) // 4, p)
        if y * y % p != y2:
            continue
        P = (x, y)
        if curve.scalar_mul(r, P) is None:
            continue                                           # (in the subgroup by accident)
        off.append(P)
    small = []
    for P in off:
        Q = curve.scalar_mul(h // 3 * r, P)                    # the 3-part of E(Fp) is Z/3
        if Q is not None:
            assert curve.scalar_mul(3, Q) is None
            small.append((3, Q))
        Q = curve.scalar_mul(h // 121 * r, P)                  # the 11-part has order 121: Z/121 or Z/11 x Z/11
        if Q is not None and curve.scalar_mul(11, Q) is not None:
            Q = curve.scalar_mul(11, Q)
        if Q is not None:
            assert curve.scalar_mul(11, Q) is None             # order exactly 11
            small.append((11, Q))
    assert {q for q, _ in small} == {3, 11}

    def enc(P, k):
        return P[0].to_bytes(64, "big") + P[1].to_bytes(64, "big") + k.to_bytes(32, "big")
    G = curve.gen
    bad_points = [Q for _, Q in small] + off
    for Q in bad_points:
        for k in (1, 2, 3, r - 1, 2**256 - 1):
            with pytest.raises(evm.EvmError) as e:
                evm.eth_evm_bls12381_g1msm(enc(Q, k))
            assert e.value.status == evm.CttEVMStatus.cttEVM_PointNotInSubgroup
    # among valid pairs, at the front, in the middle and at the end of calls that take the "MSM beside the checks" path (<= 256
    # pairs) and of one that checks first (> 256); many copies of an order-3 point (P = +-Q additions inside one bucket)
    for n, where in ((16, 0), (100, 57), (256, 255), (300, 299)):
        pairs = [enc(curve.scalar_mul(i + 2, G), 0x1234567 * (i + 1)) for i in range(8)]
        body = [pairs[i % 8] for i in range(n)]
        body[where] = enc(small[0][1], 7)
        with pytest.raises(evm.EvmError) as e:
            evm.eth_evm_bls12381_g1msm(b"".join(body))
        assert e.value.status == evm.CttEVMStatus.cttEVM_PointNotInSubgroup, (n, where)
    body = [enc(small[0][1], i + 1) for i in range(64)]
    with pytest.raises(evm.EvmError) as e:
        evm.eth_evm_bls12381_g1msm(b"".join(body))
    assert e.value.status == evm.CttEVMStatus.cttEVM_PointNotInSubgroup
    # and the engine is still fine afterwards
    out = evm.eth_evm_bls12381_g1msm(enc(G, 5) + enc(G, 6))
    Q = curve.scalar_mul(11, G)
    assert out == Q[0].to_bytes(64, "big") + Q[1].to_bytes(64, "big")


@pytest.mark.gpu
def test_precompile_calls_from_several_threads():
    """Callers on several threads no longer take turns on one default context: a second context on the same device is opened when
    the first is busy ($CTT_HIP_HOST_CONTEXTS).  Four threads, mixed call sizes, every result the single-threaded one."""
    import threading
    from constantine_amd import evm
    from oracle import pyoracle as po
    curve = po.BLS12_381_G1
    G = curve.gen

    def enc(P, k):
        return P[0].to_bytes(64, "big") + P[1].to_bytes(64, "big") + k.to_bytes(32, "big")
    pts = [curve.scalar_mul(i + 2, G) for i in range(8)]
    jobs = []
    for n in (1, 7, 64, 300):
        ks = [0x9E3779B97F4A7C15 * (i + 1) % curve.order for i in range(n)]
        want = curve.scalar_mul(sum(k * (i % 8 + 2) for i, k in enumerate(ks)) % curve.order, G)
        jobs.append((b"".join(enc(pts[i % 8], k) for i, k in enumerate(ks)), want[0].to_bytes(64, "big") + want[1].to_bytes(64, "big")))
    errors = []

    def work():
        try:
            for _ in range(5):
                for inp, want in jobs:
                    assert evm.eth_evm_bls12381_g1msm(inp) == want
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    ths = [threading.Thread(target=work) for _ in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
