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
