"""EIP-2537 G1MSM / G2MSM wire format (constantine_amd/evm.py) against the reference's pass and fail vectors
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
