#!/usr/bin/env python3
"""
Re-encode the reference's golden vectors for the MSM path into small fixtures.

Run in the build container (where /root/reference exists):
    python tests/golden/make_golden.py

Sources (reference-relative):
  tests/math_elliptic_curves/vectors/tv_<curve>_scalar_mul_<G1|G2>_<bits>bit.json
      -> scalar_mul_kat.json : {curve: [[P, k, Q], ...]}   (hex ints; Fp2 as [c0, c1])
  tests/protocol_ethereum_evm_precompiles/eip-2537/multiexp_{G1,G2}_bls.json, fail-multiexp_{G1,G2}_bls.json
      -> eip2537_multiexp.json : {"g1": [[name, input_hex, expected_hex], ...], "g2": [...],
                                  "g1_fail": [[name, input_hex, expected_error], ...], "g2_fail": [...]}

The GPU box has no /root/reference: tests read only the files written here.
"""
import json
import os

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))

SCALAR_MUL = {
    "bls12_381_g1": ("BLS12_381", "G1", (32, 64, 128, 255)),
    "bls12_381_g2": ("BLS12_381", "G2", (32, 64, 128, 255)),
    "bn254_snarks_g1": ("BN254_Snarks", "G1", (32, 64, 128, 254)),
    "bn254_snarks_g2": ("BN254_Snarks", "G2", (32, 64, 128, 254)),
    "pallas": ("Pallas", "G1", (255,)),
    "vesta": ("Vesta", "G1", (255,)),
}


def coord(v):
    if isinstance(v, dict):
        return [v["c0"], v["c1"]]
    return v


def main():
    out = {}
    for name, (curve, group, sizes) in SCALAR_MUL.items():
        rows = []
        for bits in sizes:
            path = f"{REF}/math_elliptic_curves/vectors/tv_{curve}_scalar_mul_{group}_{bits}bit.json"
            doc = json.load(open(path))
            for v in doc["vectors"]:
                rows.append([[coord(v["P"]["x"]), coord(v["P"]["y"])], v["scalar"],
                             [coord(v["Q"]["x"]), coord(v["Q"]["y"])]])
        out[name] = rows
    with open(os.path.join(HERE, "scalar_mul_kat.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))

    eip = {}
    for g in ("G1", "G2"):
        doc = json.load(open(f"{REF}/protocol_ethereum_evm_precompiles/eip-2537/multiexp_{g}_bls.json"))
        eip[g.lower()] = [[t["Name"], t["Input"], t["Expected"]] for t in doc]
        fail = json.load(open(f"{REF}/protocol_ethereum_evm_precompiles/eip-2537/fail-multiexp_{g}_bls.json"))
        eip[g.lower() + "_fail"] = [[t["Name"], t["Input"], t["ExpectedError"]] for t in fail]
    with open(os.path.join(HERE, "eip2537_multiexp.json"), "w") as f:
        json.dump(eip, f, separators=(",", ":"))
    print({k: len(v) for k, v in out.items()}, {k: len(v) for k, v in eip.items()})


if __name__ == "__main__":
    main()
