#!/usr/bin/env python3
"""
Re-encode the reference's EIP-4844 `blob_to_kzg_commitment` vectors (a 4096-point BLS12-381 G1 MSM through
kzg_commit, constantine/commitments/kzg.nim:186) into small fixtures.  Run where /root/reference exists:

    python tests/golden/make_golden_kzg.py

Sources (reference-relative):
  constantine/commitments_setups/trusted_setup_ethereum_kzg4844_reference.dat
      text form of the Ethereum KZG ceremony output: "4096\\n65\\n" then one hex line per point; the first 4096
      lines are the G1 points of the SRS in Lagrange form, 48-byte compressed (ZCash encoding)
      -> kzg4844_srs_g1_lagrange.bin   (4096 x 48 bytes, file order)
  tests/protocol_ethereum_eip4844_deneb_kzg/blob_to_kzg_commitment/kzg-mainnet/*valid_blob*/data.yaml
      -> kzg4844_blob_to_commitment.json : [[case, zlib+base64(blob), commitment_hex], ...]
      (all 7 valid blobs -- four structured, three random -- and the 4 rejected ones: every case of the reference's directory)
"""
import base64
import glob
import json
import os
import re
import zlib

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    lines = open(f"{REF}/constantine/commitments_setups/trusted_setup_ethereum_kzg4844_reference.dat").read().split()
    n1, n2 = int(lines[0]), int(lines[1])
    assert (n1, n2) == (4096, 65)
    g1 = b"".join(bytes.fromhex(h) for h in lines[2:2 + n1])
    assert len(g1) == 4096 * 48
    open(os.path.join(HERE, "kzg4844_srs_g1_lagrange.bin"), "wb").write(g1)

    cases = []
    base = f"{REF}/tests/protocol_ethereum_eip4844_deneb_kzg/blob_to_kzg_commitment/kzg-mainnet"
    for d in sorted(glob.glob(f"{base}/*valid_blob*")):
        if "invalid" in d:
            continue
        t = open(f"{d}/data.yaml").read()
        blob = bytes.fromhex(re.search(r"blob: '0x([0-9a-f]+)'", t).group(1))
        out = re.search(r"output: '0x([0-9a-f]+)'", t).group(1)
        z = zlib.compress(blob, 9)
        cases.append([os.path.basename(d), base64.b64encode(z).decode(), out])
    # invalid blobs (wrong length / a field element >= r): the reference returns an error status, output is null
    for d in sorted(glob.glob(f"{base}/*invalid_blob*")):
        t = open(f"{d}/data.yaml").read()
        blob = bytes.fromhex(re.search(r"blob: '0x([0-9a-f]*)'", t).group(1))
        assert "output: null" in t
        z = zlib.compress(blob, 9)
        if len(z) > 4096:           # the two wrong-length cases are random data: keep the length, not the bytes
            cases.append([os.path.basename(d), "LEN:%d" % len(blob), None])
        else:
            cases.append([os.path.basename(d), base64.b64encode(z).decode(), None])
    json.dump(cases, open(os.path.join(HERE, "kzg4844_blob_to_commitment.json"), "w"), separators=(",", ":"))
    print(len(cases), "cases;", os.path.getsize(os.path.join(HERE, "kzg4844_blob_to_commitment.json")), "bytes")
    make_proofs(cases)


def make_proofs(commit_cases):
    """compute_kzg_proof / compute_blob_kzg_proof vectors whose blob is one of the blobs kept above (referenced by the
    name of that case, so no blob is stored twice):
      tests/protocol_ethereum_eip4844_deneb_kzg/compute_kzg_proof/kzg-mainnet/*/data.yaml
          -> "compute_kzg_proof": [[case, blob_case, z_hex, [proof_hex, y_hex] | null], ...]
      tests/protocol_ethereum_eip4844_deneb_kzg/compute_blob_kzg_proof/kzg-mainnet/*/data.yaml
          -> "compute_blob_kzg_proof": [[case, blob_case, commitment_hex, proof_hex | null], ...]"""
    import hashlib
    known, known_len = {}, {}
    for name, z, _ in commit_cases:
        blob = bytes(int(z[4:])) if z.startswith("LEN:") else zlib.decompress(base64.b64decode(z))
        if not z.startswith("LEN:"):
            known[hashlib.sha256(blob).digest()] = name
        else:
            known_len[len(blob)] = name      # the wrong-length blobs are random data: what is tested is the length
    out = {"compute_kzg_proof": [], "compute_blob_kzg_proof": []}
    root = f"{REF}/tests/protocol_ethereum_eip4844_deneb_kzg"
    for d in sorted(glob.glob(f"{root}/compute_kzg_proof/kzg-mainnet/*")):
        t = open(f"{d}/data.yaml").read()
        blob = bytes.fromhex(re.search(r"blob: '0x([0-9a-f]*)'", t).group(1))
        z = re.search(r"z: '0x([0-9a-f]*)'", t).group(1)
        ref = known.get(hashlib.sha256(blob).digest()) or known_len.get(len(blob))
        if ref is None:
            continue
        m = re.search(r"output: \['0x([0-9a-f]+)',\s*'0x([0-9a-f]+)'\]", t)
        assert m or "output: null" in t, d
        out["compute_kzg_proof"].append([os.path.basename(d), ref, z, [m.group(1), m.group(2)] if m else None])
    for d in sorted(glob.glob(f"{root}/compute_blob_kzg_proof/kzg-mainnet/*")):
        t = open(f"{d}/data.yaml").read()
        blob = bytes.fromhex(re.search(r"blob: '0x([0-9a-f]*)'", t).group(1))
        com = re.search(r"commitment: '0x([0-9a-f]*)'", t).group(1)
        ref = known.get(hashlib.sha256(blob).digest()) or known_len.get(len(blob))
        if ref is None:
            continue
        m = re.search(r"output: '0x([0-9a-f]+)'", t)
        assert m or "output: null" in t, d
        out["compute_blob_kzg_proof"].append([os.path.basename(d), ref, com, m.group(1) if m else None])
    path = os.path.join(HERE, "kzg4844_proofs.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print({k: (len(v), sum(1 for c in v if c[3] is None)) for k, v in out.items()}, "(cases, rejected);", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
