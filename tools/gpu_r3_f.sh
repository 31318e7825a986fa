#!/bin/bash
# Round 3, GPU call F: parity (incl. the KZG quotient on the device), KZG proof timing, small-N bench lines with accumulate-only events.
set -u
OUT=$PWD/gpurun_out/r3f
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
python3 - > "$OUT/kzg_timing.txt" 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from constantine_amd import kzg
from tests import _golden
ctx = kzg.EthereumKZGContext(open(os.path.join(_golden.HERE, "kzg4844_srs_g1_lagrange.bin"), "rb").read())
rng = np.random.default_rng(3)
blob = b"".join((int.from_bytes(rng.bytes(32), "big") % kzg._R).to_bytes(32, "big") for _ in range(4096))
z = (12345678901234567890 % kzg._R).to_bytes(32, "big")
for _ in range(3): kzg.compute_kzg_proof(ctx, blob, z); kzg.blob_to_kzg_commitment(ctx, blob)
for name, fn in (("blob_to_kzg_commitment", lambda: kzg.blob_to_kzg_commitment(ctx, blob)), ("compute_kzg_proof", lambda: kzg.compute_kzg_proof(ctx, blob, z))):
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: median {sorted(ts)[10]:.3f} ms (min {min(ts):.3f})")
poly = kzg.blob_to_bigint_polynomial(blob)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); kzg.quotient_polynomial_device(ctx, poly, int.from_bytes(z, "big")); ts.append((time.perf_counter() - t0) * 1e3)
print(f"quotient_polynomial_device (upload of the blob's scalars + ctt_hip_fr_quotient): median {sorted(ts)[10]:.3f} ms")
t0 = time.perf_counter(); kzg.quotient_polynomial([int.from_bytes(bytes(r), 'little') for r in poly], int.from_bytes(z, "big")); print(f"quotient_polynomial (host integers, rounds 1-2): {(time.perf_counter()-t0)*1e3:.3f} ms")
PY
cat "$OUT/kzg_timing.txt"
for k in 16 17 18 20; do
  timeout 300 python bench.py --log2n $k --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_bls12_381_g1_2pow$k.json" 2>> "$OUT/bench.err"
done
timeout 300 python tools/sweep.py bn254_snarks_g1 20 c=0 -- bn254_snarks_g1 22 c=0 -- bls12_381_g1 16 c=0 -- bls12_381_g1 17 c=0 > "$OUT/sweep.jsonl" 2>> "$OUT/bench.err"
python3 - <<'PY'
import json,os
out=os.environ['OUT']
for k in (16,17,18,20):
    d=json.load(open(f'{out}/bench_bls12_381_g1_2pow{k}.json'))
    print(k,'step',round(d['ms_per_step'],4),'blk',round(d['latency_ms_blocking'],4),'hostptr',round(d['hostptr_ms'],3),'c',d['config']['window_bits'],{a:round(b,3) for a,b in d['stage_ms'].items()},'traffic',d['roofline']['traffic'],{a:(round(v['ms_per_step'],3),round(v['latency_ms_blocking'],3),v['window_bits']) for a,v in d['cached_bases'].items()})
for l in open(f'{out}/sweep.jsonl'):
    d=json.loads(l); print(d['curve'][:12],d['log2n'],'c',d['plan']['c'],'step',d['ms_per_step'],'blk',d['blocking_ms'],d['same'])
PY
