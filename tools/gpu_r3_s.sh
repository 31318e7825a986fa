#!/bin/bash
# Round 3, GPU call S: wide reduction passes on the tail stream beside the next MSM's sort (CTT_HIP_MSM_WIDE_EARLY=1), the next accumulation
# waiting for the wide passes only.   gpurun --timeout 900 -- 'bash tools/gpu_r3_s.sh'
set -u
OUT=$PWD/gpurun_out/r3s
mkdir -p "$OUT"; : > "$OUT/ab.jsonl"
for we in 0 1 0 1; do
  echo "== wide_early $we" >> "$OUT/ab.jsonl"
  CTT_HIP_MSM_WIDE_EARLY=$we timeout 400 python tools/sweep.py bls12_381_g1 16 c=0 -- bls12_381_g1 17 c=0 -- bls12_381_g1 18 c=0 -- bls12_381_g1 19 c=0 -- bls12_381_g1 20 c=0 \
     -- bls12_381_g1 22 c=0 -- bn254_snarks_g1 22 c=0 -- pallas 20 c=0 -- bls12_381_g2 18 c=0 -- bls12_381_g2 20 c=0 >> "$OUT/ab.jsonl" 2>> "$OUT/err.txt"
done
python - <<'PY'
import json
for l in open("gpurun_out/r3s/ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l)
    print(" ", d["curve"], "2^%d" % d["log2n"], "ms/MSM", d["ms_per_step"], "blocking", d["blocking_ms"], "c", d["plan"]["c"], "crc", d.get("crc"))
PY
