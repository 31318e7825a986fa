#!/bin/bash
# Round 3, GPU call H: accumulate kernel with the gather software-pipelined (variants built side by side), same box A/B.
#     gpurun --timeout 900 -- 'bash tools/gpu_r3_h.sh'
set -u

OUT=$PWD/gpurun_out/r3h
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in "" _ch14 "" _ch14; do
  LIB=$PWD/constantine_amd/libctt_msm_hip$v.so
  [ -f "$LIB" ] || continue
  echo "== variant '$v'" >> "$OUT/ab.jsonl"
  CTT_MSM_HIP_LIB=$LIB timeout 300 python tools/sweep.py bls12_381_g2 20 c=16 -- bls12_381_g2 18 c=14 -- vesta 20 c=16 -- bn254_snarks_g2 18 c=15 >> "$OUT/ab.jsonl" 2>> "$OUT/ab.err"
done
python - <<'PY'
import json
for l in open("gpurun_out/r3h/ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l)
    print(d["curve"], d["log2n"], "ms/MSM", d["ms_per_step"], "blocking", d["blocking_ms"], "accum", d["stage_ms_blocking"].get("accum"), "same", d["same"], "crc", d["crc"])
PY
