#!/bin/bash
# Round 3, GPU call R: the round-2 library (built from commit 628964b as constantine_amd/libctt_msm_hip_r2.so) beside the final one on one
# box: ms per MSM with two in flight and blocking call, every BASELINE curve and the size range.   gpurun --timeout 900 -- 'bash tools/gpu_r3_r.sh'
# The round-2 library is not kept in the tree; to rebuild it:
#     mkdir /tmp/r2 && git archive 628964b constantine_amd/csrc | tar -x -C /tmp/r2 && make -C /tmp/r2/constantine_amd/csrc -j8 && \
#     cp /tmp/r2/constantine_amd/libctt_msm_hip.so constantine_amd/libctt_msm_hip_r2.so
set -u
OUT=$PWD/gpurun_out/r3r
mkdir -p "$OUT"; : > "$OUT/ab.jsonl"
for v in _r2 "" _r2 ""; do
  LIB=$PWD/constantine_amd/libctt_msm_hip$v.so
  [ -f "$LIB" ] || continue
  echo "== library '$v'" >> "$OUT/ab.jsonl"
  CTT_MSM_HIP_LIB=$LIB timeout 400 python tools/sweep.py bls12_381_g1 16 c=0 -- bls12_381_g1 17 c=0 -- bls12_381_g1 18 c=0 -- bls12_381_g1 19 c=0 -- bls12_381_g1 20 c=0 \
     -- bls12_381_g1 22 c=0 -- bls12_381_g1 24 c=0 -- bn254_snarks_g1 22 c=0 -- pallas 20 c=0 -- vesta 20 c=0 -- bls12_381_g2 18 c=0 -- bls12_381_g2 20 c=0 >> "$OUT/ab.jsonl" 2>> "$OUT/err.txt"
done
python - <<'PY'
import json
for l in open("gpurun_out/r3r/ab.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l)
    print(" ", d["curve"], "2^%d" % d["log2n"], "ms/MSM", d["ms_per_step"], "blocking", d["blocking_ms"], "c", d["plan"]["c"], "crc", d.get("crc"))
PY
