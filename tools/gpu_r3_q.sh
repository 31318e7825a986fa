#!/bin/bash
# Round 3, GPU call Q: wave slots the accumulate grid of a pipelined MSM leaves free for the previous MSM's tail (0 = the accumulation
# waits for that tail, the behaviour up to here), per size.   gpurun --timeout 900 -- 'bash tools/gpu_r3_q.sh'
set -u
OUT=$PWD/gpurun_out/r3q
mkdir -p "$OUT"; : > "$OUT/sweep_free.jsonl"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for free in 0 16 32 48 64 0 32; do
  echo "== free wave slots $free" >> "$OUT/sweep_free.jsonl"
  if [ $free = 0 ]; then L=10; else L=22; fi
  CTT_HIP_MSM_TAIL_FREE=$free CTT_HIP_MSM_TAIL_FREE_LOG2N=$L timeout 300 python tools/sweep.py bls12_381_g1 18 c=0 -- bls12_381_g1 19 c=0 -- bls12_381_g1 20 c=0 -- bls12_381_g1 21 c=0 \
     -- bls12_381_g1 22 c=0 -- bn254_snarks_g1 20 c=0 -- bls12_381_g2 18 c=0 -- bls12_381_g2 20 c=0 -- pallas 20 c=0 >> "$OUT/sweep_free.jsonl" 2>> "$OUT/err.txt"
done
python - <<'PY'
import json
for l in open("gpurun_out/r3q/sweep_free.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l)
    print(d["curve"], d["log2n"], "ms", d["ms_per_step"], "blocking", d["blocking_ms"], "K", d["plan"]["K"], "c", d["plan"]["c"])
PY
