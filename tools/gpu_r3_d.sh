#!/bin/bash
# Round 3, GPU call D: balanced windows.  Parity suite, window-bit sweeps per size (re-fit of choose_window_bits), the
# window table over c, lanes / tail-stream experiments at small N.
set -u
OUT=$PWD/gpurun_out/r3d
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
export CTT_HIP_MSM_DUAL_LOG2=0
timeout 600 python tools/sweep.py bls12_381_g1 12 c=6,7,8,9,10,11 -- bls12_381_g1 14 c=8,9,10,11,12,13 -- bls12_381_g1 16 c=10,11,12,13,14,15,16 -- bls12_381_g1 17 c=11,12,13,14,15,16 \
   -- bls12_381_g1 18 c=12,13,14,15,16 -- bls12_381_g1 19 c=13,14,15,16 -- bls12_381_g1 20 c=14,15,16 -- bls12_381_g1 22 c=15,16 > "$OUT/sweep_c_g1.jsonl" 2> "$OUT/sweep_c_g1.err"
timeout 600 python tools/sweep.py bn254_snarks_g1 16 c=11,12,13,14,15,16 -- bn254_snarks_g1 18 c=13,14,15,16 -- bn254_snarks_g1 20 c=14,15,16 -- bn254_snarks_g1 22 c=15,16 \
   -- pallas 16 c=12,13,14 -- pallas 20 c=14,15,16 -- bls12_381_g2 16 c=10,11,12,13,14 -- bls12_381_g2 18 c=11,12,13,14,15,16 -- bls12_381_g2 20 c=13,14,15,16 > "$OUT/sweep_c_other.jsonl" 2> "$OUT/sweep_c_other.err"
{ timeout 300 python tools/bench_table.py bls12_381_g1 20 0 17 18 19 20 21
  timeout 300 python tools/bench_table.py bls12_381_g1 16 0 13 14 15 16 17
  timeout 300 python tools/bench_table.py bn254_snarks_g1 22 0 19 20 21
  timeout 300 python tools/bench_table.py pallas 20 0 18 19 20 21; } 2> "$OUT/table.err" | grep '^{' > "$OUT/table.jsonl"
unset CTT_HIP_MSM_DUAL_LOG2
# lanes: two independent engines for small MSMs, with and without the tail streams
for cfg in "0 0" "18 0" "18 1" "0 1"; do
  set -- $cfg
  CTT_HIP_MSM_DUAL_LOG2=$1 CTT_HIP_MSM_NO_TAIL=$2 timeout 300 python tools/sweep.py bls12_381_g1 12 c=0 -- bls12_381_g1 14 c=0 -- bls12_381_g1 16 c=0 -- bls12_381_g1 17 c=0 -- bls12_381_g1 18 c=0 \
     > "$OUT/lanes_dual$1_notail$2.jsonl" 2> "$OUT/lanes.err"
done
python3 - <<'PY'
import json,glob,os
out=os.environ['OUT'] if 'OUT' in os.environ else 'gpurun_out/r3d'
for f in sorted(glob.glob(out+'/*.jsonl')):
    print('==',os.path.basename(f))
    for l in open(f):
        d=json.loads(l)
        if 'opt' in d: print(d['curve'][:10],d['log2n'],d['opt'],'c',d['plan']['c'],'W',d['plan']['W'],'K',d['plan']['K'],'step',d['ms_per_step'],'blk',d['blocking_ms'],d['same'],' '.join(f"{k[:3]}={v}" for k,v in d['stage_ms_blocking'].items()))
        else: print(d['curve'][:10],d['log2n'],'table',d['table'],'c',d['c'],'build',d['build_ms'],'step',d['ms_per_step'],d['same'],' '.join(f"{k[:3]}={v}" for k,v in d['stage_ms_blocking'].items()))
PY
