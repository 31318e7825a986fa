#!/bin/bash
# Round 3, GPU call C: parity suite, then on the same box the round-2 library against the current one (per-pass pyramid +
# grouped bit Horner + device-decided merge + second lane for small MSMs), lane and Horner-group knobs.
set -u
OUT=$PWD/gpurun_out/r3c
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
SIZES="bls12_381_g1 20 c=16 -- bls12_381_g1 19 c=16 -- bls12_381_g1 18 c=16,13 -- bls12_381_g1 17 c=16,13 -- bls12_381_g1 16 c=13 -- bls12_381_g1 14 c=10 -- bls12_381_g1 12 c=8 -- bls12_381_g2 18 c=16,13 -- bn254_snarks_g1 16 c=13 -- bn254_snarks_g1 22 c=16"
CTT_MSM_HIP_LIB=$REPO/constantine_amd/libctt_msm_hip_r2.so timeout 400 python tools/sweep.py $SIZES > "$OUT/ab_r2.jsonl" 2> "$OUT/ab_r2.err"
timeout 400 python tools/sweep.py $SIZES > "$OUT/ab_cur.jsonl" 2> "$OUT/ab_cur.err"
CTT_HIP_MSM_DUAL_LOG2=0 timeout 400 python tools/sweep.py $SIZES > "$OUT/ab_cur_nodual.jsonl" 2> "$OUT/ab_cur_nodual.err"
timeout 400 python tools/sweep.py bls12_381_g1 20 horner_bits=1,2,3,4,6,8,15 -- bls12_381_g1 16 c=13 horner_bits=1,2,3,4,6,12 -- bls12_381_g2 18 horner_bits=1,2,4,8 > "$OUT/horner.jsonl" 2> "$OUT/horner.err"
for k in 16 18 20; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$k" -o p -- python "$REPO/bench.py" --log2n $k --steps 10 --warmup 2 \
      --no-cpu-baseline --no-latency > /dev/null 2> "$OUT/prof_$k.log" )
  DB=$(find "$OUT/prof_$k" -name "*.db" | head -1)
  python tools/kernel_timeline.py "$DB" 2 > "$OUT/rocprof_2pow$k.txt" 2>> "$OUT/prof_$k.log"
  find "$OUT/prof_$k" -name "*.db" -delete 2>/dev/null
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_2pow20.json" 2> "$OUT/bench.err"
timeout 300 python tools/bench_hostptr.py > "$OUT/hostptr.txt" 2>> "$OUT/bench.err"
python3 - <<'PY'
import json,glob,os
out=os.environ.get('OUT','gpurun_out/r3c')
for f in ('ab_r2','ab_cur','ab_cur_nodual'):
    print('==',f)
    for l in open(f'{out}/{f}.jsonl'):
        d=json.loads(l); print(d['curve'][:10],d['log2n'],d['opt'],'step',d['ms_per_step'],'blk',d['blocking_ms'],d['same'],' '.join(f"{k[:3]}={v}" for k,v in d['stage_ms_blocking'].items()))
PY
