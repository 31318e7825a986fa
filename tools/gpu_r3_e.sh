#!/bin/bash
# Round 3, GPU call E: parity, the model's window choice at every size, wide windows at large N, slices vs Infinity Cache.
set -u
OUT=$PWD/gpurun_out/r3e
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
timeout 600 python tools/sweep.py bls12_381_g1 10 c=0 -- bls12_381_g1 12 c=0 -- bls12_381_g1 14 c=0 -- bls12_381_g1 16 c=0 -- bls12_381_g1 17 c=0,13 -- bls12_381_g1 18 c=0 -- bls12_381_g1 19 c=0 -- bls12_381_g1 20 c=0 \
   -- bls12_381_g2 16 c=0 -- bls12_381_g2 18 c=0 -- bls12_381_g2 20 c=0 -- bn254_snarks_g1 16 c=0 -- bn254_snarks_g1 18 c=0 -- bn254_snarks_g1 20 c=0 -- pallas 16 c=0 -- pallas 20 c=0 -- vesta 20 c=0 -- bn254_snarks_g2 16 c=0,12 \
   > "$OUT/sweep_auto.jsonl" 2> "$OUT/sweep_auto.err"
timeout 600 python tools/sweep.py bn254_snarks_g1 22 c=16,17,18 -- bls12_381_g1 22 c=16,17,18 -- bls12_381_g1 24 c=16,18 -- pallas 22 c=16,17 > "$OUT/sweep_wide.jsonl" 2> "$OUT/sweep_wide.err"
timeout 300 python tools/bench_slices.py 22 1 2 4 8 > "$OUT/slices.jsonl" 2> "$OUT/slices.err"
timeout 300 python tools/bench_slices.py 24 1 4 8 >> "$OUT/slices.jsonl" 2>> "$OUT/slices.err"
timeout 300 python tools/bench_hostptr.py > "$OUT/hostptr.txt" 2> "$OUT/hostptr.err"
for k in 16 17 18 20; do
  timeout 300 python bench.py --log2n $k --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_bls12_381_g1_2pow$k.json" 2>> "$OUT/bench.err"
done
python3 - <<'PY'
import json,glob,os
out=os.environ['OUT']
for f in ('sweep_auto','sweep_wide'):
    print('==',f)
    for l in open(f'{out}/{f}.jsonl'):
        d=json.loads(l); print(d['curve'][:12],d['log2n'],d['opt'],'c',d['plan']['c'],'W',d['plan']['W'],'K',d['plan']['K'],'step',d['ms_per_step'],'blk',d['blocking_ms'],d['same'],' '.join(f"{k[:3]}={v}" for k,v in d['stage_ms_blocking'].items()))
print(open(f'{out}/slices.jsonl').read())
for k in (16,17,18,20):
    d=json.load(open(f'{out}/bench_bls12_381_g1_2pow{k}.json'))
    print(k,'step',round(d['ms_per_step'],4),'blk',round(d['latency_ms_blocking'],4),'hostptr',round(d['hostptr_ms'],3),{a:round(b,3) for a,b in d['stage_ms'].items()},{a:(round(v['ms_per_step'],3),round(v['latency_ms_blocking'],3),v['window_bits']) for a,v in d['cached_bases'].items()})
PY
