#!/bin/bash
# Round 3, GPU call G: host time per submit / finish at small N, MFMA micro-benchmark, host-pointer calls with the new slice sizing.
set -u
OUT=$PWD/gpurun_out/r3g
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
timeout 200 tools/microbench_mfma.bin > "$OUT/microbench_mfma.jsonl" 2> "$OUT/microbench_mfma.err"
cat "$OUT/microbench_mfma.jsonl"
timeout 300 python tools/sweep.py bls12_381_g1 12 c=0 -- bls12_381_g1 14 c=0 -- bls12_381_g1 16 c=0 -- bls12_381_g1 17 c=0 -- bls12_381_g1 18 c=0 -- bls12_381_g1 20 c=0 -- bn254_snarks_g1 16 c=0 > "$OUT/sweep_host.jsonl" 2> "$OUT/sweep.err"
timeout 300 python tools/bench_hostptr.py > "$OUT/hostptr.txt" 2> "$OUT/hostptr.err"
timeout 300 python bench.py --curve bn254_snarks_g1 --log2n 22 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_bn254_2pow22.json" 2>> "$OUT/bench.err"
timeout 300 python bench.py --curve pallas --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_pallas.json" 2>> "$OUT/bench.err"
python3 - <<'PY'
import json,os
out=os.environ['OUT']
for l in open(f'{out}/sweep_host.jsonl'):
    d=json.loads(l); print(d['curve'][:12],d['log2n'],'c',d['plan']['c'],'step',d['ms_per_step'],'blk',d['blocking_ms'],'host submit',d['host_ms_in_submit'],'finish',d['host_ms_in_finish'])
print(open(f'{out}/hostptr.txt').read())
for f in ('bench_bn254_2pow22','bench_pallas'):
    d=json.load(open(f'{out}/{f}.json')); print(f,'step',round(d['ms_per_step'],3),'blk',round(d['latency_ms_blocking'],3),'hostptr',round(d['hostptr_ms'],3))
PY
