#!/bin/bash
# Round-2 experiment E: window sums on the device (quad Horner per window), DPP quad broadcasts
set -u
OUT=$PWD/gpurun_out/r2e
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -4 "$OUT/pytest_gpu.log"
for cfg in "bls12_381_g1 16" "bls12_381_g1 18" "bls12_381_g1 20" "bn254_snarks_g1 22" "pallas 20" "bls12_381_g2 20"; do
  set -- $cfg
  timeout 300 python bench.py --curve $1 --log2n $2 --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_$1_$2.json" 2>> "$OUT/bench.err"
  python - "$OUT/bench_$1_$2.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{d['config']['workload'][:34]} c={d['config']['window_bits']}: {d['value']/1e6:.1f} M/s, {d['ms_per_step']:.3f} ms, blocking {d.get('latency_ms_blocking',0):.3f} ms, hostptr {d.get('hostptr_ms',0):.3f}", {k: round(v, 3) for k, v in d['stage_ms_blocking'].items()})
except Exception as e:
    print("FAILED", sys.argv[1], e)
PY
done
grep -v amdgpu.ids "$OUT/bench.err" | tail -5
