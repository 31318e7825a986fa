#!/bin/bash
# Round-2 experiment A (run through gpurun from the repo root): instruction rates, GPU tests, and the accumulate-kernel
# variants (CTT_FPU_CHAIN = 0 C++ columns / 1 / 4 / 8 multiply-adds per asm statement) on the headline and the other configs.
set -u
OUT=$PWD/gpurun_out/r2a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 120 tools/microbench_isa.bin > "$OUT/microbench_isa.jsonl" 2> "$OUT/microbench_isa.err"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
for v in "" _c0 _c8 _c1; do
  lib=$PWD/constantine_amd/libctt_msm_hip$v.so
  [ -f "$lib" ] || continue
  for cfg in "bls12_381_g1 20" "bn254_snarks_g1 22" "pallas 20" "bls12_381_g2 20" "bls12_381_g1 16"; do
    set -- $cfg
    CTT_MSM_HIP_LIB=$lib timeout 300 python bench.py --curve $1 --log2n $2 --steps 30 --warmup 5 --no-cpu-baseline --no-latency \
        > "$OUT/bench${v}_$1_$2.json" 2>> "$OUT/bench.err"
    python - "$OUT/bench${v}_$1_$2.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"variant '{sys.argv[2]}' {d['config']['workload'][:40]}: {d['value']/1e6:.1f} M/s, {d['ms_per_step']:.3f} ms, stages {d['stage_ms']}")
except Exception as e:
    print("FAILED", sys.argv[1], e)
PY
  done
done
timeout 600 python bench.py > "$OUT/bench_full.json" 2>> "$OUT/bench.err"
cat "$OUT/bench_full.json"
tail -20 "$OUT/bench.err"
# the multi-rank path of bench.py on this single-GPU box: two ranks (self-launched), both on device 0, gloo exchange
timeout 300 python bench.py --gpus 2 --all-ranks-on-device 0 --backend gloo --steps 10 --warmup 2 > "$OUT/bench_2ranks_one_gpu.json" 2>> "$OUT/bench.err"
cat "$OUT/bench_2ranks_one_gpu.json"
timeout 300 python bench.py --gpus 2 --total-log2n 21 --all-ranks-on-device 0 --backend gloo --steps 10 --warmup 2 > "$OUT/bench_2ranks_one_gpu_strong.json" 2>> "$OUT/bench.err"
cat "$OUT/bench_2ranks_one_gpu_strong.json"
