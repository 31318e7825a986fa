#!/bin/bash
# Round-2 experiment B: neutral flags in the accumulate kernel, v_bfi select, H2D overlap probe, GPU tests
set -u
OUT=$PWD/gpurun_out/r2b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 120 tools/h2d_overlap.bin > "$OUT/h2d_overlap.jsonl" 2> "$OUT/h2d_overlap.err"; cat "$OUT/h2d_overlap.jsonl"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
for v in "" _bfi; do
  lib=$PWD/constantine_amd/libctt_msm_hip$v.so
  [ -f "$lib" ] || continue
  for cfg in "bls12_381_g1 20" "bn254_snarks_g1 22" "pallas 20" "bls12_381_g2 20" "bn254_snarks_g1 20"; do
    set -- $cfg
    CTT_MSM_HIP_LIB=$lib timeout 300 python bench.py --curve $1 --log2n $2 --steps 30 --warmup 5 --no-cpu-baseline --no-latency \
        > "$OUT/bench${v}_$1_$2.json" 2>> "$OUT/bench.err"
    python - "$OUT/bench${v}_$1_$2.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"variant '{sys.argv[2]}' {d['config']['workload'][:40]}: {d['value']/1e6:.1f} M/s, {d['ms_per_step']:.3f} ms, acc {d['stage_ms']['accumulate']:.3f} reduce {d['stage_ms']['reduce']:.3f}")
except Exception as e:
    print("FAILED", sys.argv[1], e)
PY
  done
done
grep -v amdgpu.ids "$OUT/bench.err" | tail -5
