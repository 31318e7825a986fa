#!/bin/bash
# Round-2 experiment C: host-pointer MSM uploaded in slices
set -u
OUT=$PWD/gpurun_out/r2c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
timeout 600 python tools/bench_hostptr.py > "$OUT/hostptr.txt" 2> "$OUT/hostptr.err"; cat "$OUT/hostptr.txt"; grep -v amdgpu.ids "$OUT/hostptr.err" | tail -5
timeout 600 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench.err"; cat "$OUT/bench_full.json"
