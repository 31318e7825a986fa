#!/bin/bash
# Round 3, GPU call A: parity suite on the two-launch reduction + device-decided merge, then the sweeps that re-fit the plan
# (window bits per size, reduction block, Horner group, quad ratio) and the small-N kernel timelines.
#     gpurun --timeout 1500 -- 'bash tools/gpu_r3_a.sh'
set -u
OUT=$PWD/gpurun_out/r3a
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"

timeout 500 python tools/sweep.py bls12_381_g1 16 c=10,11,12,13,14,15 -- bls12_381_g1 17 c=12,13,14,15,16 -- bls12_381_g1 18 c=13,14,15,16 \
    -- bls12_381_g1 19 c=14,15,16 -- bls12_381_g1 20 c=15,16 -- bls12_381_g1 14 c=9,10,11,12,13 -- bls12_381_g1 12 c=7,8,9,10,11 \
    > "$OUT/sweep_c.jsonl" 2> "$OUT/sweep_c.err"
timeout 500 python tools/sweep.py bls12_381_g1 20 reduce_block=7,8,9,10 -- bls12_381_g1 20 horner_bits=1,2,4,8,15 -- bls12_381_g1 20 quad_ratio=1,2,3,5 \
    -- bls12_381_g1 16 reduce_block=5,6,8,12 -- bls12_381_g1 16 horner_bits=1,2,4,6,12 -- bls12_381_g1 16 quad_ratio=1,2,3,5,100 \
    -- bls12_381_g1 18 reduce_block=7,8,9,10 horner_bits=2,4,8 \
    > "$OUT/sweep_red.jsonl" 2> "$OUT/sweep_red.err"
timeout 500 python tools/sweep.py bls12_381_g2 18 c=13,14,15,16 -- bn254_snarks_g1 22 c=15,16 -- pallas 20 c=15,16 -- bn254_snarks_g1 16 c=12,13,14 \
    > "$OUT/sweep_other.jsonl" 2> "$OUT/sweep_other.err"

for k in 16 17 18 20; do
  timeout 300 python bench.py --log2n $k --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_bls12_381_g1_2pow$k.json" 2>> "$OUT/bench.err"
done

# kernel timelines (rocpd database summarised by tools/kernel_timeline.py)
for k in 16 18 20; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$k" -o p -- python "$REPO/bench.py" --log2n $k --steps 10 --warmup 2 \
      --no-cpu-baseline --no-latency > /dev/null 2> "$OUT/prof_$k.log" )
  DB=$(find "$OUT/prof_$k" -name "*.db" | head -1)
  python tools/kernel_timeline.py "$DB" > "$OUT/rocprof_kernel_stats_2pow$k.txt" 2>> "$OUT/prof_$k.log"
  find "$OUT/prof_$k" -name "*.db" -delete 2>/dev/null
done
timeout 200 tools/microbench_inv.bin > "$OUT/microbench_inv.jsonl" 2> "$OUT/microbench_inv.err"
timeout 200 python tools/bench_batch_ops.py > "$OUT/batch_ops.txt" 2>&1
cat "$OUT/bench_bls12_381_g1_2pow20.json" | head -c 600
