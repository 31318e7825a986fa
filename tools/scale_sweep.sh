#!/bin/bash
# One command for the day an 8-GPU node is available (no scaling curve has been measured on hardware so far: the pool hands
# out 1-GPU boxes).  For N = 1, 2, 4, 8 (as many GPUs as the node has) prints one bench line each of
#   weak    2^20 pairs per GPU                                    (the driver's own `bench.py --gpus N`)
#   strong  2^20 pairs in total                                   (BASELINE metric read as one fixed job)
#   strong  2^24 pairs in total                                   (BASELINE configs[3] at N = 8)
# One process per GPU over RCCL; results under gpurun_out/scale/.  Efficiency is for the reader to compute from `value`.
#     bash tools/scale_sweep.sh [steps] [warmup]
set -u
STEPS=${1:-20}
WARM=${2:-3}
OUT=$PWD/gpurun_out/scale
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  [ "$N" -le "$NDEV" ] || { echo "# $N GPUs: not on this node ($NDEV devices)"; continue; }
  timeout 900 python bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline --no-latency | tee "$OUT/weak_2pow20_n$N.json"
  timeout 900 python bench.py --gpus $N --total-log2n 20 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-latency | tee "$OUT/strong_2pow20_n$N.json"
  timeout 900 python bench.py --gpus $N --total-log2n 24 --steps 6 --warmup 2 --no-cpu-baseline --no-latency | tee "$OUT/strong_2pow24_n$N.json"
done
