#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_symbols or g2_vs or cached or window_table_cached or infinity or fr_coefs" 2>&1 | tail -2
for cfg in "bn254_snarks_g1 22" "bls12_381_g1 20" "bls12_381_g2 18"; do
set -- $cfg
rm -rf /tmp/prof_s && mkdir -p /tmp/prof_s
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python /root/repo/bench.py --curve $1 --log2n $2 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > /tmp/prof_s/out.json 2> /tmp/prof_s/err.log )
DB=$(find /tmp/prof_s -name "*.db" | head -1)
echo "$1 $2"
python tools/kernel_timeline.py "$DB" 2>&1 | grep -E "convert" | grep -v dur
done
