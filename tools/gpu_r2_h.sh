#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for v in "" _ns4 _ns6; do
for cfg in "bn254_snarks_g1 22" "bls12_381_g1 20"; do
set -- $cfg
rm -rf /tmp/prof_s && mkdir -p /tmp/prof_s
( cd /tmp && CTT_MSM_HIP_LIB=/root/repo/constantine_amd/libctt_msm_hip$v.so rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python /root/repo/bench.py --curve $1 --log2n $2 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > /tmp/prof_s/out.json 2> /tmp/prof_s/err.log )
DB=$(find /tmp/prof_s -name "*.db" | head -1)
echo "variant '$v' $1 $2"
python tools/kernel_timeline.py "$DB" 2>&1 | grep -E "part_scatter"
done
done
