#!/bin/bash
# Round-2 experiment F: four waves per SIMD for the 9-limb fields (same box, same run)
set -u
OUT=$PWD/gpurun_out/r2f
mkdir -p "$OUT"
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" _w4; do
  lib=$PWD/constantine_amd/libctt_msm_hip$v.so
  [ -f "$lib" ] || continue
  for cfg in "bn254_snarks_g1 22" "bn254_snarks_g1 20" "pallas 20" "vesta 20"; do
    set -- $cfg
    CTT_MSM_HIP_LIB=$lib timeout 300 python bench.py --curve $1 --log2n $2 --steps 30 --warmup 5 --no-cpu-baseline --no-latency \
        > "$OUT/bench${v}_$1_$2_$rep.json" 2>> "$OUT/bench.err"
    python - "$OUT/bench${v}_$1_$2_$rep.json" "${v:-_w3}" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"waves{sys.argv[2][2:]}  {d['config']['workload'][:30]:30s} {d['value']/1e6:7.1f} M pairs/s  {d['ms_per_step']:7.3f} ms/MSM  accumulate {d['stage_ms']['accumulate']:.3f} ms  K={d['config']['entries_per_lane']}")
except Exception as e:
    print("FAILED", sys.argv[1], e)
PY
  done
done
done
