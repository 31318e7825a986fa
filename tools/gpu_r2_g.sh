#!/bin/bash
# round 2, window-table experiment: parity first, then timings next to the table-less engine on the same box
cd /root/repo
mkdir -p gpurun_out/r02g
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "window_table or cached_bases" 2>&1 | tail -8 > gpurun_out/r02g/pytest.txt
cat gpurun_out/r02g/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02g/bench_bls.json 2> gpurun_out/r02g/bench_bls.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r02g/bench_bls.json').read().strip().split('\n')[-1])
print('headline', d['value'], d['ms_per_step'], d.get('latency_ms_blocking'))
print(json.dumps(d.get('cached_bases'), indent=1))
PY
tail -3 gpurun_out/r02g/bench_bls.err
