#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02g
for K in "" 16 24 32; do
  TABLE_K=$K timeout 600 python tools/bench_table.py bls12_381_g1 16 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02g/table_bls_2pow16_K.jsonl
done
for K in "" 16 32; do
  TABLE_K=$K timeout 600 python tools/bench_table.py bls12_381_g1 12 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02g/table_bls_2pow12_K.jsonl
done
for K in "" 48 64; do
  TABLE_K=$K timeout 600 python tools/bench_table.py bls12_381_g1 18 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02g/table_bls_2pow18_K.jsonl
done
