#!/usr/bin/env python3
"""Print DESIGN.md section 5's table rows and the headline figures from the bench lines of one collection:  python tools/design_table.py [profiles]"""
import json
import os
import re
import sys

D = sys.argv[1] if len(sys.argv) > 1 else "profiles"
ROWS = [("**BLS12-381 𝔾₁, 2²⁰ (headline)**", "bench_r06.json"), ("BLS12-381 𝔾₁, 2¹⁶", "bench_r06_bls12_381_g1_2pow16.json"),
        ("BLS12-381 𝔾₁, 2¹⁷", "bench_r06_bls12_381_g1_2pow17.json"), ("BLS12-381 𝔾₁, 2¹⁸", "bench_r06_bls12_381_g1_2pow18.json"),
        ("BLS12-381 𝔾₁, 2¹⁹", "bench_r06_bls12_381_g1_2pow19.json"), ("BLS12-381 𝔾₁, 2²²", "bench_r06_bls12_381_g1_2pow22.json"),
        ("BLS12-381 𝔾₁, 2²⁴", "bench_r06_bls12_381_g1_2pow24.json"), ("BN254-Snarks 𝔾₁, 2²² (Halo2-ZAL config)", "bench_r06_bn254_snarks_g1.json"),
        ("Pallas, 2²⁰", "bench_r06_pallas.json"), ("Vesta, 2²⁰", "bench_r06_vesta.json"), ("BLS12-381 𝔾₂, 2²⁰", "bench_r06_bls12_381_g2.json")]


def line(f):
    return [json.loads(l) for l in open(os.path.join(D, f)) if l.startswith("{")][-1]


for name, f in ROWS:
    d = line(f)
    st, cb = d["stage_ms_blocking"], d.get("cached_bases") or {}
    cached = "—"
    if cb.get("records"):
        cached = f"{cb['records']['ms_per_step']:.2f} → {cb['window_table']['ms_per_step']:.2f} (c = {cb['window_table']['window_bits']})"
    print(f"| {name} | {d['value'] / 1e6:.1f} M | {d['ms_per_step']:.2f} | {d['latency_ms_blocking']:.2f} | {d['hostptr_ms']:.2f} | {d['config']['window_bits']} / {d['config']['windows']} | "
          f"{st['digits']:.2f} · {st['sort']:.2f} · {st['accumulate']:.2f} · {st['merge']:.2f} · {st['reduce']:.2f} | {d['roofline']['int_mad']['frac']:.2f} | {d['roofline']['traffic'] / 1e9:.1f} GB | {cached} |")
h = line("bench_r06.json")
r = h["roofline"]
print(f"\nheadline {h['value'] / 1e6:.1f} M pairs/s, {h['ms_per_step']:.3f} ms; k_accum {r['kernel_ms']:.3f} ms; frac {r['frac']:.4f}; peak_measured {r['peak_measured']:.0f} GB/s frac_measured {r['frac_measured']:.4f}; "
      f"int_mad {r['int_mad']['frac']:.2f}; blocking {h['value_blocking'] / 1e6:.1f} M ({h['latency_ms_blocking']:.3f} ms); hostptr {h['value_hostptr'] / 1e6:.1f} M ({h['hostptr_ms']:.2f} ms); "
      f"cpu {h['cpu_baseline']['value'] / 1e6:.2f} M ({h['cpu_baseline']['cores']} threads)")
u = line("bench_r06_under_rocprof.json")
print(f"under rocprof: {u['ms_per_step']:.3f} ms per step, k_accum {u['roofline']['kernel_ms'] * 1e3:.0f} us by HIP events")
for l in open(os.path.join(D, "rocprof_r06_kernel_stats.txt")):
    if re.search(r"k_accum|k_convert|k_pyrINS|k_merge_tail_queue|k_group_sort|k_part_scatter|k_part_count|k_part_scan|k_scan_u32|k_merge_queue_quad|k_merge_long", l) and re.search(r"\s\d+\s+\d+\.\d+\s+\d+\.\d+\s+\d+\.\d+\s+\d+\.\d+", l):
        print(l.rstrip()[:150])
