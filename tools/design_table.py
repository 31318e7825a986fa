#!/usr/bin/env python3
"""Print the round's measurement table from profiles/bench_<tag>*.json: python tools/design_table.py r03"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def L(name):
    return json.load(open(os.path.join(P, name)))


def row(label, d):
    sb = d.get("stage_ms_blocking", {})
    st = " · ".join(f"{sb.get(k, 0):.2f}" for k in ("digits", "sort", "accumulate", "merge", "reduce"))
    cb = d.get("cached_bases", {})
    cbs = f"{cb['records']['ms_per_step']:.2f} → {cb['window_table']['ms_per_step']:.2f} (c = {cb['window_table']['window_bits']})" if cb else "—"
    im = d["roofline"].get("int_mad", {}).get("frac", 0)
    tr = d["roofline"].get("traffic")
    return (f"| {label} | {d['value'] / 1e6:.1f} M | {d['ms_per_step']:.2f} | {d.get('latency_ms_blocking', 0):.2f} | {d.get('hostptr_ms', 0):.2f} | "
            f"{d['config']['window_bits']} / {d['config']['windows']} | {st} | {im:.2f} | {tr / 1e9 if tr else 0:.1f} GB | {cbs} |")


print("| config | pairs/s | ms per MSM (two in flight) | blocking call | host pointers | c / W | blocking call: digits+convert · sort · accumulate · merge · reduce (ms) | `int_mad.frac` | `traffic` per `k_accum` launch | cached bases: records → window table (ms per MSM) |")
print("|---|---|---|---|---|---|---|---|---|---|")
print(row("**BLS12-381 𝔾₁, 2²⁰ (headline)**", L(f"bench_{tag}.json")))
for k, sup in ((16, "2¹⁶"), (17, "2¹⁷"), (18, "2¹⁸"), (19, "2¹⁹"), (22, "2²²"), (24, "2²⁴")):
    print(row(f"BLS12-381 𝔾₁, {sup}", L(f"bench_{tag}_bls12_381_g1_2pow{k}.json")))
for c, lab in (("bn254_snarks_g1", "BN254-Snarks 𝔾₁, 2²² (Halo2-ZAL config)"), ("pallas", "Pallas, 2²⁰"), ("vesta", "Vesta, 2²⁰"), ("bls12_381_g2", "BLS12-381 𝔾₂, 2²⁰")):
    print(row(lab, L(f"bench_{tag}_{c}.json")))
