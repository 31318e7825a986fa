#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (csv output) for one kernel: average counter value per launch + derived readings.

    python tools/sq_summary.py <kernel substring> <mixed adds per launch> <dir> [<dir> ...]

Counters are the guide's SQ set (MI355X_MICROARCH.md): SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY, and GRBM_GUI_ACTIVE SQ_INSTS_SALU in a second pass.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    kern, madds = sys.argv[1], float(sys.argv[2])
    acc = defaultdict(list)
    dur = []
    for d in sys.argv[3:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"]:
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    v = {k: sum(x) / len(x) for k, x in acc.items()}
    for k in sorted(v):
        print(f"{k} = {v[k]:.0f}   (avg of {len(acc[k])} launches)")
    if dur:
        print(f"dur_us_under_pmc = {sum(dur) / len(dur):.0f}")
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        print("# per-wave view (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES):")
        for k, label in (("SQ_ACTIVE_INST_ANY", "issuing instructions"), ("SQ_WAIT_INST_ANY", "issue stalls (dependency / pipe)"),
                         ("SQ_WAIT_ANY", "parked on s_waitcnt (gather, bucket_start loads)")):
            if k in v:
                print(f"#   {label}: {100 * v[k] / wc:.1f}% of wave cycles")
    if "SQ_INSTS_VALU" in v and madds > 0:
        print(f"#   VALU wave-instructions per launch {v['SQ_INSTS_VALU']:.4g} = {v['SQ_INSTS_VALU'] * 64 / madds:.0f} per mixed addition (one lane)")
    if "SQ_INSTS_SALU" in v and madds > 0:
        print(f"#   SALU wave-instructions per launch {v['SQ_INSTS_SALU']:.4g} = {v['SQ_INSTS_SALU'] * 64 / madds:.0f} per mixed addition")
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # SQ_ACTIVE_INST_VALU counts, per SIMD, the cycles (in units of four) in which a VALU instruction is executing; the chip has
        # 8 XCDs x 32 CUs x 4 SIMDs, GRBM_GUI_ACTIVE is summed over the XCDs
        simd_quad_cycles = v["GRBM_GUI_ACTIVE"] / 8 * 1024 / 4
        print(f"#   VALU busy: SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs / 4) = {100 * v['SQ_ACTIVE_INST_VALU'] / simd_quad_cycles:.1f} % of the SIMD cycles")
    if "GRBM_GUI_ACTIVE" in v and dur:
        print(f"#   effective clock under the profiler: GRBM_GUI_ACTIVE / 8 XCDs / duration = {v['GRBM_GUI_ACTIVE'] / 8 / (sum(dur) / len(dur)) / 1e3:.2f} GHz")


if __name__ == "__main__":
    main()
