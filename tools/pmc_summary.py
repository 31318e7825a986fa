#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output) into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <key> <out.json> <out.txt>

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts every 128-B request as 64 B, so
read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 as is.  In-run calibration: k_part_count reads exactly
N x 32 B of scalars, k_group_sort writes W x N x 4 B of entries (+ the bucket_start array).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main():
    fd, wd, key, out_json, out_txt = sys.argv[1:6]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fe) | set(wr), key=lambda k: -(sum(fe.get(k, [0])) / max(1, len(fe.get(k, [0]))))):
        f = fe.get(k, [])
        w = wr.get(k, [])
        rows.append((k, max(len(f), len(w)), sum(f) / len(f) if f else 0.0, sum(w) / len(w) if w else 0.0))
    with open(out_txt, "w") as o:
        o.write("kernel, calls, FETCH_SIZE KiB (raw avg/launch), WRITE_SIZE KiB (raw avg/launch), "
                "corrected bytes/launch (2*FETCH*1024 + WRITE*1024)\n")
        for k, n, f, w in rows:
            o.write(f"{k}, {n}, {f:.1f}, {w:.1f}, {int(2 * f * 1024 + w * 1024)}\n")
    acc = next((r for r in rows if "k_accum" in r[0]), None)
    cal_r = next((r for r in rows if r[0].endswith("k_part_count")), None)
    cal_w = next((r for r in rows if "k_group_sort" in r[0]), None)
    doc = {}
    if os.path.exists(out_json):
        try:
            doc = json.load(open(out_json))
        except Exception:
            doc = {}
    if acc:
        doc[key] = int(2 * acc[2] * 1024 + acc[3] * 1024)
        doc.setdefault("_detail", {})[key] = {
            "kernel": acc[0], "launches_averaged": acc[1], "FETCH_SIZE_KiB_raw": acc[2], "WRITE_SIZE_KiB_raw": acc[3],
            "calibration": {"k_part_count_FETCH_SIZE_KiB_raw": cal_r[2] if cal_r else None,
                            "k_group_sort_WRITE_SIZE_KiB_raw": cal_w[3] if cal_w else None},
        }
        doc["_correction"] = ("gfx950 rocprofv3 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section): "
                              "fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 as is. In-run calibration: k_part_count reads "
                              "N x 32 B (2^20 pairs: 33.55 MB), k_group_sort writes W x N x 4 B + bucket_start (69.2 MB).")
        doc["_commands"] = [
            "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
            "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
        ]
    # what the figures were measured on: bench.py refuses them for any other kernel sources (it computes the same digest)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_digest
    import time
    doc["_sources_sha256"] = kernel_sources_digest()
    doc["_collected"] = time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime())
    json.dump(doc, open(out_json, "w"), indent=1)
    print(open(out_txt).read())


if __name__ == "__main__":
    main()
