#!/usr/bin/env python3
"""Dump a window of consecutive kernel dispatches from a rocprofv3 rocpd database -- start, duration, hardware queue, kernel -- taken
from the middle of the run (the pipelined loop of tools/cu_mask_sweep.py / tools/sweep.py), so that what runs BESIDE what is visible:

    python tools/timeline_window.py <db> [first_fraction=0.3] [count=70]"""
import sqlite3
import sys


def main(path, frac=0.3, count=70):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "d.queue_id" if "queue_id" in cols else "0"
    scol = "d.stream_id" if "stream_id" in cols else "0"
    rows = list(cur.execute(
        f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, {qcol}, {scol} "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    i0 = int(len(rows) * frac)
    # start the window at the first sort kernel at or after i0 (the first launch of an MSM)
    while i0 < len(rows) and "k_part_count" not in rows[i0][0]:
        i0 += 1
    win = rows[i0:i0 + count]
    if not win:
        print("no dispatches")
        return
    t0 = win[0][1]
    print(f"# {len(rows)} dispatches in the run; window of {len(win)} from #{i0}; columns: start us, duration us, end us, queue, stream, grid, kernel")
    for r in win:
        name = r[0].split("(")[0][:60]
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f} {(r[2]-t0)/1e3:9.1f}  q{r[6]} s{r[7]}  {r[3]}x{r[4]}/{r[5]}  {name}")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], float(a[2]) if len(a) > 2 else 0.3, int(a[3]) if len(a) > 3 else 70)
