#!/usr/bin/env python3
"""Print the per-kernel timeline of the last MSM step from a rocprofv3 rocpd database
(rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python bench.py ...)  and a per-kernel summary.

    python tools/kernel_timeline.py <db> [warmup_msms]

warmup_msms: the per-kernel statistics leave out everything dispatched before that many MSMs have started (bench.py's
warm-up steps: their launches include first-touch allocations and cold caches; round 2's averages contained them)."""
import collections
import sqlite3
import sys


def main(path, warm=0):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = list(cur.execute(
        "select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    idx = [i for i, r in enumerate(rows) if "k_digits" in r[0]] or [i for i, r in enumerate(rows) if "k_part_count" in r[0] and (i == 0 or "k_part_count" not in rows[i - 1][0])]
    if not idx:
        print("no MSM dispatch found")
        return
    last = rows[idx[-1]:]
    t0 = last[0][1]
    print("== timeline of the last MSM step ==")
    for r in last:
        name = r[0].split("(")[0][:70]
        print(f"{(r[1]-t0)/1e3:9.1f} us  dur {(r[2]-r[1])/1e3:8.1f} us  grid {r[3]}x{r[4]} wg {r[5]}  {name}")
    first = idx[warm] if 0 < warm < len(idx) else 0
    print(f"== per-kernel stats over the {'timed steps (first ' + str(warm) + ' MSMs left out)' if first else 'whole run'} ==")
    agg = collections.defaultdict(list)
    for r in rows[first:]:
        agg[r[0].split("(")[0][:70]].append((r[2] - r[1]) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:72s} {len(v):6d} {sum(v):12.1f} {sum(v)/len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100*sum(v)/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
