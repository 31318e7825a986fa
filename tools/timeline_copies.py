#!/usr/bin/env python3
"""Kernels and memory copies of the last call in a rocprofv3 rocpd database (--kernel-trace --memory-copy-trace), sorted by start:
what the link and the GPU do during one host-pointer MSM (profiles/hostptr_timeline_r04.txt).  python tools/timeline_copies.py <db>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if 'memory_copy' in t]
print("tables:", mc)
rows = [(r[1], r[2], "K " + r[0].split("(")[0][:60]) for r in cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id")]
for t in mc:
    if t.startswith('rocpd_memory_copy'):
        cols = [c[1] for c in cur.execute(f"pragma table_info({t})")]
        print(t, cols)
        try:
            for r in cur.execute(f"select start, end, size from {t}"):
                rows.append((r[0], r[1], f"COPY {r[2]} B"))
        except Exception as e:
            print("err", e)
        break
rows.sort()
# last MSM: from the last big gap (> 3 ms idle) onward
if len(sys.argv) > 2:      # python tools/timeline_copies.py <db> <ms>: everything in the last <ms> milliseconds
    t_end = max(r[1] for r in rows)
    first = next(i for i, r in enumerate(rows) if r[0] >= t_end - float(sys.argv[2]) * 1e6)
else:
    starts = [i for i in range(1, len(rows)) if rows[i][0] - max(r[1] for r in rows[max(0,i-5):i]) > 300_000]
    first = starts[-1] if starts else 0
t0 = rows[first][0]
for s, e, nm in rows[first:]:
    if "k_pyr" in nm: continue
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us  {nm}")
