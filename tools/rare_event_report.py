#!/usr/bin/env python3
"""Companion of tools/rare_event_soak.py: in a rocprofv3 rocpd database, list the kernel dispatches that took longer than
1 ms and the gaps longer than 2 ms between consecutive dispatches (name, start, duration / gap).   rare_event_report.py run.db"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
cand = [t for t in tables if "kernel" in t.lower()]
print("kernel tables / views:", cand[:12])
src = next((t for t in ("kernels", "kernel_dispatch", "rocpd_kernel_dispatch") if t in tables), None) or (cand[0] if cand else None)
if not src:
    sys.exit("no kernel dispatch table")
cols = [r[1] for r in cur.execute(f"pragma table_info({src})").fetchall()]
print(src, "columns:", cols)
name = "name" if "name" in cols else next((c for c in cols if "name" in c), cols[0])
start = "start" if "start" in cols else next((c for c in cols if "start" in c), None)
end = "end" if "end" in cols else next((c for c in cols if c.startswith("end")), None)
rows = cur.execute(f"select {name}, {start}, {end} from {src} order by {start}").fetchall()
print("dispatches:", len(rows))
prev_end = None
long_k, gaps = [], []
for nm, s, e in rows:
    if e - s > 1_000_000: long_k.append((str(nm)[:60], s, round((e - s) / 1e6, 3)))
    if prev_end is not None and s - prev_end > 2_000_000: gaps.append((str(nm)[:60], s, round((s - prev_end) / 1e6, 3)))
    prev_end = max(prev_end or e, e)
print("kernels longer than 1 ms (name, start ns, ms):", long_k[:40])
print("gaps longer than 2 ms before a dispatch (name, start ns, gap ms):", gaps[:40])
