#!/usr/bin/env python3
"""Per-counter averages over the LAST n dispatches of a kernel in a rocprofv3 rocpd database (the passes a probe runs at
one pose, behind the align() that produced the trajectory).   rocpd_last.py run.db <kernel substring> <n>"""
import sqlite3, sys
db, sub, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)").fetchall()]
key = "start" if "start" in cols else ("dispatch_id" if "dispatch_id" in cols else "rowid")
rows = cur.execute(f"select {key}, counter_name, counter_value, duration from pmc_events where name like ? order by {key}", (f"%{sub}%",)).fetchall()
if not rows:
    print("no rows for", sub, "columns:", cols); sys.exit(0)
disp = {}
for k, cn, v, d in rows:
    e = disp.setdefault(k, {"dur": d, "c": {}})
    e["c"][cn] = e["c"].get(cn, 0.0) + v                      # summed over the counter's instances
keys = sorted(disp)[-n:]
names = sorted({c for k in keys for c in disp[k]["c"]})
print(f"{sub}: last {len(keys)} of {len(disp)} dispatches, avg duration {sum(disp[k]['dur'] for k in keys) / len(keys) / 1000.0:.1f} us")
for c in names:
    vals = [disp[k]["c"].get(c, 0.0) for k in keys]
    print(f"  {c:<28} sum over instances, avg per dispatch {sum(vals) / len(vals):.1f}")
