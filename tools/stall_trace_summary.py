#!/usr/bin/env python3
"""Longest API calls of a rocprofv3 --hip-trace --hsa-trace rocpd database, and the calls nested inside the longest ones."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print("tables/views:", [t for t in tabs if "region" in t or "api" in t.lower()][:20])
view = "regions" if "regions" in tabs else None
if view is None:
    print("no regions view; all:", tabs); sys.exit(0)
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
print("columns:", cols)
rows = cur.execute(f"select name, start, end, (end - start) as d, tid from {view} order by d desc limit 15").fetchall()
for name, s, e, d, tid in rows:
    print(f"{d / 1e6:10.3f} ms  {name}  tid {tid}")
    if d > 1e6:
        inner = cur.execute(f"select name, (end - start) as d from {view} where start >= ? and end <= ? and tid = ? and not (start = ? and end = ?) order by d desc limit 6",
                            (s, e, tid, s, e)).fetchall()
        for n2, d2 in inner:
            print(f"      inside: {d2 / 1e6:10.3f} ms  {n2}")
