#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) output as text: per-kernel calls / avg duration (--stats view)
and per-kernel averages of any PMC counters collected.  Usage: rocpd_summary.py run.db [more.db ...]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"radix_sort_onesweep_iteration|onesweep_histograms|lookback_scan|run_length|partition", name)
    if name.startswith("void rocprim") and m:
        return "rocprim::" + m.group(0) + "<...>"
    return name if len(name) < 110 else name[:107] + "..."


for path in sys.argv[1:]:
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f"== {path}")
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    if rows:
        print(f"{'kernel':<112} {'calls':>6} {'total_ms':>12} {'avg_ms':>10} {'%':>6}")
        agg = {}
        for name, calls, tot, avg, pct in rows:
            k = short(name)
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += calls; a[1] += tot; a[2] += pct
        for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:<112} {calls:>6} {tot / 1000.0:>12.3f} {tot / calls / 1000.0:>10.3f} {pct:>6.2f}")
    try:
        pm = cur.execute("select name, counter_name, count(*), avg(counter_value), sum(counter_value), avg(duration) "
                         "from pmc_events group by name, counter_name").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print(f"{'kernel':<112} {'counter':>12} {'n':>5} {'avg_value':>14} {'avg_dur_us':>11}")
        for name, cn, n, avg, tot, dur in sorted(pm, key=lambda r: -r[4]):
            print(f"{short(name):<112} {cn:>12} {n:>5} {avg:>14.3f} {dur / 1000.0:>11.3f}")
