#!/usr/bin/env python3
"""Developer probe: device timeline of ONE set_target-side build (kernels and copies with the idle gaps between them), from a
rocprofv3 rocpd database of `tools/build_timeline.py run <what> <n>`:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d out -o r -- python tools/build_timeline.py run index 1.06e6
    python tools/build_timeline.py show out/.../r_results.db
what = index | normals | voxels | scan.  The run sleeps 30 ms between builds; `show` prints the last group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(what, n):
    import numpy as np
    from point_cloud_registration_amd import _capi
    from point_cloud_registration_amd.synthetic import street, street_tiled
    ctx = _capi.get_context(0)
    pts = street(n) if n <= 2_000_000 else street_tiled(n)
    walls = []
    keep = _capi.Target.points(ctx, pts) if what == "normals" else None
    for r in range(8):
        time.sleep(0.03)
        t0 = time.perf_counter()
        if what == "index":
            t = _capi.Target.points(ctx, pts)
        elif what == "normals":
            keep.estimate_normals(15, compat=n <= 2_000_000, want=False); t = None
        elif what == "voxels":
            t = _capi.Target.voxels(ctx, pts, 1.0, 10)
        else:
            t = _capi.Scan(ctx, pts)
        ctx.synchronize()
        walls.append(time.perf_counter() - t0)
        if t is not None:
            t.close()
    print(f"{what} n={n} host wall ms:", " ".join(f"{1e3 * w:.3f}" for w in walls), flush=True)


def show(db):
    import re, sqlite3
    cur = sqlite3.connect(db).cursor()
    ev = []
    try:
        ev += [(s, e, re.sub(r"\(.*", "", re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", nm)).replace("void ", "")[:70])
               for nm, s, e in cur.execute("select name, start, end from kernels").fetchall()]
    except sqlite3.Error as ex:
        print("kernels view:", ex, [r[0] for r in cur.execute("select name from sqlite_master").fetchall()][:60])
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(memory_copies)").fetchall()]
        nm = "name" if "name" in cols else cols[0]
        ev += [(s, e, f"copy {k} {b} B") for k, s, e, b in cur.execute(f"select {nm}, start, end, size from memory_copies").fetchall()]
    except sqlite3.Error as ex:
        print("memory_copies view:", ex)
    ev.sort()
    groups, cur_g = [], []
    for s, e, nm in ev:
        if cur_g and s - cur_g[-1][1] > 5_000_000:
            groups.append(cur_g); cur_g = []
        cur_g.append((s, e, nm))
    if cur_g:
        groups.append(cur_g)
    g = groups[-1]
    t0, busy, prev = g[0][0], 0, None
    print(f"{len(groups)} groups; last: {len(g)} events, span {(g[-1][1] - t0) / 1e3:.1f} us")
    for s, e, nm in g:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        busy += e - s
        print(f"{(s - t0) / 1e3:9.1f} us  +{gap:7.1f} idle  {(e - s) / 1e3:8.1f} us  {nm}")
        prev = max(prev or e, e)
    print(f"busy {busy / 1e3:.1f} us of {(g[-1][1] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(float(sys.argv[3])))
    else:
        show(sys.argv[2])
