#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 5.1 from the bench lines under profiles/ (r03_bench_<config>.json)."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
prev = {"plane_b01": 6776, "icp_b01": 5961, "icp_b01_harness": 2013, "plane_b01_100k": 1328, "vplane_10m": 8609, "ndt_10m": 14111, "plane_100m": 4181}
print("| config | scan pts / GPU | ms / pass (median) | M corr/s | r02 M corr/s | search ms | reduce ms | B_alg GB/s (frac of 8 TB/s) | `align` ms (iterations) | class seam: array / handle ms |")
print("|---|---|---|---|---|---|---|---|---|---|")
for c in ("plane_b01", "icp_b01", "icp_b01_harness", "plane_b01_100k", "vplane_10m", "ndt_10m", "plane_100m",
          "plane_b01_resampled", "plane_b01_crop", "plane_100m_resampled"):
    f = os.path.join(REPO, "profiles", f"{tag}_bench_{c}.json")
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["kernels"]
    nn = k.get("nn", k.get("linearize", {})).get("avg_ms")
    cert = k.get("certify", {}).get("avg_ms")
    red = k.get("reduce", {}).get("avg_ms")
    s = d.get("seam", {})
    seam = f"{s.get('calc_H_g_e2_array_ms_per_call', '–')} / {s.get('calc_H_g_e2_handle_ms_per_call', '–')}"
    nn_s = f"{nn:.3f}" + (" (fused kernel)" if "linearize" in k else "") + (f" (+ certify {cert:.3f} on its list passes)" if cert else "")
    print(f"| `{c}` | {d['config']['scan_points_per_gpu'] / 1e6:.2f} M | {d['ms_per_step']:.3f} | **{d['value']:.0f}** | {prev.get(c, '–')} | {nn_s} | "
          f"{(f'{red:.3f}' if red else '—')} | {d['roofline']['achieved']:.0f} ({100 * d['roofline']['frac']:.1f} %) | "
          f"{s.get('align_ms', '–')} ({s.get('align_iterations', '–')}) | {seam} |")
