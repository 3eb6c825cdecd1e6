#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 5 from the bench lines under profiles/ (<tag>_bench_<config>.json; previous round for the delta)."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
prev_tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
def load(t, c):
    f = os.path.join(REPO, "profiles", f"{t}_bench_{c}.json")
    if not os.path.exists(f):
        return None
    return json.loads(open(f).read().strip().splitlines()[-1])
print(f"| config | scan pts / GPU | ms / pass | M corr/s | {prev_tag} | search ms | reduce ms | B_alg GB/s (of 8 TB/s) | live traffic / pass (× B_alg) | fresh target: `set_target` + first `align` ms (iterations) | warm `align` ms |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for c in ("plane_b01", "icp_b01", "icp_b01_harness", "plane_b01_100k", "vplane_b01_harness", "ndt_b01_harness", "vplane_10m", "ndt_10m", "plane_100m",
          "plane_b01_resampled", "plane_b01_crop", "plane_100m_resampled", "plane_lidar", "icp_lidar_harness"):
    d = load(tag, c)
    if d is None:
        continue
    p = load(prev_tag, c)
    k = d["kernels"]
    nn = k.get("nn", k.get("linearize", {})).get("avg_ms")
    red = k.get("reduce", {}).get("avg_ms")
    s = d.get("seam", {})
    cfg = d["config"]
    tr = d["roofline"].get("traffic")
    alg = d["roofline"]["algorithmic_bytes_per_launch"]
    cold = "–" if cfg.get("first_align_ms") is None else f"{cfg['set_target_ms']:.2f} + {cfg['first_align_ms']:.3f} ({cfg['first_align_iterations']})"
    print(f"| `{c}` | {cfg['scan_points_per_gpu'] / 1e6:.2f} M | {d['ms_per_step']:.4f} | **{d['value']:.0f}** | {(str(round(p['value'])) if p else '–')} | "
          f"{nn:.4f}" + (" (fused)" if "linearize" in k else "") + f" | {(f'{red:.4f}' if red else '—')} | "
          f"{d['roofline']['achieved']:.0f} ({100 * d['roofline']['frac']:.1f} %) | "
          + (f"{tr / 1e6:.0f} MB ({tr / alg:.1f} ×)" if tr else "–") + f" | {cold} | {s.get('align_ms', '–')} ({s.get('align_iterations', '–')}) |")
