#!/usr/bin/env python3
"""profiles/pmc_summary.json from the rocprofv3 PMC passes collected by tools/collect_profiles.sh
(gpurun_out/<tag>_pmc_fetch.txt / _pmc_write.txt, copied to profiles/).  Bytes per calc_H_g_e2 pass leaving
L2 (fabric side: Infinity-Cache hits are counted, so an upper bound on HBM traffic), corrected as
MI355X_MICROARCH.md prescribes: counters in KB, FETCH_SIZE x2 on gfx950.  The summary records the hash of
the HIP sources it was collected on; bench.py only reports `roofline.traffic` when that hash matches."""
import json, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench

HOT = ("k_nn_scan", "k_nn_filter", "k_nn_fix", "k_certify", "k_nn_coop", "k_reduce_finalize", "k_reduce", "k_linearize", "k_finalize", "k_gn_update")
ROUND = os.environ.get("PCR_PROFILE_ROUND", "r04")


def parse(path, counter):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"^(.*?)\s+%s\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$" % counter, line.rstrip())
        if m and any(h in m.group(1) for h in HOT):
            name = re.sub(r"\(.*", "", m.group(1).replace("void ", "")).strip()
            out[name] = {"dispatches": int(m.group(2)), "kb": float(m.group(3)), "avg_us": float(m.group(4))}
    return out


def main():
    commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    summary = {"_comment": __doc__.replace("\n", " "), "kernel_source_hash": bench.kernel_source_hash(), "commit": commit}
    for cfg in ("plane_b01", "icp_b01", "icp_b01_harness", "plane_b01_100k", "vplane_b01_harness", "ndt_b01_harness", "vplane_10m", "ndt_10m",
                "plane_100m", "plane_b01_resampled"):
        tag = f"{ROUND}_{cfg}"
        f = parse(os.path.join(REPO, "profiles", tag + "_pmc_fetch.txt"), "FETCH_SIZE")
        w = parse(os.path.join(REPO, "profiles", tag + "_pmc_write.txt"), "WRITE_SIZE")
        if not f:
            continue
        by = {}
        total = 0.0
        # one pass = one launch of the kernel that folds (k_reduce_finalize / k_linearize_finalize); a pass runs ONE of the
        # k_nn_scan instantiations (tile hand-out chosen per launch), so the per-pass figure weights every kernel's
        # average by its share of the launches
        passes = max([v["dispatches"] for k, v in f.items() if "finalize" in k] or [1])
        for k in sorted(set(f) | set(w)):
            fb = f.get(k, {}).get("kb", 0.0) * 1024 * 2          # FETCH_SIZE reads half on gfx950
            wb = w.get(k, {}).get("kb", 0.0) * 1024
            n = f.get(k, w.get(k, {})).get("dispatches", 0)
            by[k] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "avg_us": f.get(k, w.get(k, {})).get("avg_us"), "launches": n}
            if "gn_update" in k:
                continue                                          # (the trajectory align of the bench set-up, not a pass)
            total += (fb + wb) * n / passes
        summary[cfg] = {"hbm_bytes_per_pass": round(total), "passes": passes, "by_kernel": by,
                        "source": [f"profiles/{tag}_pmc_fetch.txt", f"profiles/{tag}_pmc_write.txt"]}
    json.dump(summary, open(os.path.join(REPO, "profiles", "pmc_summary.json"), "w"), indent=2)
    print(json.dumps(summary, indent=1)[:1500])


if __name__ == "__main__":
    main()
