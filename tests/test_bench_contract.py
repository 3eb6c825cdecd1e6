"""bench.py pieces that do not need a GPU: configuration table, argument defaults, and the
cpu_baseline leg (the oracle timed on a bounded sample)."""

import importlib.util
import json
import os
import sys

import numpy as np

from conftest import REPO


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_configs_match_baseline_json():
    bench = _load_bench()
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert "correspondences/sec" in base["metric"]
    # one bench config per BASELINE.json config line
    assert {"icp_b01_harness", "plane_b01", "vplane_10m", "ndt_10m", "plane_100m"} <= set(bench.CONFIGS)
    assert bench.CONFIGS["icp_b01_harness"][0] == "icp" and bench.CONFIGS["icp_b01_harness"][2] == 100_000
    assert bench.CONFIGS["plane_b01"][0] == "plane" and bench.CONFIGS["plane_b01"][1] == 1_060_000
    assert bench.CONFIGS["vplane_10m"][3] == 0.5 and bench.CONFIGS["ndt_10m"][3] == 1.0
    assert bench.B_ALG == {"icp": 24, "plane": 36, "vplane": 36, "ndt": 48}
    assert bench.HBM_PEAK_GBS == 8000.0
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
    finally:
        sys.argv = old
    assert a.gpus == 1 and a.config == "plane_b01" and a.steps > 0 and a.warmup >= 0


def test_cpu_baseline_leg_runs_on_the_oracle():
    bench = _load_bench()
    from point_cloud_registration_amd.synthetic import street, perturbed_scan
    target = street(20000, seed=1)
    scan, _ = perturbed_scan(target, 5000, seed=2)
    out = bench.cpu_baseline("icp", target, scan, None, [np.eye(4)], 2.0, None, 1)
    assert out["kind"] == "port" and out["unit"] == "Mcorr/s" and out["value"] > 0 and out["cores"] >= 1
    out = bench.cpu_baseline("ndt", target, scan, None, [np.eye(4)], 2.0, 1.0, 1)
    assert out["value"] > 0 and "sample" in out
