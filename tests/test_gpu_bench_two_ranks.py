"""bench.py's N > 1 branch end to end on the 1-GPU test box: ``python bench.py --gpus 2`` with NO launcher must spawn
its two ranks itself (torch.distributed.run on 127.0.0.1), both ranks share device 0, the process group runs on gloo
and the 29-double exchange takes the agreed host fallback (RCCL refuses two ranks on one device) -- the plumbing of
the driver's SCALE run (init, barrier + synchronize bracketing, max over ranks, per-rank kernel times, one JSON line
from rank 0), everything but the xGMI hops."""

import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(extra, timeout=900):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PCR_BENCH_SELF_LAUNCHED")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--config", "plane_b01_100k", "--steps", "5", "--warmup", "2",
           "--repeats", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_two_ranks_self_launched_on_one_gpu():
    line = _run(["--gpus", "2", "--backend", "gloo"])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["unit"] == "Mcorr/s"
    assert line["config"]["workload"] == "plane_b01_100k" and line["config"]["scan_points_per_gpu"] == 100_000
    assert line["config"]["scan_points_job"] == 200_000            # weak scaling: every rank its own scan
    assert line["config"]["backend"] == "gloo" and line["config"]["allreduce_transport"] in ("host-gloo", "rccl-in-stream")
    pr = line["per_rank_kernel_ms"]
    assert isinstance(pr, list) and len(pr) == 2 and all(isinstance(d, dict) and d for d in pr)
    assert line["value"] > 0 and line["ms_per_step"] > 0
    # value = whole-job units / max-over-ranks time
    assert abs(line["value"] - 200_000 / (line["ms_per_step"] * 1e-3) / 1e6) < 0.02 * line["value"]
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line   # rank-0-at-N=1-only legs stay out


def test_bench_strong_scaling_two_ranks():
    line = _run(["--gpus", "2", "--backend", "gloo", "--scaling", "strong"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["scan_points_job"] == 100_000 and line["config"]["scan_points_per_gpu"] == 50_000


def test_bench_reads_a_real_b01_pcd(tmp_path):
    """the path bench.py takes when data/B-01.pcd exists (benchmark/test_data.py:11,24) has now executed --
    a B-01-sized ``binary_compressed`` PCD (the stand-in's points, so the numbers are known), found through PCR_B01_PCD."""
    import numpy as np
    from point_cloud_registration_amd.io import save_pcd
    from point_cloud_registration_amd.synthetic import street
    p = tmp_path / "B-01.pcd"
    save_pcd(str(p), street(1_060_000, seed=0), compressed=True)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PCR_B01_PCD"] = str(p)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--config", "plane_b01", "--steps", "5", "--warmup", "1", "--repeats", "1",
           "--no-pmc", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert line["data"] == "B-01.pcd"
    assert line["config"]["target_points"] == 1_060_000 and line["config"]["scan_points_per_gpu"] == 1_060_000
    assert line["config"]["pose_error_m"] < 2e-3 and line["config"]["gauss_newton_iters_to_converge"] == 5
    assert np.isfinite(line["value"]) and line["value"] > 0


def test_bench_single_process_group_on_one_gpu():
    """``bench.py --gpus 2 --single-process``: the same JSON line from ONE process driving two contexts of GPU 0 through
    pcr_group_*  -- no torchrun, no torch.distributed, the in-process peer-to-peer exchange."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PCR_BENCH_SELF_LAUNCHED")}
    env["PCR_BENCH_GROUP_DEVICES"] = "0,0"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--config", "plane_b01", "--steps", "5", "--warmup", "2", "--repeats", "2",
           "--gpus", "2", "--single-process", "--no-pmc", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["parallelism"] == "single-process group x2"
    assert line["config"]["scan_points_job"] == 2 * 1_060_000 and line["config"]["allreduce_transport"] == "p2p-in-process"
    assert abs(line["value"] - 2 * 1_060_000 / (line["ms_per_step"] * 1e-3) / 1e6) < 0.02 * line["value"]
    assert line["config"]["pose_error_m"] < 2e-3
