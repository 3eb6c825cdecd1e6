#!/usr/bin/env python3
"""Headline benchmark: M-correspondences/s of the registration hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config plane_b01|icp_b01|icp_b01_harness|plane_b01_100k|vplane_10m|ndt_10m|plane_100m]

A "step" is ONE pass of the hot path -- one ``calc_H_g_e2``: float32 transform of the whole
scan shard, exact nearest-neighbour search against the target, the ``dist < max_dist`` gate,
residuals + Jacobians, and the reduction to the 6x6 Gauss-Newton normal equations (29 doubles
back on the host; with N > 1 GPUs an RCCL all-reduce of those 29 doubles sits in between).
Default workload = BASELINE.json configs[1]: Point-to-Plane ICP, B-01 stand-in target
(``street(1_060_000, seed=0)``, the .pcd itself is absent from the reference checkout) vs a
perturbed full-size scan; steps walk along a recorded Gauss-Newton trajectory.

Multi-GPU: one process per GPU, the target replicated, no data-path collective besides the 232-byte
all-reduce.  Either launch form works: ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``
(ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), or plain ``python bench.py --gpus N ...``, which re-executes
itself under ``torch.distributed.run`` on 127.0.0.1 with a free port (``self_launch``).  ``--backend nccl`` (RCCL, the
default) or ``gloo`` (host all-reduce: the agreed fallback transport, and what two ranks sharing ONE GPU use --
``tests/test_gpu_bench_two_ranks.py``).
``--scaling weak`` (default): every rank owns a scan shard of the SAME size (its own perturbed scan);
``--scaling strong``: ONE scan of the configured size, rank r takes ``shard_scan(scan, r, N)``.

Timing: ``--repeats R`` (default 5) timed blocks of exactly ``--steps`` passes each, every block
bracketed by barrier + synchronize on both sides, max over ranks; ``ms_per_step`` / ``value`` are the
MEDIAN block, min / max and all blocks are reported beside it.  HIP events bracket every 3rd pass of
those blocks (an event pair costs microseconds of stream time); one further block with the events off
is reported as ``ms_per_step_events_off``.

Rank 0 prints one JSON line (contract in the task statement) with ``roofline`` (HIP-event kernel
time measured here, live) and ``cpu_baseline`` (the CPU oracle on this box's host cores, bounded
sample, N = 1 only).
"""

import argparse
import hashlib
import json
import os
import sys
import time

# Host hygiene, before NumPy loads its BLAS: on a 256-CPU box inside a container with a 16-CPU quota the BLAS thread
# pool (one spinning thread per visible CPU after the first matmul) exhausts the cgroup's CPU bandwidth and the
# kernel parks EVERY thread of the process for the rest of the 100 ms period -- the "12-42 ms launch stalls" of
# round 2 (root-caused in round 3: profiles/archive/r03_stall_root_cause.txt).  The oracle's OpenMP pool (cpu_baseline) is
# not affected by this variable.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6300
B_ALG = {"icp": 24, "plane": 36, "vplane": 36, "ndt": 48}     # bytes / scan point / pass

CONFIGS = {
    # name: (kind, target points, scan points per GPU, voxel_size, description)
    "plane_b01": ("plane", 1_060_000, 1_060_000, None,
                  "Point-to-Plane ICP (PlaneICP, k=15 normals), B-01 stand-in street(1.06M) vs perturbed scan"),
    "icp_b01": ("icp", 1_060_000, 1_060_000, None, "Point-to-Point ICP, B-01 stand-in street(1.06M) vs perturbed scan"),
    "plane_b01_100k": ("plane", 1_060_000, 100_000, None,
                       "Point-to-Plane ICP, B-01 stand-in, 100k-point perturbed scan"),
    # BASELINE.json configs[0]: the reference harness' own case (benchmark/test_data.py:21-44)
    "icp_b01_harness": ("icp", 1_060_000, 100_000, None,
                        "Point-to-Point ICP, B-01 stand-in, reference-harness scan: 100k random points shifted "
                        "by t=(0,0,0.3) + N(0,0.005) noise, init_T = I"),
    # the reference's own benchmark for the voxel methods (benchmark/speed_test_comparison.py:36-55,166-170): map 1.06 M
    # points, voxel_size 1, the harness' 100 k-point scan
    "vplane_b01_harness": ("vplane", 1_060_000, 100_000, 1.0,
                           "VPlaneICP voxel_size=1.0, B-01 stand-in, reference-harness scan (100k points, t=(0,0,0.3) + noise)"),
    "ndt_b01_harness": ("ndt", 1_060_000, 100_000, 1.0,
                        "NDT voxel_size=1.0, B-01 stand-in, reference-harness scan (100k points, t=(0,0,0.3) + noise)"),
    "vplane_10m": ("vplane", 10_000_000, 10_000_000, 0.5, "VPlaneICP voxel_size=0.5, synthetic 10M-pt cloud"),
    "ndt_10m": ("ndt", 10_000_000, 10_000_000, 1.0, "NDT voxel_size=1.0, synthetic 10M-pt cloud"),
    "plane_100m": ("plane", 100_000_000, 12_500_000, None,
                   "Point-to-Plane ICP, synthetic 100M-pt target, 12.5M scan points per GPU"),
    # scans that are NOT copies of target points: the same surfaces sampled independently (what a real
    # sweep is: matches sit at ~half the point spacing instead of at the noise level), and a 30 %-overlap crop
    "plane_b01_resampled": ("plane", 1_060_000, 1_060_000, None,
                            "Point-to-Plane ICP, B-01 stand-in vs an INDEPENDENT sample of the same surfaces (street seed != 0)"),
    "plane_b01_crop": ("plane", 1_060_000, 1_060_000, None,
                       "Point-to-Plane ICP, B-01 stand-in vs an independent sample of the street and of the next block, "
                       "cropped to x in [24, 144] m: 30 % of the scan overlaps the map"),
    "plane_100m_resampled": ("plane", 100_000_000, 12_500_000, None,
                             "Point-to-Plane ICP, synthetic 100M-pt target vs an independent 12.5M-pt sample of the same surfaces"),
    # NON-UNIFORM density: one revolution of a 64-beam LiDAR standing in the same street -- density falls
    # like 1/r^2, the ground is a set of ring lines, most of space is empty (synthetic.lidar_sweep).  Scan protocols as above.
    "plane_lidar": ("plane", 1_060_000, 1_060_000, None,
                    "Point-to-Plane ICP, lidar_sweep(1.06M) map (density ~ 1/r^2) vs its full perturbed scan"),
    "icp_lidar_harness": ("icp", 1_060_000, 100_000, None,
                          "Point-to-Point ICP, lidar_sweep(1.06M) map, reference-harness scan (100k points, t=(0,0,0.3) + noise)"),
}
SCAN_FAMILY = {"plane_b01_resampled": "resampled", "plane_b01_crop": "crop", "plane_100m_resampled": "resampled"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="plane_b01", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-passes", type=int, default=3)
    ap.add_argument("--variant", type=int, default=None, help="0 fused kernel, 1 NN + reduce kernels, 2 per launch (default)")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps passes; the median is reported")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--event-period", type=int, default=3, help="HIP events around every n-th pass")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two live rocprofv3 --pmc passes behind roofline.traffic (also: PCR_BENCH_NO_PMC=1)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)     # the short run rocprofv3 wraps
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process: a pcr_group (one context + host thread per GPU, in-process peer-to-peer "
                         "exchange; Registration(devices=[...])) instead of one rank per GPU; PCR_BENCH_GROUP_DEVICES=0,0 picks the ids")
    ap.add_argument("--backend", default=os.environ.get("PCR_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 plumbing (barrier, max over ranks); nccl = RCCL")
    return ap.parse_args()


def effective_cpus():
    """CPUs this process may actually use: the cgroup CPU-bandwidth quota (cpu.max, v2; cfs_quota, v1) capped by the
    affinity mask -- NOT os.cpu_count(), which reports the 256 visible CPUs of a GPU box whose container has 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(int(quota), 1))
    return max(n, 1)


HOT_KERNELS = ("k_nn_scan", "k_nn_filter", "k_nn_fix", "k_nn_coop", "k_certify", "k_reduce_finalize", "k_linearize_finalize",
               "k_reduce<", "k_linearize<", "k_finalize")


def live_traffic(args):
    """roofline.traffic measured by THIS run: the same workload is re-executed twice as a short child process under
    ``rocprofv3 --kernel-trace --pmc FETCH_SIZE`` and ``... --pmc WRITE_SIZE`` (separate passes, as MI355X_MICROARCH.md
    prescribes for the TCC counters; nothing else is traced), the rocpd database is read back and the bytes of the
    hot-path kernels are averaged per pass (= per launch of the kernel that folds).  Corrections of the guide: the
    counters are in KB, FETCH_SIZE reads half of the true bytes on gfx950.  Fabric-side bytes: Infinity-Cache hits are
    counted, so this is an upper bound on HBM traffic.  Returns (bytes_per_pass, source) or (None, {"error": ...})."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"error": "rocprofv3 not found"}
    child = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", "10", "--warmup", "2", "--repeats", "1",
             "--no-cpu-baseline", "--no-pmc", "--pmc-child"]
    if args.variant is not None:
        child += ["--variant", str(args.variant)]
    env = dict(os.environ, TMPDIR="/tmp", PCR_BENCH_NO_RCCL_PROBE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    by, passes, errs = {}, None, []
    for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        out = tempfile.mkdtemp(prefix="pcr_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "rocpd", "-d", out, "-o", "r", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True,
                               timeout=float(os.environ.get("PCR_BENCH_PMC_TIMEOUT", "240")))   # (a pass takes 10-60 s; on a timeout
            # the line says so in roofline.traffic_source.live_attempt and falls back to profiles/pmc_summary.json)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                errs.append(f"{counter}: rc {r.returncode} {r.stderr[-200:]}")
                continue
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name = ? group by name",
                               (counter,)).fetchall()
            for name, n, total in rows:
                if not any(h in name for h in HOT_KERNELS):
                    continue
                k = name.replace("void ", "").split("(")[0].strip()
                e = by.setdefault(k, {"launches": int(n), "fetch_bytes": 0, "write_bytes": 0})
                e["fetch_bytes" if counter == "FETCH_SIZE" else "write_bytes"] = int(round(total * scale / max(n, 1)))
        except Exception as exc:                              # noqa: BLE001 -- a probe must not take the bench line down
            errs.append(f"{counter}: {type(exc).__name__}: {exc}"[:200])
        finally:
            shutil.rmtree(out, ignore_errors=True)
    if not by or errs:
        return None, {"error": "; ".join(errs) or "no hot-path kernels in the PMC output"}
    passes = max([v["launches"] for k, v in by.items() if "finalize" in k] or [0])
    if passes <= 0:
        return None, {"error": "no folding kernel in the PMC output"}
    total = sum((v["fetch_bytes"] + v["write_bytes"]) * v["launches"] for v in by.values()) / passes
    return int(round(total)), {"live": True, "tool": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes)",
                               "passes": passes, "corrections": "KB -> bytes; FETCH_SIZE x2 (gfx950)",
                               "by_kernel_bytes_per_launch": by}


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: become ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`` (exec: same stdout,
    same exit code).  Rank 0 of that run prints the one JSON line."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")                     # what torchrun would set anyway; keeps the CPU quota for the ranks
    env["PCR_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def kernel_source_hash():
    """Identity of the kernels a PMC profile was collected on: sha1 of the DEVICE code of libpcr_hip.so (its
    .hip_fatbin section: unchanged by host-only edits), or of the HIP sources when the library is not built."""
    import struct
    lib = os.path.join(REPO, "point_cloud_registration_amd", "libpcr_hip.so")
    try:
        with open(lib, "rb") as f:
            data = f.read()
        assert data[:4] == b"\x7fELF" and data[4] == 2
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
        def sec(i):
            name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
            return name, off, size
        _, stroff, strsize = sec(shstrndx)
        for i in range(shnum):
            name, off, size = sec(i)
            end = data.index(b"\0", stroff + name)
            if data[stroff + name:end] == b".hip_fatbin":
                return "fatbin:" + hashlib.sha1(data[off:off + size]).hexdigest()[:16]
    except Exception:
        pass
    h = hashlib.sha1()
    d = os.path.join(REPO, "point_cloud_registration_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return "src:" + h.hexdigest()[:16]


def make_cloud(n, seed, config=""):
    from point_cloud_registration_amd.synthetic import street, street_tiled, lidar_sweep
    if "lidar" in config:
        return lidar_sweep(n, seed=seed)
    return street(n, seed=seed) if n <= 2_000_000 else street_tiled(n, seed=seed)


def make_scan(config, target, n_scan, family=None, seed=2):
    """The scan of a config and the pose align() should recover.  Families: "copy" (SURVEY.md 8d / the reference
    harness: a noisy copy of target points), "resampled" (an independent sample of the same surfaces, moved by
    T_true^-1, same noise), "crop" (the same, over a region of which only 30 % lies inside the map)."""
    from point_cloud_registration_amd.synthetic import (harness_scan, perturbed_scan, street, street_tiled, make_T,
                                                        T_TRUE_SO3, T_TRUE_T)
    family = family or SCAN_FAMILY.get(config, "copy")
    n_target = target.shape[0]
    if "harness" in config:
        T_true = np.eye(4)
        T_true[2, 3] = -0.3                                   # align(scan, I) undoes the +0.3 m shift
        return harness_scan(target, n_scan, seed=seed - 1), T_true
    if family == "copy":
        return perturbed_scan(target, n_scan if n_scan < n_target else None, seed=seed)
    if family == "crop":
        # the map's street and the next block (its own walls at x = 60 and 180), seen from a sensor that covers
        # x in [24, 144]: 36 m of the map (with its x = 60 wall, which pins the solution along the street) + 84 m beyond
        two = np.concatenate([street(2 * n_scan, seed=1000 + seed), street(2 * n_scan, seed=2000 + seed, center=(120.0, 0.0))])
        two = two[(two[:, 0] > 24.0) & (two[:, 0] < 144.0)]
        world = two[np.random.default_rng(seed).permutation(two.shape[0])[:n_scan]]
    elif n_target <= 2_000_000:
        world = street(n_scan, seed=1000 + seed)
    else:
        world = street_tiled(n_scan, seed=1000 + seed, per_tile=max(n_scan // max(round(n_target / 1_000_000), 1), 1))
    T_true = make_T(T_TRUE_SO3, T_TRUE_T)
    Ri = T_true[:3, :3].T
    rng = np.random.default_rng(seed)
    out = np.empty_like(world)
    for lo in range(0, world.shape[0], 4_000_000):            # chunked: the 12.5 M-point case in float64
        w = world[lo:lo + 4_000_000].astype(np.float64)
        out[lo:lo + 4_000_000] = ((w - T_true[:3, 3]) @ Ri.T + rng.normal(0.0, 0.005, w.shape)).astype(np.float32)
    return out, T_true


def main():
    args = parse()
    if args.gpus > 1 and not args.single_process and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(args)                                      # does not return
    # stdout carries exactly ONE JSON line: route everything else that libraries print there (RCCL's
    # start-up banner, for one) to stderr by swapping the file descriptor until the final print
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch                                   # first: one HIP runtime / one RCCL per process
    from point_cloud_registration_amd import _capi
    from point_cloud_registration_amd import distributed as pdist
    from point_cloud_registration_amd.synthetic import harness_scan, perturbed_scan

    sp = bool(args.single_process)                             # one process, a pcr_group over args.gpus devices
    world = args.gpus if sp else int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0 if sp else int(os.environ.get("RANK", "0"))
    local = 0 if sp else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    dev = local % torch.cuda.device_count()
    torch.cuda.set_device(dev)

    kind_name, n_target, n_scan, voxel_size, desc = CONFIGS[args.config]
    kind = {"icp": _capi.ICP, "plane": _capi.PLANE, "vplane": _capi.VPLANE, "ndt": _capi.NDT}[kind_name]
    max_dist = 2.0

    if sp:
        ids = os.environ.get("PCR_BENCH_GROUP_DEVICES")
        devs = [int(v) for v in ids.split(",")] if ids else [d % torch.cuda.device_count() for d in range(args.gpus)]
        if len(devs) != args.gpus:
            raise SystemExit("PCR_BENCH_GROUP_DEVICES must list --gpus device ids")
        ctx = _capi.get_group(devs)
        prof_ctx = ctx.member(0)
    else:
        ctx = _capi.get_context(dev)
        prof_ctx = ctx
    if args.variant is not None:
        for c in ([ctx.member(i) for i in range(world)] if sp else [ctx]):
            c.set_variant(args.variant)
    comm = None
    use_comm = (world > 1 and not sp) or bool(os.environ.get("PCR_BENCH_FORCE_COMM"))     # the latter: 1-rank self-test
    if use_comm:
        pdist.init_from_env(args.backend)
        comm = pdist.Communicator(ctx, in_library=True)      # RCCL inside libpcr_hip.so; every rank agrees on a fallback
    red_dev = "cuda" if args.backend == "nccl" else "cpu"    # where the bench's own max-over-ranks tensor lives

    # ---- workload (synthetic; same target on every rank, own scan shard per rank) -------------
    t_setup = time.perf_counter()
    data_tag = "synthetic"
    pcd = os.environ.get("PCR_B01_PCD", os.path.join(REPO, "data", "B-01.pcd"))
    if "b01" in args.config and os.path.exists(pcd):          # the real cloud, if it ever is on the box
        from point_cloud_registration_amd.io import load_pcd
        target = np.ascontiguousarray(load_pcd(pcd)["xyz"], dtype=np.float32)
        n_target = target.shape[0]
        if n_scan > n_target or args.config in ("plane_b01", "icp_b01"):
            n_scan = n_target
        data_tag = "B-01.pcd"
    else:
        target = make_cloud(n_target, seed=0, config=args.config)
    strong = args.scaling == "strong"
    sseed = 0 if strong else rank                              # strong: every rank builds the SAME scan, keeps a shard
    scan, T_true = make_scan(args.config, target, n_scan, seed=2 + sseed)
    n_scan_job = scan.shape[0] if strong else scan.shape[0] * world
    if sp and not strong and world > 1:
        # weak scaling in one process: the job's scan = what the N ranks of the SPMD run would hold, one after the other
        # (equal sizes, so pcr_group_scan_create's contiguous shards are exactly those scans)
        scan = np.concatenate([scan] + [make_scan(args.config, target, n_scan, seed=2 + r)[0] for r in range(1, world)])
    if strong and not sp:
        scan = np.ascontiguousarray(pdist.shard_scan(scan, rank, world))
    if kind_name in ("icp", "plane"):
        tgt = _capi.Target.points(ctx, target)
        if kind_name == "plane" and "lidar" in args.config:
            # PlaneICP.set_target(target, tree, normals) (plane_icp.py:25-27; what the reference's own benchmark does,
            # speed_test_comparison.py:25-32): k = 15 neighbours of a sweep's ring LINES are collinear, their PCA normal is
            # arbitrary and PlaneICP -- the reference's included -- converges to a wrong pose on them
            from point_cloud_registration_amd.synthetic import lidar_normals
            lidar_n = lidar_normals(target)
            tgt.set_normals(lidar_n)
        elif kind_name == "plane":
            # reference default k=15 (plane_icp.py:14).  The reference's float32 single-pass covariance
            # (estimate_normals.py:56-72) is kept where it works (|p| <= 60 m); at the 100 M cloud's
            # |p| ~ 600 m it loses every digit, so that config uses the centred float64 form
            tgt.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
    else:
        tgt = _capi.Target.voxels(ctx, target, voxel_size, 10)
    sc = _capi.Scan(ctx, scan)
    info = tgt.index_info()
    # a real Gauss-Newton trajectory to walk along (same on every rank: the sums are all-reduced)
    T_fin, iters, trace = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, max_dist, want_trace=True)
    traj = [trace[i, :16].reshape(4, 4).copy() for i in range(iters)]
    pose_err = float(np.linalg.norm(T_fin[:3, 3] - T_true[:3, 3]))
    t_setup = time.perf_counter() - t_setup

    # ---- the COLD numbers: what the reference's own protocol sees -- set_target on a fresh object, then
    # ONE align (benchmark/speed_test_comparison.py:14-55).  The timed steps below run on a target that has served dozens of
    # passes: its second list set / filter index exist (built lazily after 12 / 8 passes); a fresh target's first align never
    # sees them.  Kernels are loaded by now, the scan is device-resident: this is the library's time, not the process start-up.
    first_align_ms = set_target_ms = first_align_iters = None
    if n_target <= 20_000_000:
        ctx.synchronize()
        t0 = time.perf_counter()
        if kind_name in ("icp", "plane"):
            tgt2 = _capi.Target.points(ctx, target)
            if kind_name == "plane" and "lidar" in args.config:
                tgt2.set_normals(lidar_n)
            elif kind_name == "plane":
                tgt2.estimate_normals(15, compat=n_target <= 2_000_000, want=False)
        else:
            tgt2 = _capi.Target.voxels(ctx, target, voxel_size, 10)
        ctx.synchronize()
        set_target_ms = (time.perf_counter() - t0) * 1e3
        sc2 = _capi.Scan(ctx, scan)
        ctx.synchronize()
        t0 = time.perf_counter()
        _, first_align_iters = _capi.align(tgt2, sc2, kind, np.eye(4), 30, 1e-3, max_dist)
        first_align_ms = (time.perf_counter() - t0) * 1e3
        sc2.close(); tgt2.close()

    host_reduce = comm is not None and not comm.in_library       # fallback transport (see distributed.py)

    def step(k):
        o = _capi.linearize(tgt, sc, kind, traj[k % len(traj)], max_dist)
        return comm.allreduce(o) if host_reduce else o

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_comm:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    # With torch.cuda initialised the interpreter tracks ~1e6 objects and a generation-2 garbage
    # collection costs 35-50 ms -- a hundred passes' worth -- whenever the allocation counter trips
    # (measured: tools/latency_probe.py).  Collect now, keep the collector out of the timed region.
    import gc
    gc.collect()
    gc.disable()

    def timed_block():
        """EXACTLY args.steps passes, barrier + synchronize on both sides, max over ranks."""
        sync_all()
        t0 = time.perf_counter()
        o = None
        for k in range(args.steps):
            o = step(k)
        sync_all()
        dt = time.perf_counter() - t0
        if use_comm:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, o

    prof_ctx.profile_enable(True, period=args.event_period)
    prof_ctx.profile_reset()
    blocks = []
    out = None
    for r in range(max(args.repeats, 1)):
        dt, out = timed_block()
        blocks.append(dt)
    prof = prof_ctx.profile_read()
    prof_ctx.profile_enable(False)
    t_noev, _ = timed_block()                                  # the same block with the events off
    gc.enable()
    elapsed = float(np.median(blocks))
    step_t = [b / args.steps for b in blocks]

    per_rank = None
    if use_comm:                                                # every rank's own kernel / all-reduce times
        mine = {k: round(v[1] / v[0], 5) for k, v in prof.items() if v[0]}
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)
    if rank == 0:
        units = n_scan_job * args.steps                     # correspondences searched, whole job
        value = units / elapsed / 1e6
        kern = {k: {"launches": v[0], "avg_ms": v[1] / v[0]} for k, v in prof.items() if v[0]}
        # dominant kernel(s) of one pass: everything that touches the scan / target
        hot_ms = sum(kern[k]["avg_ms"] for k in ("linearize", "nn", "reduce") if k in kern)
        shard_n = sc.n // world if sp else sc.n             # points one launch of the profiled context processes
        alg_bytes = B_ALG[kind_name] * shard_n              # per launch (one pass over this rank's / member's shard)
        achieved = alg_bytes / (hot_ms * 1e-3) / 1e9
        # HBM traffic per pass from the PMC passes of the SAME command (profiles/pmc_summary.json, collected
        # as MI355X_MICROARCH.md prescribes: separate --pmc runs, FETCH_SIZE x2); only trusted when that
        # profile was taken on these very kernels (hash of the HIP sources), otherwise null
        traffic, traffic_src = None, None
        pmc_file = os.path.join(REPO, "profiles", "pmc_summary.json")
        if world == 1 and not args.no_pmc and not args.pmc_child and not os.environ.get("PCR_BENCH_NO_PMC"):
            traffic, traffic_src = live_traffic(args)
        if traffic is None and os.path.exists(pmc_file):
            live_err = traffic_src
            try:
                pm = json.load(open(pmc_file))
                entry = pm.get(args.config, {})
                if pm.get("kernel_source_hash") == kernel_source_hash():
                    traffic = entry.get("hbm_bytes_per_pass")
                    traffic_src = {"file": "profiles/pmc_summary.json", "collected_at_commit": pm.get("commit"),
                                   "kernel_source_hash": pm.get("kernel_source_hash"), "raw": entry.get("source")}
                else:
                    traffic_src = {"file": "profiles/pmc_summary.json", "stale": True,
                                   "note": "PMC profile predates the current kernels; traffic withheld"}
            except Exception:
                traffic = None
            if live_err and isinstance(traffic_src, dict):
                traffic_src["live_attempt"] = live_err
        line = {
            "metric": "M-correspondences/sec, %s calc_H_g_e2 on the B-01 stand-in (1.06 M pts)" % kind_name
                      if "b01" in args.config else "M-correspondences/sec, %s calc_H_g_e2" % kind_name,
            "value": round(value, 3), "unit": "Mcorr/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "iterations_per_sec": round(args.steps / elapsed, 2),
            "ms_per_step_min": round(min(step_t) * 1e3, 4), "ms_per_step_max": round(max(step_t) * 1e3, 4),
            "repeat_ms_per_step": [round(t * 1e3, 4) for t in step_t], "repeats": len(blocks),
            "ms_per_step_events_off": round(t_noev / args.steps * 1e3, 4), "event_period": args.event_period,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "accumulate_dtype": "f64", "data": data_tag,
            "config": {"workload": args.config, "description": desc, "kind": kind_name,
                       "target_points": int(n_target), "scan_points_per_gpu": int(sc.n // world if sp else sc.n),
                       "max_dist": max_dist, "voxel_size": voxel_size,
                       "parallelism": f"single-process group x{world}" if sp else f"scan-shard x{world}",
                       "backend": (args.backend if use_comm else None),
                       "allreduce_transport": ("p2p-in-process" if sp and world > 1 else None) if comm is None else (
                                               (("p2p-ipc-in-stream" if comm.transport == "p2p" else "rccl-in-stream")
                                                if comm.in_library else f"host-{args.backend}")),
                       "group_devices": (list(ctx.devices) if sp else None),
                       "devices_visible": torch.cuda.device_count(),
                       "scan_points_job": int(n_scan_job),
                       "nn_index": {"cell": info["cell"], "dims": info["dims"], "occupied_cells": info["occupied"],
                                    "halo_m": info["halo"], "halo_records": info["halo_records"],
                                    "cell_population_max": info.get("pop_max"), "cell_population_p99": info.get("pop_p99"),
                                    "heavy_cells_index": info.get("heavy")},
                       "gauss_newton_iters_to_converge": iters, "pose_error_m": round(pose_err, 6),
                       "first_align_ms": None if first_align_ms is None else round(first_align_ms, 3),
                       "first_align_iterations": first_align_iters,
                       "set_target_ms": None if set_target_ms is None else round(set_target_ms, 3),
                       "first_align_note": "fresh target (no second list set / filter index yet), one align of the device-resident "
                                           "scan: the reference's protocol; `value` is the steady state of a target that has served "
                                           "dozens of passes",
                       "correspondences_last_step": int(out[28]), "setup_s": round(t_setup, 2)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "+".join(k for k in ("linearize", "nn", "reduce") if k in kern),
                         "kernel_ms": round(hot_ms, 5), "algorithmic_bytes_per_launch": int(alg_bytes),
                         "bytes_per_point": B_ALG[kind_name]},
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 5)} for k, v in kern.items()},
        }
        if "reduce" in kern:
            # SURVEY.md section 8d asks for the streaming kernel on its own: B_alg + the 4-byte index per point
            k2 = (B_ALG[kind_name] + 4) * shard_n / (kern["reduce"]["avg_ms"] * 1e-3) / 1e9
            line["roofline_reduce_kernel"] = {"bound": "hbm", "achieved": round(k2, 3), "peak": HBM_PEAK_GBS,
                                              "unit": "GB/s", "frac": round(k2 / HBM_PEAK_GBS, 6),
                                              "bytes_per_point": B_ALG[kind_name] + 4,
                                              "kernel_ms": round(kern["reduce"]["avg_ms"], 5)}
        if per_rank is not None:
            line["per_rank_kernel_ms"] = per_rank
        if world == 1 and not use_comm and not os.environ.get("PCR_BENCH_NO_RCCL_PROBE") and not args.pmc_child:
            one = rccl_one_rank_probe(ctx, step, args.steps, sync_all)
            line["rccl_1rank"] = one.get("rccl", one)                 # in-stream exchange with ONE rank attached: everything but the hops
            line["p2p_1rank"] = one.get("p2p", one)                   # ... and the same for the peer-to-peer transport (PCR_COMM=p2p)
        if world == 1 and not args.pmc_child:
            line["seam"] = seam_timings(kind_name, target, scan, tgt, sc, kind, traj, max_dist, voxel_size, n_target)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kind_name, target, scan, tgt, traj, max_dist, voxel_size,
                                                args.cpu_passes)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if comm is not None:
        torch.distributed.barrier()
        comm.close()
        torch.distributed.destroy_process_group()


def rccl_one_rank_probe(ctx, step, steps, sync_all):
    """What the multi-GPU path adds to a pass, measured with a ONE-rank communicator on this GPU: the in-stream
    ncclAllReduce(29 doubles) + the hand-off kernel (k_publish) sit between the fold and the host exactly as they
    do at N = 8; only the xGMI hops are missing.  A baseline to judge the first real SCALE run against; never part
    of ``value``.  Failures are reported, not raised."""
    try:
        import socket
        import torch
        from point_cloud_registration_amd import distributed as pdist
        if not torch.distributed.is_initialized():
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            pdist.init_from_env("nccl")
        res = {}
        try:
            for transport in ("rccl", "p2p"):
                comm = pdist.Communicator(ctx, in_library=True, transport=transport)
                try:
                    if not comm.in_library:
                        res[transport] = {"error": f"{transport} communicator not available inside libpcr_hip.so"}
                        continue
                    for k in range(5):
                        step(k)
                    ctx.profile_enable(True, period=1); ctx.profile_reset()
                    sync_all()
                    t0 = time.perf_counter()
                    for k in range(steps):
                        step(k)
                    sync_all()
                    dt = time.perf_counter() - t0
                    prof = ctx.profile_read(); ctx.profile_enable(False)
                    sync_all()
                    t0 = time.perf_counter()
                    for k in range(steps):
                        step(k)
                    sync_all()
                    dt_off = time.perf_counter() - t0
                    n, ms = prof["allreduce"]
                    res[transport] = {"ms_per_step_events_off": round(dt_off / steps * 1e3, 4), "ms_per_step_profiled": round(dt / steps * 1e3, 4),
                                      "allreduce29_plus_publish_avg_ms": round(ms / max(n, 1), 5), "launches": n}
                finally:
                    comm.close()
            return res
        finally:
            torch.distributed.destroy_process_group()
    except Exception as e:                                   # noqa: BLE001 -- a probe must not take the bench line down
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def seam_timings(kind_name, target, scan, tgt, sc, kind, traj, max_dist, voxel_size, n_target):
    """Beside the C-ABI headline: the same pass through the reference-shaped CLASS seam (S1, registration.py:55-68:
    ``calc_H_g_e2(cur_T, source)`` with the scan as a host array -- content hash per call -- and with an uploaded
    handle), and whole ``align()`` calls (``set_target`` excluded): behind the C ABI on the resident scan, and
    through the class from the host array (upload + Morton sort included).  Not part of ``value``."""
    import gc
    import point_cloud_registration_amd as pcr
    from point_cloud_registration_amd import _capi
    out = {}
    reps = 3
    ts = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        T, it = _capi.align(tgt, sc, kind, np.eye(4), 30, 1e-3, max_dist)
        ts.append(time.perf_counter() - t0)
    out["align_ms"] = round(float(np.median(ts[1:])) * 1e3, 4)
    out["align_iterations"] = int(it)
    if n_target > 20_000_000:
        out["note"] = "class seam not timed at this size (it would rebuild the 1e8-point target)"
        return out
    cls = {"icp": lambda: pcr.ICP(max_dist=max_dist), "plane": lambda: pcr.PlaneICP(max_dist=max_dist, k=15),
           "vplane": lambda: pcr.VPlaneICP(voxel_size=voxel_size, max_dist=max_dist),
           "ndt": lambda: pcr.NDT(voxel_size=voxel_size, max_dist=max_dist)}[kind_name]()
    t0 = time.perf_counter()
    cls.set_target(target)
    out["set_target_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    n = max(3 * len(traj), 15)
    gc.collect(); gc.disable()
    try:
        for form in ("array", "handle"):
            src = scan if form == "array" else cls.upload(scan)
            for k in range(len(traj)):
                cls.calc_H_g_e2(traj[k], src)
            t0 = time.perf_counter()
            for k in range(n):
                cls.calc_H_g_e2(traj[k % len(traj)], src)
            out[f"calc_H_g_e2_{form}_ms_per_call"] = round((time.perf_counter() - t0) / n * 1e3, 4)
        t0 = time.perf_counter()
        for _ in range(20):
            _capi.hash64(scan)
        out["content_hash_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        ts = []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            cls.align(scan)
            ts.append(time.perf_counter() - t0)
        out["class_align_from_host_array_ms"] = round(float(np.median(ts[1:])) * 1e3, 4)
    finally:
        gc.enable()
    return out


def cpu_baseline(kind_name, target, scan, gpu_target, traj, max_dist, voxel_size, passes):
    """The CPU oracle (a port of the reference's NumPy path to C + OpenMP) on this box's host cores:
    the same calc_H_g_e2 on a bounded sample of the same workload."""
    from oracle import oracle as orc
    cpus = effective_cpus()
    team = int(os.environ.get("PCR_CPU_BASELINE_THREADS", cpus))
    orc.set_threads(team)                             # the container's quota, not the 256 visible CPUs (oversubscribed + throttled)
    cap = 1_060_000                                   # bound the CPU work to ~10-30 s
    src = scan[:cap]
    tgt_pts = target
    t0 = time.perf_counter()
    if kind_name in ("icp", "plane"):
        normals = gpu_target.get_normals() if kind_name == "plane" else None   # same normals as the GPU run
        ot = orc.TargetPoints(tgt_pts, normals=normals, cell=0.5)
    else:
        ot = orc.TargetVoxels(tgt_pts, voxel_size)
    t_build = time.perf_counter() - t0
    kind = {"icp": orc.ICP, "plane": orc.PLANE, "vplane": orc.VPLANE, "ndt": orc.NDT}[kind_name]
    orc.calc_H_g_e2(kind, ot, traj[0], src[:10000], max_dist)            # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < passes or (time.perf_counter() - t0 < 10.0 and done < 200):     # ~10 s of CPU work
        orc.calc_H_g_e2(kind, ot, traj[done % len(traj)], src, max_dist)
        done += 1
    passes = done
    dt = time.perf_counter() - t0
    return {"value": round(src.shape[0] * passes / dt / 1e6, 4), "unit": "Mcorr/s",
            "cores": orc.max_threads(), "effective_cpus": cpus, "host_cpus_visible": os.cpu_count(), "kind": "port",
            "sample": f"{passes} passes of the same calc_H_g_e2 over {src.shape[0]} scan points "
                      f"(oracle/pcr_oracle.c, OpenMP team of {orc.max_threads()} = the cgroup CPU quota, exact grid NN); "
                      f"index build {t_build:.2f} s excluded"}


if __name__ == "__main__":
    main()
