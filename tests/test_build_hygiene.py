"""Build-time guarantees that need no GPU: the cross-XCD hand-off of k_reduce_finalize as the compiler
actually emits it (pinned against a compiler bump), and the CPU oracle under ASan + UBSan."""

import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

CSRC = os.path.join(REPO, "point_cloud_registration_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_ticket_handoff_isa(tmp_path):
    """k_reduce_finalize: every block's 32 partial sums are stored write-through at agent scope (sc1) and
    the block's ticket atomic may only be issued once those stores have completed.  The protocol is
    'sc1 payload -> s_waitcnt vmcnt(0) -> ticket atomic -> sc1 (L1-bypassing) loads by the folding block'
    (MI355X guide, inter-workgroup visibility).  Check the emitted gfx950 ISA for exactly that order at
    both ticket levels, for every kind."""
    asm = tmp_path / "kernels.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    f"-I{REPO}/include", f"-I{CSRC}", "-S", "--cuda-device-only",
                    os.path.join(CSRC, "kernels.hip"), "-o", str(asm)], check=True, capture_output=True)
    text = asm.read_text()
    found = 0
    # (FIX = 1: the voxel kinds behind the float32 filter, with the pending-point prologue in front of the stream)
    for kind, fix in ((0, 0), (1, 0), (2, 0), (3, 0), (2, 1), (3, 1)):
        m = re.search(rf"^_Z17k_reduce_finalizeILi{kind}ELi{fix}EEv7LinArgs7FinArgs:(.*?)^\.Lfunc_end", text, re.S | re.M)
        assert m, f"k_reduce_finalize<{kind}, {fix}> not found"
        lines = [l.strip() for l in m.group(1).splitlines()]
        atomics = [i for i, l in enumerate(lines) if l.startswith("global_atomic_add")]
        assert len(atomics) == 2, (kind, atomics)                     # group ticket, leader ticket
        for a in atomics:
            # walking back from the ticket: a full vmcnt(0) drain must come before any sc1 payload store
            drained, stores_seen = False, 0
            for l in reversed(lines[:a]):
                if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                    drained = True
                    break
                if l.startswith("global_store_dwordx2") and "sc1" in l:
                    stores_seen += 1
                    break
            assert drained and stores_seen == 0, f"kind {kind}: ticket atomic not behind a drained payload store"
        # payload stores are write-through, the folding loads bypass L1
        assert any(l.startswith("global_store_dwordx2") and l.endswith("sc1") for l in lines)
        assert sum(1 for l in lines if l.startswith("global_load_dwordx2") and l.endswith("sc1")) >= 24
        found += 1
    assert found == 6


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_kernels_use_no_scratch(tmp_path):
    """Round 5 lesson: a template branch added to k_nn_filter moved 928 bytes of its state per lane to scratch and cost the voxel
    configs 1.5x, and a run-time switch between two tile hand-out schemes did the same to its block-local variant -- silently, every
    test green.  The compiler's own resource report is therefore part of the suite: the kernels a plain pass runs (point search,
    filter search, every reduce kernel, the certificate) use NO scratch; the rare variants (tracking / list passes over voxel
    targets, the fused small-scan kernels) stay below 128 bytes per lane."""
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
                        f"-I{REPO}/include", f"-I{CSRC}", "-Rpass-analysis=kernel-resource-usage", "-c",
                        os.path.join(CSRC, "kernels.hip"), "-o", str(tmp_path / "k.o")], capture_output=True, text=True, check=True)
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            usage[name] = int(m.group(1))
    assert len(usage) > 60, "the resource report was not parsed"
    hot = [n for n in usage if re.match(r"_Z9k_nn_scanILi0E", n) or "k_nn_filter" in n or "k_reduce_finalize" in n or "k_certify" in n
           or "k_scan_reduce" in n]                 # (k_scan_reduce: opt-in, round 6 -- at 4 waves / SIMD it must not spill either)
    assert len(hot) >= 30
    assert {n: usage[n] for n in hot if usage[n] != 0} == {}
    assert {n: b for n, b in usage.items() if b > 128} == {}


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_oracle_under_sanitizers(tmp_path):
    """the C oracle once under -fsanitize=address,undefined (grid NN, all four
    linearisations, voxel build, k-NN normals, the Gauss-Newton loop) on a small seeded case."""
    so = tmp_path / "libpcr_oracle_san.so"
    subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-omit-frame-pointer",
                    "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    os.path.join(REPO, "oracle", "pcr_oracle.c"), "-o", str(so), "-lm"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as orc
from point_cloud_registration_amd.synthetic import street, perturbed_scan
tgt = (street(6000, seed=3) * np.float32(0.1)).astype(np.float32); scan, T = perturbed_scan(tgt, 1500, seed=4, noise=0.0005)
dk, ik = orc.knn_brute(tgt, tgt, 10)
nrm = orc.normals_from_knn(tgt, ik, compat=False)
pt = orc.TargetPoints(tgt, normals=nrm); vx = orc.TargetVoxels(tgt, 1.0)
assert vx.mean.shape[0] > 20
for kind, t in ((orc.ICP, pt), (orc.PLANE, pt), (orc.VPLANE, vx), (orc.NDT, vx)):
    H, g, e2 = orc.calc_H_g_e2(kind, t, T, scan, 2.0)
    assert np.isfinite(H).all() and np.isfinite(g).all()
    orc.align(kind, t, scan, np.eye(4), 5, 1e-3, 2.0)
d, i = orc.nn_brute(tgt, orc.transform(T, scan)); dk, ik = orc.knn_brute(tgt, tgt[:200], 7)
orc.normals_from_knn(tgt, ik, compat=True); orc.normals_from_knn(tgt, ik, compat=False)
orc.voxel_build(tgt, 0.5, 10)
print("sanitized-ok")
""" % REPO
    env = dict(os.environ, PCR_ORACLE_LIB=str(so), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
