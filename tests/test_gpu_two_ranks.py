"""Two processes sharing the one GPU of the test box: the sharded GPU path end to end.

RCCL refuses two ranks on the same device, so ``Communicator(ctx, in_library=True)`` must detect the
failure on every rank, agree on it, and fall back to the host all-reduce (gloo) -- exactly the
fallback a broken RCCL set-up would take on a real 8-GPU node.  Each rank runs the HIP kernels on its
scan shard; the result must equal the single-process run on the whole scan."""

import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PCR_DEVICE="0")
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    import point_cloud_registration_amd as pcr
    from point_cloud_registration_amd import _capi, distributed as pdist
    pdist.init_from_env("gloo")
    g2 = load_golden("g2_mini_street.npz")
    ctx = _capi.get_context(0)
    comm = pdist.Communicator(ctx, in_library=True)          # RCCL init fails (same GPU twice) -> agreed fallback
    icp = pcr.PlaneICP(max_dist=float(g2["max_dist"]), k=int(g2["k"]), comm=comm)
    icp.set_target(g2["target"], None, None)
    icp.set_target(g2["target"], icp.kdtree, g2["plane_normals"])
    shard = pdist.shard_scan(g2["source"], rank, world)
    T = icp.align(shard, np.eye(4))
    H, g, e2 = icp.calc_H_g_e2(g2["T"], shard)
    q.put((rank, comm.in_library, T, H, icp.last_iterations, icp.last_correspondences))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_plane_icp(g2):
    import multiprocessing as mp            # (not torch.multiprocessing: keep torch out of the parent)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, lib_a, Ta, Ha, ita, ca), (_, lib_b, Tb, Hb, itb, cb) = res
    assert lib_a == lib_b                                     # both ranks took the same transport
    print("transport:", "RCCL inside libpcr_hip.so" if lib_a else "host all-reduce (gloo) fallback")
    assert np.array_equal(Ta, Tb) and np.array_equal(Ha, Hb) and ita == itb and ca == cb
    assert ita == g2["align_plane_T"].shape[0]
    final = g2["align_plane_final"]
    assert np.max(np.abs(Ta[:3, 3] - final[:3, 3])) < 1e-4
    assert np.max(np.abs(Ha - g2["T_plane_H"])) < 1e-5 * np.max(np.abs(g2["T_plane_H"]))


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["rccl_then_torch", "torch_then_rccl"])
def test_process_exits_cleanly_with_rccl_and_torch(order):
    """Interpreter exit after RCCL was used through libpcr_hip.so AND torch was imported, in either
    order: must be exit code 0 (regression: librocm_smi64.so in the global symbol scope clashed with
    torch's libamd_smi.so -> 'double free or corruption' in a static destructor)."""
    import subprocess
    import sys
    body = ("from point_cloud_registration_amd import _capi\n"
            "ctx = _capi.get_context(0)\n"
            "ctx.comm_init(_capi.comm_unique_id(), 1, 0); ctx.comm_destroy()\n")
    torch_part = "import torch\ntorch.zeros(4).cuda(); torch.cuda.synchronize()\n"
    code = f"import sys; sys.path.insert(0, {REPO!r})\n" + (body + torch_part if order == "rccl_then_torch" else torch_part + body)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
