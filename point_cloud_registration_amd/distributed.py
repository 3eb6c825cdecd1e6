"""Multi-GPU: one process per GPU, the scan sharded across ranks, one 29-double all-reduce of the
normal equations per iteration.  The reference is single-process.

Two transports for the same 232-byte exchange:
* ``in_library=True`` (GPU runs): RCCL inside libpcr_hip.so, on the kernel's own HIP stream,
  between the finalize kernel and the device-to-host copy -- no Python, no torch in the loop.
  The RCCL unique id is distributed once through ``torch.distributed`` (any backend).
* ``in_library=False``: ``torch.distributed.all_reduce`` on a host tensor (``gloo``); used by the
  CPU tests of the N>1 path and available as a fallback.
"""

import os

import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of an n-point scan for ``rank`` of ``world``."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_scan(source, rank, world):
    lo, hi = shard_bounds(source.shape[0], rank, world)
    return source[lo:hi]


class Communicator:
    """``transport`` (in-library exchanges only): "rccl" (default) or "p2p" -- the peer-to-peer exchange of csrc/comm.hip
    (IPC-mapped slots, no collective library in the loop; ``PCR_COMM=p2p`` selects it where no argument is given)."""

    def __init__(self, ctx=None, in_library=True, group=None, transport=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (torchrun / init_process_group)")
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.ctx = ctx
        self.in_library = bool(in_library) and ctx is not None
        self.transport = (transport or os.environ.get("PCR_COMM") or "rccl").lower() if self.in_library else "host"
        if self.in_library and self.transport == "p2p":
            ok, handle = 1, None
            try:
                handle = ctx.comm_p2p_export()
            except Exception as exc:
                ok = 0
                print(f"[pcr] rank {self.rank}: peer-to-peer export failed ({exc})", flush=True)
            #  coarse-grained slots -- the fallback of pcr_comm_p2p_export -- are coherent only between ranks that
            # share ONE device; ranks on different devices that could not get fine-grained slots take the host transport
            import socket
            me = (handle, bool(ok and ctx.comm_p2p_finegrained()), (socket.gethostname(), int(ctx.device)))
            infos = [None] * self.world
            dist.all_gather_object(infos, me, group=group)
            handles = [i[0] for i in infos]
            if not all(i[1] for i in infos) and len({i[2] for i in infos}) > 1:
                ok = 0
                if self.rank == 0:
                    print("[pcr] peer-to-peer slots are not fine-grained on every rank and the ranks span several devices: "
                          "host all-reduce instead", flush=True)
            if ok and all(h is not None for h in handles):
                try:
                    ctx.comm_p2p_attach(handles, self.rank)
                except Exception as exc:
                    ok = 0
                    print(f"[pcr] rank {self.rank}: peer-to-peer attach failed ({exc})", flush=True)
            else:
                ok = 0
            flags = [None] * self.world
            dist.all_gather_object(flags, ok, group=group)
            if min(flags) == 0:                              # every rank takes the same path
                ctx.comm_destroy()
                self.in_library = False
                self.transport = "host"
            dist.barrier(group=group)                        # nobody exchanges before everybody has mapped everybody
        elif self.in_library:
            from . import _capi
            ok = 1
            try:
                uid = [_capi.comm_unique_id() if self.rank == 0 else None]
            except Exception as exc:                      # RCCL not loadable on rank 0
                uid, ok = [None], 0
                print(f"[pcr] RCCL unavailable ({exc}); using torch.distributed for the all-reduce", flush=True)
            dist.broadcast_object_list(uid, src=0, group=group)
            if uid[0] is None:
                ok = 0
            else:
                try:
                    ctx.comm_init(uid[0], self.world, self.rank)
                except Exception as exc:
                    ok = 0
                    print(f"[pcr] rank {self.rank}: in-library RCCL init failed ({exc})", flush=True)
            # every rank must take the same path: agree on the minimum
            flags = [None] * self.world
            dist.all_gather_object(flags, ok, group=group)
            if min(flags) == 0:
                if ok:
                    ctx.comm_destroy()
                self.in_library = False
                self.transport = "host"

    def allreduce(self, out29):
        """Host-side sum of the 29 doubles over all ranks (gloo path)."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(out29, dtype=np.float64).copy())
        # (the fallback of a failed in-library init on a GPU run: torch's nccl backend reduces device tensors only)
        on_device = self._dist.get_backend(self.group) == "nccl" and torch.cuda.is_available()
        if on_device:
            t = t.cuda()
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy() if on_device else t.numpy()

    def barrier(self):
        self._dist.barrier(group=self.group)

    def close(self):
        if self.in_library and self.ctx is not None:
            self.ctx.comm_destroy()


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl":
        dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev)
        try:
            kw["device_id"] = torch.device("cuda", dev)
        except Exception:
            pass
    dist.init_process_group(backend=backend, **kw)
