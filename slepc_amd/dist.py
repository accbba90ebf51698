"""One process per GPU: wiring of a libksgpu context to the other ranks.

``torch.distributed`` is used only as the rendezvous / plumbing layer (it reads
RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT as set by ``torch.distributed.run``):

* ``transport="rccl"``  -- rank 0 creates an RCCL unique id, the id is broadcast
  over the (gloo) process group, and every rank joins the RCCL communicator
  inside libksgpu; all data-path collectives (dot-product all-reduce, halo
  exchange) then run natively on RCCL over xGMI.
* ``transport="gloo"``  -- the collectives are served by callbacks into
  ``torch.distributed`` on host buffers (used by the tests, where two ranks may
  share one GPU, which RCCL does not allow).
"""
import ctypes as C
import os
import sys

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend="gloo")
    return dist


def gloo_callbacks(dist):
    """(allreduce, exchange) working on raw host addresses through gloo."""
    import torch

    def _tensor(addr, count):
        buf = (C.c_double * count).from_address(addr)
        return torch.from_numpy(np.ctypeslib.as_array(buf))

    def allreduce(addr, count, op):
        t = _tensor(addr, count)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)

    def exchange(saddr, soff, raddr, roff):
        rank, size = dist.get_rank(), dist.get_world_size()
        reqs = []
        for p in range(size):
            ns, nr = soff[p + 1] - soff[p], roff[p + 1] - roff[p]
            if p == rank:
                if ns:
                    C.memmove(raddr + 8 * roff[p], saddr + 8 * soff[p], 8 * ns)
                continue
            if nr:
                reqs.append(dist.irecv(_tensor(raddr + 8 * roff[p], nr), src=p))
            if ns:
                reqs.append(dist.isend(_tensor(saddr + 8 * soff[p], ns), dst=p))
        for r in reqs:
            r.wait()

    return allreduce, exchange


def _check_allreduce(ctx, dist, size):
    """The library's own self-test (a sum and a ring exchange over the halo path), then one
    reduction through a vector kernel against the known answer."""
    ctx.comm_selftest()
    n = 1000
    x = ctx.vec_alloc(n)
    ctx.vec_set(x, 1.0, n)
    got = ctx.vec_dot(x, x, n)
    ctx.vec_free(x)
    if abs(got - float(n * size)) > 1e-9 * n * size:
        raise RuntimeError("communicator self-test: sum over %d ranks gave %r" % (size, got))


def init_comm(ctx, transport="rccl", fallback=True):
    """Joins `ctx` to the communicator of all ranks.  Returns (rank, nranks, transport used).

    With transport "rccl" every rank reports over gloo whether it joined the RCCL communicator
    and passed a one-reduction self-test; unless all did, and `fallback` is set, all of them
    leave it and the collectives are served by gloo on host buffers instead (slower, correct),
    so that a multi-GPU run is not lost to a communicator that will not come up."""
    rank, _, size = env_rank()
    if size == 1 and transport != "rccl-force":
        return 0, 1, "none"
    dist = init_process_group()
    if transport.startswith("rccl"):
        import torch
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
            # one node: RCCL's bootstrap sockets go over the loopback interface, which is there
            # whatever the host name resolves to; the data path (xGMI peer-to-peer) is not affected
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        ok = 1
        try:
            uid = [ctx.unique_id() if rank == 0 else None]
        except Exception as e:           # RCCL not loadable on rank 0
            uid, ok = [None], 0
            print("rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            ok = 0
        joined = False
        if ok:
            try:
                ctx.comm_init_rccl(rank, size, uid[0])
                joined = True
                _check_allreduce(ctx, dist, size)
            except Exception as e:
                ok = 0
                print("rank %d: RCCL communicator unusable: %s" % (rank, e), file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            return rank, size, "rccl"
        if joined:
            try:
                ctx.comm_finalize()
            except Exception as e:       # pragma: no cover
                print("rank %d: leaving the RCCL communicator failed: %s" % (rank, e), file=sys.stderr, flush=True)
        if not fallback:
            raise RuntimeError("RCCL communicator could not be set up on every rank")
        transport = "gloo"
    if transport == "gloo":
        ar, ex = gloo_callbacks(dist)
        ctx.comm_init_callbacks(rank, size, ar, ex, on_device=False)
        ctx.comm_selftest()
        return rank, size, "gloo"
    raise ValueError("unknown transport " + transport)
