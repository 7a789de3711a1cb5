"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" for the CPU tests).

The hot path shards embarrassingly by query: every rank
runs the full path on its own block of the query stream.  The only collective is
the one-off broadcast of the device-resident k-mer index (and, if wanted, the
packed reference store) from rank 0 at start-up; there are no per-step
collectives.  Timing: barrier + max over ranks.
"""
import ctypes as C
import os

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Initialises torch.distributed from the environment if WORLD_SIZE > 1."""
    rank, local_rank, world = env_world()
    if world == 1:
        return rank, local_rank, world, None
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world, dist


def shard_range(n, rank, world):
    """Contiguous block of [0, n) owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_array(arr, src, dist, device=None):
    """Broadcasts a numpy array (shape/dtype known on all ranks) through torch.distributed."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def reduce_max(value, dist, device=None):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, dist, device=None):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def merge_by_seqno(local, dist, dst=0, chunk=2048, device=None, stats=None):
    """--preserve-order across ranks (reference src/sina.cpp:529-538 orders trays by seqno before the
    writers): every rank hands in its [(seqno, payload), ...]; rank `dst` gets them all, ascending by
    seqno, the others get None -- and RECEIVE nothing: the results travel to `dst` only (a gather, in
    rounds of `chunk` items per rank so that no rank pickles its whole block at once; a 100 000-query run
    carries ~6 KB of alignment per query).  `dist` None: a single process, just sorted.  `stats`, if a
    dict, receives this rank's "sent_items" / "received_items"."""
    import heapq
    mine = sorted(local, key=lambda x: x[0])
    if dist is None:
        return mine
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None and str(dist.get_backend()) == "nccl":  # (RCCL moves device tensors only)
        import torch
        device = torch.device("cuda", torch.cuda.current_device())
    rounds = int(reduce_max((len(mine) + chunk - 1) // chunk, dist, device))
    parts = [[] for _ in range(world)] if rank == dst else None
    received = 0
    for r in range(rounds):
        piece = mine[r * chunk:(r + 1) * chunk]
        got = [None] * world if rank == dst else None
        dist.gather_object(piece, got, dst=dst)
        if rank == dst:
            for w in range(world):
                parts[w].extend(got[w])
                if w != dst:
                    received += len(got[w])
    if stats is not None:
        stats["sent_items"] = 0 if rank == dst else len(mine)
        stats["received_items"] = received
    if rank != dst:
        return None
    return list(heapq.merge(*parts, key=lambda x: x[0]))  # (every rank's part is sorted already)


class _DevView:
    """Exposes a raw device pointer through __cuda_array_interface__ so that torch can wrap
    it without copying (torch.as_tensor)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def broadcast_device_index(store, k, nofast, rank, dist, device):
    """Rank 0 has built the index on its GPU; every other rank allocates buffers of the same
    shape inside its sina_hip context and receives the bytes in place over RCCL (xGMI).
    `store` is a sina_amd.pipeline.Store whose references are already uploaded."""
    import torch
    from . import capi
    L = capi.load()
    ctx = store.ctx_handle()
    view = capi.StoreView()
    meta = np.zeros(6, np.int64)
    if rank == 0:
        store.build_index(k, nofast)
        if L.sina_hip_store_view_get(ctx, C.byref(view)) != 0:
            raise RuntimeError(L.sina_hip_last_error().decode())
        meta[:] = [view.n_refs, view.width, view.k, view.nofast, view.n_postings, view.total_bases]
    meta = broadcast_array(meta, 0, dist, device)
    if rank != 0:
        view.n_refs, view.width, view.k, view.nofast = int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])
        view.n_postings, view.total_bases = int(meta[4]), int(meta[5])
        if L.sina_hip_store_alloc_like(ctx, C.byref(view)) != 0:
            raise RuntimeError(L.sina_hip_last_error().decode())
    for ptr, nbytes in ((view.idx_offsets, view.idx_offsets_bytes), (view.idx_ids, view.idx_ids_bytes),
                        (view.ref_ab, view.ref_ab_bytes), (view.ref_off, view.ref_off_bytes)):
        if nbytes:
            t = torch.as_tensor(_DevView(ptr, nbytes), device=device)
            dist.broadcast(t, src=0)
    torch.cuda.synchronize(device)
    if rank != 0:
        store.index_ready(k, nofast)
    return int(meta[4])
