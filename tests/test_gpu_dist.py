"""The multi-rank start-up path on ONE GPU (the 8-GPU run itself is the driver's): the device-resident
reference store + k-mer index handed from rank 0 to the other ranks in place -- buffers allocated by
sina_hip_store_alloc_like, wrapped as torch tensors over the C ABI's raw pointers, filled by
torch.distributed.broadcast -- and what those ranks then compute from it.

  * world size 1 over RCCL (backend "nccl"): process group, device views, in-place broadcast, as
    bench.py does under SINA_BENCH_FORCE_DIST;
  * world size 2 on the same GPU over gloo (RCCL refuses two ranks on one device): rank 1 never uploads
    the references -- it only receives -- runs its block of the queries, and the merged result equals a
    single process's.
Every rank runs in a child process (a process group per process)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    from sina_amd import synth
    refs = synth.make_refs(600, length=300, width=2400, seed=811)
    qs = synth.make_queries(refs, 24, seed=812)
    return refs, qs


FF = {"fs-min-len": 100, "fs-full-len": 250}


def _run_block(store, qs, lo, hi):
    from sina_amd import pipeline
    pl = pipeline.Pipeline(store, famfinder=FF)
    off = (qs.off[lo:hi + 1] - qs.off[lo]).astype(np.uint64)
    pl.run(qs.mask[qs.off[lo]:qs.off[hi]], off, batch=8, inflight=2)
    out = []
    for i in range(hi - lo):
        r = pl.result(i)
        out.append((lo + i, (r["status"], r["family"], r["packed"].tobytes(), r["head"], r["tail"], r["qual"], r["log"])))
    pl.close()
    return out


def _rank(rank, world, port, backend, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as tdist
    from sina_amd import dist as sdist
    from sina_amd import pipeline
    tdist.init_process_group(backend=backend, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    refs, qs = _inputs()
    store = pipeline.Store(":mem:dist-rank%d" % rank, refs, device=0, upload=(rank == 0))
    n_post = sdist.broadcast_device_index(store, 10, False, rank, tdist, device)
    assert n_post > 0
    lo, hi = sdist.shard_range(qs.n, rank, world)
    merged = sdist.merge_by_seqno(_run_block(store, qs, lo, hi), tdist)
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), np.array(merged, dtype=object), allow_pickle=True)
    tdist.barrier()
    store.close()
    tdist.destroy_process_group()


def _single_process_results():
    from sina_amd import pipeline
    refs, qs = _inputs()
    st = pipeline.Store(":mem:dist-single", refs, device=0)
    st.build_index(10, False)
    want = _run_block(st, qs, 0, qs.n)
    st.close()
    return want


def _same(merged, want):
    assert [int(q) for q, _ in merged] == [q for q, _ in want]
    for (_, got), (_, exp) in zip(merged, want):
        assert tuple(got) == tuple(exp)


def test_rccl_start_up_path_world_1(tmp_path):
    port = _free_port()
    mp.spawn(_rank, args=(1, port, "nccl", str(tmp_path)), nprocs=1, join=True)
    _same(np.load(os.path.join(str(tmp_path), "merged.npy"), allow_pickle=True), _single_process_results())


def test_second_rank_receives_store_and_index_by_broadcast(tmp_path):
    port = _free_port()
    mp.spawn(_rank, args=(2, port, "gloo", str(tmp_path)), nprocs=2, join=True)
    merged = np.load(os.path.join(str(tmp_path), "merged.npy"), allow_pickle=True)
    want = _single_process_results()
    _same(merged, want)
    assert sum(1 for _, r in want if r[0] == 0) >= 20  # (the DP ran: not all copied / failed)
