"""world_size-2 run of the multi-rank plumbing on CPU (gloo): block sharding of the query
stream, the one start-up broadcast of the index, max-over-ranks timing, and the merge of
per-rank results by seqno.  The per-rank "hot path" here is the ORACLE (no GPU in this
container); what is under test is sina_amd.dist, which bench.py uses unchanged with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import pyoracle as po
    from sina_amd import dist as sdist
    from sina_amd import synth
    r, lr, w, dist = sdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    refs = synth.make_refs(150, length=200, width=1200, seed=61)       # same on every rank
    qs = synth.make_queries(refs, 11, seed=62)                          # the global query stream
    cs = [po.Cseq.from_packed("ref%d" % i, refs.seq(i), refs.width) for i in range(refs.n)]
    # rank 0 "builds" the index, the others receive it (CSR offsets + ids), like the RCCL broadcast
    nk = 4 ** 6 + 1
    if rank == 0:
        off, ids = po.Index(cs, k=6).csr()
        n_post = np.array([len(ids)], np.int64)
    else:
        off, ids, n_post = np.zeros(nk, np.uint32), None, np.zeros(1, np.int64)
    n_post = sdist.broadcast_array(n_post, 0, dist)
    if rank != 0:
        ids = np.zeros(int(n_post[0]), np.uint32)
    off = sdist.broadcast_array(off.view(np.int32), 0, dist).view(np.uint32)
    ids = sdist.broadcast_array(ids.view(np.int32), 0, dist).view(np.uint32)
    mine_off, mine_ids = po.Index(cs, k=6).csr()
    assert (off == mine_off).all() and (ids == mine_ids).all()
    # every rank runs the full path on its block of queries
    lo, hi = sdist.shard_range(qs.n, rank, world)
    idx = po.Index(cs, k=10)
    res = []
    for qi in range(lo, hi):
        m = qs.seq(qi)
        q = po.Cseq.from_packed("q%d" % qi, np.arange(len(m), dtype=np.uint32) | (m.astype(np.uint32) << 24), len(m))
        fids, _, _ = idx.famfinder(q, po.ff_opts(fs_min_len=50, fs_full_len=150))
        res.append((qi, po.align([cs[i] for i in fids], q)["aligned"]))
    t = sdist.reduce_max(1.0 + rank, dist)
    n = sdist.reduce_sum(len(res), dist)
    assert t == float(world) and n == qs.n
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(res, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_broadcast(tmp_path, oracle):
    from sina_amd import dist as sdist
    assert [sdist.shard_range(11, r, 2) for r in range(2)] == [(0, 6), (6, 11)]
    assert [sdist.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    merged = {}
    for r in range(2):
        for qi, al in np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True):
            merged[int(qi)] = al
    assert sorted(merged) == list(range(11))
    # single-process result of the same stream
    from sina_amd import synth
    refs = synth.make_refs(150, length=200, width=1200, seed=61)
    qs = synth.make_queries(refs, 11, seed=62)
    cs = [oracle.Cseq.from_packed("ref%d" % i, refs.seq(i), refs.width) for i in range(refs.n)]
    idx = oracle.Index(cs, k=10)
    for qi in range(qs.n):
        m = qs.seq(qi)
        q = oracle.Cseq.from_packed("q", np.arange(len(m), dtype=np.uint32) | (m.astype(np.uint32) << 24), len(m))
        fids, _, _ = idx.famfinder(q, oracle.ff_opts(fs_min_len=50, fs_full_len=150))
        assert merged[qi] == oracle.align([cs[i] for i in fids], q)["aligned"]


def _worker3(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sina_amd import dist as sdist
    r, lr, w, dist = sdist.init(backend="gloo")
    n = 20  # 20 queries over 3 ranks: blocks of 7, 7, 6
    lo, hi = sdist.shard_range(n, rank, world)
    # results arrive out of order inside a rank (batches finish as they finish): merged by seqno
    local = [(q, "aligned-%d-by-rank-%d" % (q, rank)) for q in reversed(range(lo, hi))]
    merged = sdist.merge_by_seqno(local, dist)
    total = sdist.reduce_sum(hi - lo, dist)
    slowest = sdist.reduce_max(float(hi - lo), dist)
    assert total == n and slowest == 7.0
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), np.array(merged, dtype=object), allow_pickle=True)
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_remainder_shards_and_ordered_merge(tmp_path):
    """World size 3 (gloo): the query stream does not divide evenly -- every query belongs to exactly one
    rank, block sizes differ by at most one -- and the per-rank results come back to rank 0 in sequence
    order (--preserve-order, src/sina.cpp:529-538)."""
    from sina_amd import dist as sdist
    blocks = [sdist.shard_range(20, r, 3) for r in range(3)]
    assert blocks == [(0, 7), (7, 14), (14, 20)]
    for n in (0, 1, 2, 3, 7, 100001):
        for w in (1, 2, 3, 8):
            b = [sdist.shard_range(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(y - x for x, y in b) - min(y - x for x, y in b) <= 1
    port = _free_port()
    mp.spawn(_worker3, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    merged = np.load(os.path.join(str(tmp_path), "merged.npy"), allow_pickle=True)
    assert [int(q) for q, _ in merged] == list(range(20))
    assert [str(p) for _, p in merged] == ["aligned-%d-by-rank-%d" % (q, 0 if q < 7 else (1 if q < 14 else 2)) for q in range(20)]
    assert sdist.merge_by_seqno([(2, "c"), (0, "a"), (1, "b")], None) == [(0, "a"), (1, "b"), (2, "c")]


def _worker_merge_at_size(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sina_amd import dist as sdist
    r, lr, w, dist = sdist.init(backend="gloo")
    n = 10000
    # round-robin ownership (the ranks' sequence numbers interleave), 6 KB of payload per query
    local = [(q, bytes([q % 251]) * 6144) for q in range(rank, n, world)]
    local.reverse()
    stats = {}
    merged = sdist.merge_by_seqno(local, dist, dst=1, stats=stats)
    if rank == 1:
        assert [q for q, _ in merged] == list(range(n))
        assert all(p == bytes([q % 251]) * 6144 for q, p in merged[::97])
        assert stats["received_items"] == n - len(local) and stats["sent_items"] == 0
    else:
        assert merged is None
        assert stats["received_items"] == 0 and stats["sent_items"] == len(local)   # nothing comes back to a sender
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_ordered_merge_is_a_gather_at_size(tmp_path):
    """--preserve-order at run size (src/sina.cpp:529-538): 10 000 results of 6 KB each from three ranks (gloo),
    sequence numbers interleaved, arrive at ONE rank in order; the other ranks send their share and receive
    nothing (until round 4 every rank received every rank's results)."""
    port = _free_port()
    mp.spawn(_worker_merge_at_size, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(3))
