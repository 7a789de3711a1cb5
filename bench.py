#!/usr/bin/env python3
"""bench.py -- aligned sequences/sec of the SINA hot path on MI355X.

Metric (BASELINE.json): aligned sequences / second, whole job, index build
excluded -- the definition SINA prints (reference src/sina.cpp:584-589).

Workload at N=1 (config.workload): BASELINE.json configs[1] -- full-length 16S
queries (~1500 bp) against a 100k-sequence SILVA-NR-like aligned reference
(synthetic clade model of SURVEY.md section 8d, alignment width 50 000, seed 2),
SINA default options.  One "step" = one batch of --batch queries through
famfinder (k-mer search + family selection) and aligner (family DAG, mesh DP,
backtrack, NAST fix-up).  Queries, references and index are resident before the
timed region; every rank runs its own K steps on its own queries (weak scaling,
no per-step collective; one RCCL broadcast of the index at start-up).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--refs R]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import sina_amd  # noqa: E402,F401  (first: the package sets the HIP runtime's hardware-queue default before anything starts it)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DP_BYTES_PER_CELL = 8          # two u32 trace-back indices per mesh cell (algorithmic)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=9216, help="queries per step and rank")
    ap.add_argument("--refs", type=int, default=100000)
    ap.add_argument("--length", type=int, default=1500)
    ap.add_argument("--width", type=int, default=50000)
    ap.add_argument("--window", type=int, default=0, help="cut queries to this many bases (V4: 250)")
    ap.add_argument("--inflight", type=int, default=6, help="batches worked on concurrently per rank")
    ap.add_argument("--exact-rate", type=float, default=0.0,
                    help="this share of the queries are UNMUTATED copies (windows) of their source reference: an amplicon run "
                         "against the database its organisms are in.  The aligner copies such a query's alignment from a "
                         "family member that contains it (src/align.cpp:349-388) -- no DP, but a string search per member")
    ap.add_argument("--sub-batch", type=int, default=9216,
                    help="queries per GPU launch inside a step (one DP wave per query; an MI355X has 3072 wave slots "
                         "at three waves per SIMD: 9216 = three rounds.  Every DP launch ends with ~4.4 ms in which its "
                         "last waves finish on a draining device -- tools/perf_dp.py: 3072 / 6144 / 9216 / 12288 queries "
                         "in 23.4 / 42.3 / 61.6 / 80.4 ms = 595 / 660 / 680 / 694 Gcell/s -- so longer launches are "
                         "faster per query; the trace-back planes (66 GB each at 9216 16S queries) come from the "
                         "device's pool of two, csrc/ctx.h)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="CPU baseline: queries per thread and thread count (0 = 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=64,
                    help="queries of the timed run re-done by the oracle afterwards and compared (0 = none; "
                         "skipped together with the CPU baseline, whose oracle index it shares)")
    ap.add_argument("--dup-rate", type=float, default=0.0,
                    help="fraction of every step's queries that repeat another query of the same step (amplicon-like "
                         "input; identical queries of a batch are searched and aligned once).  Default 0: the headline "
                         "workload has no repeats")
    ap.add_argument("--divergence-mix", action="store_true",
                    help="queries at 0.5 / 3 / 10 / 20 %% substitutions (indels in proportion), interleaved: every launch "
                         "mixes near-identical and distant queries (default: 3 %% throughout, the headline workload)")
    ap.add_argument("--confined-cpus", type=int, default=2,
                    help="N = 1 only: after the timed region the same number of steps runs once more with every thread of "
                         "the process confined to this many CPUs (an 8-rank node inside a 16-CPU quota leaves a rank two) "
                         "and --confined-threads loop-pool threads; the line reports confined_rate_frac = that rate / "
                         "the timed region's.  0 = skip")
    ap.add_argument("--confined-threads", type=int, default=2)
    ap.add_argument("--host-graph", action="store_true", help="build family DAGs on the host")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="threads of the host-side loop pool (0 = default: 6 for a single rank -- measured on one MI355X: "
                         "3 threads 140.1 k seq/s at 1.56 busy cores, 4: 141.8 k / 1.67, 6: 143.1 k / 1.69, 12: 143.9 k / "
                         "1.97, profiles/r04_host_threads.txt; with several ranks inside one CPU quota: 1.5 x the rank's share)")
    return ap.parse_args()


def kernel_source_rev():
    """Hash of the DP kernel's sources without comments and blank space: a recorded PMC profile only
    describes the code it was taken on (a comment may change, an instruction may not)."""
    import hashlib
    import re
    h = hashlib.sha1()
    for f in ("mesh_dp.hip", "common.h"):
        text = open(os.path.join(ROOT, "sina_amd", "csrc", f), "r").read()
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", " ", text).encode())
    return h.hexdigest()[:16]


def physical_cores():
    """Physical cores of this host (SMT siblings counted once)."""
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), or None if unlimited / unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


class OracleWorld:
    """The CPU oracle's view of the workload (test infrastructure): reference cseqs + k-mer index,
    built once and shared by the cpu_baseline and --verify legs.  Never part of the measured path."""

    def __init__(self, refs):
        from oracle import pyoracle as po
        self.po = po
        t0 = time.time()
        self.cs = [po.Cseq.from_packed("ref%d" % i, refs.seq(i), refs.width) for i in range(refs.n)]
        self.idx = po.Index(self.cs, k=10)
        self.build_s = time.time() - t0

    def query(self, qs, i):
        m = qs.seq(i)
        ab = np.arange(len(m), dtype=np.uint32) | (m.astype(np.uint32) << 24)
        return self.po.Cseq.from_packed("q%d" % i, ab, len(m))


def cpu_baseline(world, refs, qs, per_thread):
    """Times the oracle (CPU restatement of the reference algorithm, full 28-byte-cell mesh) on a
    bounded sample of the same workload, as SURVEY 8d / BASELINE.md 3 ask: one thread alone, a sweep
    over thread counts (one query per thread at a time like the reference's TBB nodes, pages
    interleaved over the NUMA nodes, every thread warmed with one untimed query) and the per-core
    rate x physical cores.  `value` is the FASTEST of those -- the extrapolation included, so the
    baseline errs on the CPU's side.  Reported beside the GPU number; never part of it."""
    po = world.po
    hw = os.cpu_count() or 1
    cores = physical_cores()
    counts = sorted({t for t in (16, 32, 64, 128, cores) if 1 < t <= hw})
    n_max = per_thread * max(counts + [1])
    queries = [world.query(qs, i) for i in range(min(qs.n, max(n_max, 8)))]
    t_all = time.time()
    one = po.bench_run(world.idx, queries[:6], 1)
    one_rate = one["aligned"] / one["seconds"]
    one_mcell = one["cells"] / one["seconds"] / 1e6
    sweep = []
    for th in counts:
        r = po.bench_run(world.idx, queries[:per_thread * th], th, interleave=True)
        sweep.append(dict(threads=th, seq_per_s=r["aligned"] / r["seconds"],
                          mcell_per_s_per_thread=r["cells"] / r["seconds"] / 1e6 / th, interleaved=r["interleaved"]))
    best = max(sweep, key=lambda x: x["seq_per_s"]) if sweep else dict(threads=1, seq_per_s=one_rate)
    extrapolated = one_rate * cores
    value = max(best["seq_per_s"], extrapolated)
    quota = cpu_quota()
    return dict(value=value, unit="sequences/s", cores=cores, kind="port", container_cpu_quota=quota,
                one_thread=dict(seq_per_s=one_rate, mcell_per_s=one_mcell),
                measured_best=best, per_core_rate_x_cores=extrapolated, sweep=sweep,
                sample="oracle (plain C restatement of the reference algorithm, full 28-byte-cell mesh) on the same "
                       "synthetic queries vs the same %d references: 1 thread %.1f seq/s = %.1f Mcell/s; best "
                       "measured %.0f seq/s at %d threads (%d queries per thread, NUMA-interleaved: %s); 1-thread "
                       "rate x %d physical cores = %.0f seq/s; value = the larger; %.1f s of CPU-baseline wall, "
                       "index build %.0f s not timed%s"
                       % (refs.n, one_rate, one_mcell, best["seq_per_s"], best["threads"], per_thread,
                          "yes" if any(x["interleaved"] for x in sweep) else "single node", cores, extrapolated,
                          time.time() - t_all, world.build_s,
                          ("; this container's cgroup grants %.0f CPUs of time, which is what bounds the measured "
                           "sweep -- the extrapolation is not bounded by it" % quota) if quota else ""))


def verify_against_oracle(world, qs, picked):
    """Untimed: re-does the picked queries of the timed run with the oracle (k-mer search + family
    selection + DAG + DP + backtrack + NAST at the full reference count) and compares family, aligned
    columns and case bits, head / tail / quality.  Returns (checked, identical, first differences)."""
    po = world.po
    diffs = []
    for q, got in sorted(picked.items()):
        c = world.query(qs, q)
        ids, sc, fflog = world.idx.famfinder(c, po.ff_opts())
        if len(ids) == 0:
            ok = got["status"] == 2
            what = "status"
        else:
            want = po.align([world.cs[i] for i in ids], c, po.align_opts())
            fam = "".join("ref%d.0:%.2f " % (i, x) for i, x in zip(ids, sc))
            checks = (("family", got["family"] == fam), ("status", got["status"] == want["status"]),
                      ("alignment", len(got["packed"]) == len(want["packed"]) and
                       bool((got["packed"] == want["packed"]).all())),
                      ("head/tail/qual", (got["head"], got["tail"], got["qual"]) ==
                       (want["head"], want["tail"], want["qual"])))
            ok = all(v for _, v in checks)
            what = ",".join(k for k, v in checks if not v)
        if not ok:
            diffs.append("query %d: %s" % (q, what))
    return len(picked), len(picked) - len(diffs), diffs[:4]


def self_launch(a):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks ourselves, as
    children of `torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1), hand their one
    JSON line through and leave with their exit code.  Decided before anything touches a GPU: this
    parent process never does (the reference's counterpart is `--threads`, src/sina.cpp:241-243,450)."""
    import socket
    import subprocess
    try:  # (counting devices does not start the HIP runtime; a node with fewer GPUs than asked for is said so here)
        import torch
        have = torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        have = None
    if have is not None and have < a.gpus:
        print("bench.py --gpus %d: this node shows %d GPU(s)" % (a.gpus, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(a))
    import torch
    from sina_amd import dist as sdist
    from sina_amd import pipeline, synth

    rank, local_rank, world, dist = sdist.init()
    if dist is None and os.environ.get("SINA_BENCH_FORCE_DIST"):
        # exercise the RCCL start-up path on a single GPU (world size 1): process group, device
        # tensor views over the C ABI's buffers, in-place broadcast
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        tdist.init_process_group(backend="nccl", rank=0, world_size=1)
        dist = tdist
    if world != a.gpus:
        # (a launcher's WORLD_SIZE that disagrees with --gpus: the line below says n_gpus = WORLD_SIZE, the
        # ranks that exist -- never the number that was asked for)
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # host threads of this rank on a block of cores of the GPU's NUMA node (sina_amd/affinity.py);
    # before the host library creates its threads
    from sina_amd import affinity
    full_mask = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = affinity.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    # ---- synthetic inputs (identical on every rank; queries differ per rank)
    refs = synth.make_refs(a.refs, length=a.length, width=a.width, seed=2)
    # + one untimed step for the isolated kernel timings + the set-up pass (launches of the timed size)
    prime_n = a.sub_batch * 2 * max(1, a.inflight)
    n_q = a.batch * (a.steps + a.warmup + 1) + prime_n
    window = (1.0 / 3.0, a.window) if a.window else None
    mix = dict(sub=[0.005, 0.03, 0.10, 0.20], dele=[0.001, 0.005, 0.015, 0.03], ins=[0.001, 0.003, 0.01, 0.02]) if a.divergence_mix else {}
    if a.exact_rate > 0 and not mix:  # (query i of every hundred: exact if i < 100 x the rate)
        k = int(round(100 * min(1.0, a.exact_rate)))
        mix = dict(sub=[0.0] * k + [0.03] * (100 - k), dele=[0.0] * k + [0.005] * (100 - k), ins=[0.0] * k + [0.003] * (100 - k))
    qs = synth.make_queries(refs, n_q, seed=3 + 1000 * rank, window=window, **mix)
    if a.dup_rate > 0:
        qs = synth.with_repeats(qs, a.dup_rate, a.batch, seed=17 + rank)

    # ---- resident state: references + index in HBM, stages constructed
    # (ranks other than 0 do not upload: the references arrive with the index, by broadcast)
    store = pipeline.Store(":mem:bench", refs, device=local_rank, upload=(dist is None or rank == 0))
    t_idx = time.time()
    if dist is not None:
        n_post = sdist.broadcast_device_index(store, 10, False, rank, dist, device)
    else:
        store.build_index(10, False)
        n_post = None
    idx_s = time.time() - t_idx
    al_opts = {"device-graph": not a.host_graph}
    # loop-pool threads per rank: the library's default (12) when the rank has the host to itself; with
    # several ranks inside one CPU quota, no more threads than the rank's share can run (measured on one
    # GPU: 6 threads 128.8 k seq/s at 3.3 busy cores, 12 threads 131.1 k at 3.8)
    host_threads = a.host_threads or None
    if host_threads is None and world == 1:
        host_threads = 6
    if host_threads is None and world > 1:
        quota = cpu_quota()
        share = (quota if quota else (os.cpu_count() or 16)) / float(world)
        host_threads = int(max(4, min(12, round(1.5 * share))))
    pl = pipeline.Pipeline(store, aligner=al_opts, host_threads=host_threads)

    def run_steps(first, count):
        lo, hi = qs.off[first * a.batch], qs.off[(first + count) * a.batch]
        off = (qs.off[first * a.batch:(first + count) * a.batch + 1] - lo).astype(np.uint64)
        return pl.run(qs.mask[lo:hi], off, batch=a.sub_batch, inflight=a.inflight)

    # Set-up, like the index build: one pass that lets every context of the pipeline allocate its
    # scratch (a 48 GB trace-back plane takes 1.4 s to allocate; with few warm-up steps not every
    # aligner context would have seen a batch before the timed region).  Its own queries; launches of
    # the timed size, two per aligner context.
    q_p = a.batch * (a.steps + a.warmup + 1)  # its own queries, behind those of the steps
    lo_p, hi_p = qs.off[q_p], qs.off[q_p + prime_n]
    off_p = (qs.off[q_p:q_p + prime_n + 1] - lo_p).astype(np.uint64)
    pl.run(qs.mask[lo_p:hi_p], off_p, batch=a.sub_batch, inflight=a.inflight)
    if a.warmup:
        run_steps(0, a.warmup)
    pl.profile(reset=True)
    s0 = store.stats()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)

    def thread_times():  # (SINA_HOST_PROFILE: which threads the region's CPU time belongs to)
        out = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/stat" % tid).read()
                rest = f[f.rindex(")") + 2:].split()
                out[tid] = (int(rest[11]), int(rest[12]), int(rest[7]))
            except OSError:
                pass
        return out
    tt0 = thread_times() if os.environ.get("SINA_HOST_PROFILE") else None
    if os.environ.get("SINA_HIP_TRACE_ALLOC"):
        print("[bench] %.3f timed region starts" % (time.clock_gettime(time.CLOCK_MONOTONIC) % 1000), file=sys.stderr)
    if os.environ.get("SINA_HOST_TRACE"):
        pipeline.load_host().sina_host_profile_mark(0)
    t0 = time.time()
    timing = run_steps(a.warmup, a.steps)
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    elapsed = time.time() - t0
    if os.environ.get("SINA_HOST_TRACE"):
        pipeline.load_host().sina_host_profile_mark(1)
    if os.environ.get("SINA_HIP_TRACE_ALLOC"):
        print("[bench] %.3f timed region ends" % (time.clock_gettime(time.CLOCK_MONOTONIC) % 1000), file=sys.stderr)
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    if tt0 is not None and rank == 0:
        tt1, tck = thread_times(), os.sysconf("SC_CLK_TCK")
        rows = sorted(((tt1[t][0] - tt0[t][0] + tt1[t][1] - tt0[t][1], t) for t in tt1 if t in tt0), reverse=True)
        for tot, t in rows[:20]:
            print("timed region: thread tid %-8s user %6.2f s  kernel %6.2f s  minor faults %8d%s" % (
                t, (tt1[t][0] - tt0[t][0]) / tck, (tt1[t][1] - tt0[t][1]) / tck, tt1[t][2] - tt0[t][2],
                "  (main)" if int(t) == os.getpid() else ""), file=sys.stderr)
        print("timed region: whole process user %.2f s kernel %.2f s over %.2f s (threads that started and ended "
              "inside it are in the 'thread: ... (whole run)' lines)" % (
                  ru1.ru_utime - ru0.ru_utime, ru1.ru_stime - ru0.ru_stime, elapsed), file=sys.stderr)
    host_cores = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / elapsed
    host_cores_sys = (ru1.ru_stime - ru0.ru_stime) / elapsed
    host_minor_faults = (ru1.ru_minflt - ru0.ru_minflt) / elapsed
    host_ctx_switches = ((ru1.ru_nvcsw - ru0.ru_nvcsw) + (ru1.ru_nivcsw - ru0.ru_nivcsw)) / elapsed
    s1 = store.stats()

    n_done = a.batch * a.steps
    n_aligned = sum(1 for q in range(n_done) if pl.result(q)["status"] in (0, 1))
    picked = {}
    if a.verify and not a.no_cpu_baseline and rank == 0 and world == 1:
        rng = np.random.default_rng(12345)
        first = a.warmup * a.batch  # (results are indexed from the first query of the timed call)
        for q in rng.choice(n_done, size=min(a.verify, n_done), replace=False):
            picked[first + int(q)] = pl.result(int(q))

    # One more step, untimed and with ONE batch in flight: kernels run alone on the GPU, so their
    # HIP-event durations are the kernels' own (in the timed region batches overlap on separate
    # streams and every kernel's duration includes the time it shares the CUs with the others).
    lo, hi = qs.off[(a.warmup + a.steps) * a.batch], qs.off[(a.warmup + a.steps + 1) * a.batch]
    off = (qs.off[(a.warmup + a.steps) * a.batch:(a.warmup + a.steps + 1) * a.batch + 1] - lo).astype(np.uint64)
    pl.run(qs.mask[lo:hi], off, batch=a.sub_batch, inflight=1)
    s2 = store.stats()
    iso = {k: s2[k] - s1[k] for k in s1}

    # ---- the host budget of an 8-rank node, proven on this one GPU: the same steps once more with the whole
    # process (loop pool, stage driver threads, the HIP runtime's helpers) on --confined-cpus CPUs
    confined = None
    if world == 1 and a.confined_cpus > 0 and hasattr(os, "sched_setaffinity"):
        def all_tasks():
            return [int(t) for t in os.listdir("/proc/self/task")]
        allowed = sorted(os.sched_getaffinity(0))
        cpus = set(allowed[:a.confined_cpus])
        before = {}
        for tid in all_tasks():
            try:
                before[tid] = os.sched_getaffinity(tid)
                os.sched_setaffinity(tid, cpus)
            except OSError:
                pass
        pl._set("host", "threads", a.confined_threads)
        try:
            run_steps(a.warmup, min(2, a.steps))  # (the pool's new threads, the caches on the new CPUs)
            ru_a = resource.getrusage(resource.RUSAGE_SELF)
            t_c = time.time()
            run_steps(a.warmup, a.steps)
            torch.cuda.synchronize(device)
            dt_c = time.time() - t_c
            ru_b = resource.getrusage(resource.RUSAGE_SELF)
            n_c = sum(1 for q in range(n_done) if pl.result(q)["status"] in (0, 1))
            confined = {"cpus": len(cpus), "pool_threads": a.confined_threads, "sequences_per_s": n_c / dt_c,
                        "host_cores_busy": ((ru_b.ru_utime - ru_a.ru_utime) + (ru_b.ru_stime - ru_a.ru_stime)) / dt_c,
                        "rate_frac": (n_c / dt_c) / (n_aligned / elapsed)}
        finally:
            for tid in all_tasks():
                try:
                    os.sched_setaffinity(tid, before.get(tid, set(allowed)))
                except OSError:
                    pass
            pl._set("host", "threads", host_threads or 12)  # (12: the library's own default)

    per_rank = None
    if dist is not None:
        # per rank: its own rate and host load (the job's rate below is total work / slowest rank's time)
        mine = torch.tensor([n_aligned / elapsed, host_cores, host_cores_sys], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(max(world, 1))]
        dist.all_gather(every, mine)
        per_rank = [[float(x) for x in t.cpu()] for t in every]
        elapsed = sdist.reduce_max(elapsed, dist, device)
        n_aligned = int(sdist.reduce_sum(n_aligned, dist, device))
    # DP launches are chained (csrc/ctx.h, heavy_launch): a launch starts when the launch before it has dispatched
    # its last workgroup, so consecutive DP launches overlap while the older one drains.  dp_busy_ms = the time
    # during which a DP kernel was resident (sum of the launches' start-to-end durations minus their overlaps):
    # the denominator of the kernel's throughput.  dp_ms_sum = the plain sum of start-to-end durations, what a
    # kernel trace's "average duration" multiplies out to; it counts every overlap twice.
    dp_ms_sum = s1["dp_ms"] - s0["dp_ms"]
    dp_ms = s1["dp_busy_ms"] - s0["dp_busy_ms"]
    dp_cells = s1["dp_cells"] - s0["dp_cells"]            # nominal: N x L of every query (what the reference fills)
    dp_launches = s1["dp_launches"] - s0["dp_launches"]
    # The DP kernel skips rows of a strip that provably cannot hold a cell of the optimal path (certified: csrc/mesh_dp.hip
    # PRUNE): the roofline is priced on the cells it actually COMPUTED -- a failed certificate's second
    # sweep included --, the nominal count is reported beside it.
    dp_cells_swept = s1["dp_cells_swept"] - s0["dp_cells_swept"]
    dp_rows, dp_rows_swept = s1["dp_rows"] - s0["dp_rows"], s1["dp_rows_swept"] - s0["dp_rows_swept"]
    achieved = DP_BYTES_PER_CELL * dp_cells_swept / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0

    # HBM bytes per DP launch from the PMC passes of this same command (tools/prof_bench.sh ->
    # profiles/r06_traffic.json: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE), and the DP kernel's VALU
    # wave-instructions per COMPUTED cell from its SQ pass (tools/prof_dp_pmc.sh -> profiles/r06_dp_valu.json);
    # both only if they were recorded on this kernel source revision, else null
    dp_kernel_name = "mesh_dp_simple_kernel"  # (SINA defaults: simple scheme, gap_open >= gap_extend; mesh_dp.hip)
    dp_traffic, traffic_note = None, "no PMC profile recorded for this kernel source + configuration"
    other_traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r06_traffic.json")))
        meta = tj.get("_meta", {})
        if (meta.get("kernel_source_rev") == kernel_source_rev() and meta.get("batch") == a.batch and
                meta.get("sub_batch") == a.sub_batch and meta.get("refs") == a.refs and
                meta.get("length") == a.length and meta.get("window") == a.window):
            dp_traffic = tj[dp_kernel_name]["hbm_bytes"]
            other_traffic = {k: tj[k] for k in ("family_graph_kernel", "kmer_count_kernel") if k in tj}
            for short, key in (("chain_scout_kernel", "chain_scou"), ("backtrack_kernel", "backtrack_")):  # (the trace's truncated names)
                for k in tj:
                    if key in k:
                        other_traffic[short] = tj[k]
            traffic_note = ("HBM bytes per launch, FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, from the separate "
                            "--pmc passes of this same command recorded in profiles/r06_traffic.json (same kernel "
                            "source revision and configuration as this run; not measured by this run)")
    except Exception:
        pass
    valu_per_cell = None
    try:
        vj = json.load(open(os.path.join(ROOT, "profiles", "r06_dp_valu.json")))
        if vj.get("kernel_source_rev") == kernel_source_rev():
            valu_per_cell = float(vj["valu_wave_instructions_per_cell"])
    except Exception:
        pass
    verify_failed = False
    if rank == 0 and os.environ.get("SINA_HOST_TRACE") and not os.environ.get("SINA_HOST_PROFILE"):
        pl.profile()  # (writes the trace file)
    if rank == 0 and os.environ.get("SINA_HOST_PROFILE"):
        print(pl.profile(), file=sys.stderr)
        # CPU time of every thread of this process so far (the stage driver's threads are gone by now;
        # what is left are the loop pool, the interpreter and the HIP / ROCr runtime's own threads)
        try:
            tck = os.sysconf("SC_CLK_TCK")
            rows = []
            for tid in os.listdir("/proc/self/task"):
                f = open("/proc/self/task/%s/stat" % tid).read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                sw = {}
                for ln in open("/proc/self/task/%s/status" % tid):
                    if "ctxt_switches" in ln:
                        sw[ln.split(":")[0]] = int(ln.split(":")[1])
                rows.append(((int(rest[11]) + int(rest[12])) / tck, int(rest[11]) / tck, int(rest[12]) / tck, comm, tid,
                             int(rest[7]), sw.get("voluntary_ctxt_switches", 0), sw.get("nonvoluntary_ctxt_switches", 0)))
            for tot, u, k, comm, tid, minflt, vol, invol in sorted(rows, reverse=True)[:24]:
                print("thread %-18s tid %-8s user %7.2f s  kernel %7.2f s  minor faults %8d  ctx switches %8d vol %6d invol"
                      % (comm, tid, u, k, minflt, vol, invol), file=sys.stderr)
        except Exception as e:  # noqa: BLE001
            print("thread times unavailable: %s" % e, file=sys.stderr)
    if rank == 0:
        out = {
            "metric": "aligned sequences/sec (whole node), 100k full-length 16S vs SILVA-NR-scale ref",
            "value": n_aligned / elapsed,
            "unit": "sequences/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s (~%d bp%s) vs %d-seq SILVA-NR-like aligned reference, "
                            "width %d, SINA defaults (k=10 fast, family 40, match 2/mismatch -1/gap 5/ext 2)"
                            % ("configs[2] shape: V4 amplicons cut from full-length 16S" if a.window else
                               ("configs[4] shape: full-length 23S" if a.length >= 2500 else
                                ("configs[3] shape: full-length 16S, large reference" if a.refs >= 400000 else
                                 "configs[1]: full-length 16S")),
                               a.length, (", cut to %d" % a.window) if a.window else "", a.refs, a.width),
                "refs": a.refs, "length": a.length, "width": a.width, "window": a.window, "dup_rate": a.dup_rate, "exact_rate": a.exact_rate, "divergence_mix": bool(a.divergence_mix),
                "queries_per_step_per_gpu": a.batch,
                "queries_per_launch": a.sub_batch,
                "inflight_batches": a.inflight,
                "family_dag": "host" if a.host_graph else "device",
                "sharding": "queries block-sharded over %d rank(s); index %s" %
                            (world, "RCCL-broadcast from rank 0" if world > 1 else "built on device"),
                "index_build_s": idx_s,
            },
            "roofline": {
                "kernel": dp_kernel_name,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": dp_traffic,
                "traffic_unit": traffic_note,
                "algorithmic_bytes_per_launch": DP_BYTES_PER_CELL * dp_cells_swept / dp_launches if dp_launches else 0,
                "cells_computed_per_launch": dp_cells_swept / dp_launches if dp_launches else 0,
                "cells_nominal_per_launch": dp_cells / dp_launches if dp_launches else 0,
                "cells_computed_frac": dp_cells_swept / dp_cells if dp_cells else 0.0,
                "wave_rows_computed_frac": dp_rows_swept / dp_rows if dp_rows else 0.0,
                "ms_per_launch": dp_ms / dp_launches if dp_launches else 0,
                "ms_per_launch_start_to_end": dp_ms_sum / dp_launches if dp_launches else 0,
                "frac_by_start_to_end": (DP_BYTES_PER_CELL * dp_cells_swept / (dp_ms_sum * 1e-3) / 1e9 / HBM_PEAK_GBS
                                         if dp_ms_sum > 0 else 0.0),
                "gcells_per_s": dp_cells_swept / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0,
                "gcells_per_s_nominal": dp_cells / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0,
                "row_skip": {
                    "queries": s1["dp_queries_pruned"] - s0["dp_queries_pruned"],
                    "second_attempts": s1["dp_second_attempts"] - s0["dp_second_attempts"],
                    "full_sweeps": s1["dp_full_sweeps"] - s0["dp_full_sweeps"],
                    "guess_rho": s1["dp_prune_rho"],
                    "scout_launches": s1["scout_launches"] - s0["scout_launches"],
                    "scout_ms_per_launch": ((s1["scout_ms"] - s0["scout_ms"]) / (s1["scout_launches"] - s0["scout_launches"])
                                            if s1["scout_launches"] > s0["scout_launches"] else 0.0),
                    "what": "certified-exact: a (row, 512-column strip) is swept only if a cell in it can still lie on a "
                            "path ending at or below the query's bound U (value <= U + bound on the gain still to come; U = "
                            "the cost of the query's alignment against the chain of its family's first member -- the scout "
                            "pass, a real path of the mesh -- guarded by the store's learnt guess); "
                            "certificate: the end cell found has value <= U, else the query is swept again (second_attempts: "
                            "under the bound the first attempt found; full_sweeps: without one).  cells_computed counts "
                            "every sweep; results are bit-identical to the full sweep (tests/test_gpu_prune.py, verify)",
                },
                "note": "contractual accounting (SURVEY 8d): 8 algorithmic bytes per COMPUTED mesh cell against the HBM peak "
                        "(cells_nominal = N x L of every query, what the reference fills, is reported beside it and is NOT "
                        "what achieved / frac are priced on); "
                        "the kernel writes 2 B per cell and is bound by VALU issue, see roofline_valu.  Timed "
                        "region: HIP events around every launch on the FIFO streams it runs on.  Launches are chained: "
                        "a DP launch starts when the one before it has DISPATCHED its last workgroup, so two DP "
                        "launches share the device while the older one drains; ms_per_launch / achieved / frac "
                        "count that shared time once (time with a DP kernel resident / launches), "
                        "ms_per_launch_start_to_end / frac_by_start_to_end count it in both launches (what a kernel "
                        "trace's average duration gives; tools/kt_union.py on the committed trace gives both); "
                        "`isolated` = one extra untimed step with a single batch in flight, nothing overlapping",
                "isolated": {
                    "achieved": DP_BYTES_PER_CELL * iso["dp_cells_swept"] / (iso["dp_ms"] * 1e-3) / 1e9,
                    "frac": DP_BYTES_PER_CELL * iso["dp_cells_swept"] / (iso["dp_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "ms_per_launch": iso["dp_ms"] / max(1, iso["dp_launches"]),
                    "gcells_per_s": iso["dp_cells_swept"] / (iso["dp_ms"] * 1e-3) / 1e9,
                    "gcells_per_s_nominal": iso["dp_cells"] / (iso["dp_ms"] * 1e-3) / 1e9,
                },
            },
            # what actually binds the kernel: VALU wave-instructions (SQ_INSTS_VALU of the same kernel source,
            # profiles/r04_dp_sq_counters.txt) at the guide's 2 cycles per wave-instruction and SIMD
            "roofline_valu": None if valu_per_cell is None or dp_ms <= 0 else {
                "kernel": dp_kernel_name,
                "bound": "valu",
                "achieved": valu_per_cell * dp_cells_swept / (dp_ms * 1e-3) / 1e12,
                "peak": 1024 * 2.4e9 / 2 / 1e12,
                "unit": "T wave-instructions/s",
                "frac": valu_per_cell * dp_cells_swept / (dp_ms * 1e-3) / (1024 * 2.4e9 / 2),
                "valu_wave_instructions_per_cell": valu_per_cell,
                "note": "SQ_INSTS_VALU per launch / cells per launch of the DP kernel alone (tools/prof_dp_pmc.sh), "
                        "x this run's cells; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave-instruction "
                        "(MI355X_MICROARCH.md).  At three waves per SIMD the instruction classes of this kernel "
                        "issue at 3.5 (add / select / logic) and 5.5 (compare / min / DPP) cycles per wave-instruction "
                        "(profiles/r03_valu_issue_rates.txt): at those measured rates the kernel's mix would need ~1.5 ns "
                        "per wave-instruction and SIMD, it takes ~2.4 -- the rest is dependency stalls and scalar work that "
                        "three waves per SIMD do not hide (DESIGN.md 3.1, 8)",
            },
            # the other device-filling kernels against the HBM roofline: what they MOVE (FETCH_SIZE x2 + WRITE_SIZE of the
            # PMC passes recorded on this kernel source revision, profiles/r06_traffic.json; null without such a record)
            # over their time alone (the untimed step with one batch in flight) -- a fraction of what the memory system can
            # deliver, never above 1.  SURVEY 8d's algorithmic bytes are reported beside it: the k-mer count moves LESS than
            # those (dense lists are read as bitmaps, the score rows never leave LDS), the DAG build more (2.3 x).  Both are
            # bound by instruction issue, not bytes.
            "roofline_other": {
                name: (lambda tr, alg, ms_alone, ms_pipe, launches, note: None if launches <= 0 else {
                    "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                    "measured_bytes_per_launch": tr["hbm_bytes"] if tr else None,
                    "achieved": (tr["hbm_bytes"] / (ms_alone * 1e-3) / 1e9) if tr and ms_alone > 0 else None,
                    "frac": (tr["hbm_bytes"] / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr and ms_alone > 0 else None,
                    "ms_per_launch_alone": ms_alone, "ms_per_launch_in_pipeline": ms_pipe / launches,
                    "algorithmic_bytes_per_launch": (alg / launches) if alg is not None else None,
                    "algorithmic_over_measured": (alg / launches / tr["hbm_bytes"]) if alg is not None and tr and tr["hbm_bytes"] > 0 else None,
                    "note": note})(
                    other_traffic.get(name), alg, ms_alone, ms_pipe, launches, note)
                for name, alg, ms_alone, ms_pipe, launches, note in (
                    ("family_graph_kernel", s1["graph_bytes"] - s0["graph_bytes"], iso["graph_ms"] / max(1, iso["graph_launches"]), s1["graph_ms"] - s0["graph_ms"],
                     s1["graph_launches"] - s0["graph_launches"],
                     "algorithmic: the families' packed bases read once, the DAGs written once; in the chained pipeline the build "
                     "starts in a DP launch's drain, so its start-to-end time there is longer than alone"),
                    ("kmer_count_kernel", 4.0 * (s1["postings"] - s0["postings"]) + 4.0 * a.refs * (s1["kmer_queries"] - s0["kmer_queries"]),
                     iso["kmer_count_ms"] / max(1, iso["kmer_launches"]), s1["kmer_count_ms"] - s0["kmer_count_ms"], s1["kmer_launches"] - s0["kmer_launches"],
                     "algorithmic: 4 B per posting of the query's k-mers + 2 x 2 B per reference for the score row; measured bytes are "
                     "fabric traffic (the bitmaps of dense lists come out of the MALL); VALU at 0.90 of the pipe "
                     "(profiles/r05_kmer_sq_counters.txt)"),
                    # the two kernels with ONE LANE per query (144 waves per 9216 queries): a chain of dependent loads per
                    # lane, bound by latency -- they run beside the device-filling kernels, off the FIFO's critical path
                    ("chain_scout_kernel", None, iso["scout_ms"] / max(1, iso["scout_launches"]), s1["scout_ms"] - s0["scout_ms"],
                     s1["scout_launches"] - s0["scout_launches"],
                     "one lane per query walking its nearest relative's chain (band of 8 columns in registers): latency-bound by "
                     "construction, 144 waves per launch; its time in the pipeline is beside the other batches' kernels"),
                    ("backtrack_kernel", None, iso["backtrack_ms"] / max(1, iso["dp_launches"]), s1["backtrack_ms"] - s0["backtrack_ms"],
                     s1["dp_launches"] - s0["dp_launches"],
                     "one lane per query walking its trace-back cells (a dependent 2-byte read per step; 74 MB of path cells per "
                     "launch are useful, a 64-byte sector holds 32 columns of ONE row): latency-bound, runs beside the next launch"))
            },
            "kernels_ms_per_step_isolated": {
                "kmer_count_kernel": iso["kmer_count_ms"],
                "kmer_select_kernel": iso["kmer_select_ms"],
                "graph_kernel": iso["graph_ms"],
                "mesh_dp_kernel": iso["dp_ms"],
                "backtrack_kernel": iso["backtrack_ms"],
            },
            "confined_rate_frac": confined["rate_frac"] if confined else None,  # rate on --confined-cpus CPUs / rate above
            "confined": confined,
            # family DAGs the device built per query aligned by DP (queries with the same ordered family share one)
            "family_dags_built_per_query": ((s1["dags_built"] - s0["dags_built"]) / max(1, s1["dags_used"] - s0["dags_used"])),
            "host_cores_busy": host_cores,  # CPU seconds per wall second of this rank in the timed region
            "host_cpus_pinned": pinned,     # logical CPUs this rank's host threads are confined to (None: not pinned)
            "host_pool_threads": host_threads,
            "host_cores_busy_kernel_mode": host_cores_sys,
            "host_minor_faults_per_s": host_minor_faults,
            "host_context_switches_per_s": host_ctx_switches,
            "stages_ms_per_step": {
                "famfinder_host_wall": 1e3 * timing["famfinder_s"] / a.steps,
                "aligner_host_wall": 1e3 * timing["aligner_s"] / a.steps,
                "kmer_count_kernel": (s1["kmer_count_ms"] - s0["kmer_count_ms"]) / a.steps,
                "kmer_select_kernel": (s1["kmer_select_ms"] - s0["kmer_select_ms"]) / a.steps,
                "graph_kernel": (s1["graph_ms"] - s0["graph_ms"]) / a.steps,
                "mesh_dp_kernel": dp_ms / a.steps,
                "mesh_dp_kernel_start_to_end": dp_ms_sum / a.steps,
                "backtrack_kernel": (s1["backtrack_ms"] - s0["backtrack_ms"]) / a.steps,
            },
        }
        quota = cpu_quota()
        out["container_cpu_quota"] = quota
        if per_rank is not None:
            rates = [r[0] for r in per_rank]
            out["per_rank"] = {"sequences_per_s": rates, "min": min(rates), "max": max(rates),
                               "host_cores_busy": [r[1] for r in per_rank],
                               "host_cores_busy_kernel_mode": [r[2] for r in per_rank]}
            need = sum(r[1] for r in per_rank)
            if quota and need > 0.9 * quota:
                out["per_rank"]["warning"] = ("the ranks keep %.1f host cores busy and the container's CPU quota is %.1f: "
                                              "the job is host-bound, not GPU-bound" % (need, quota))
                print("warning: " + out["per_rank"]["warning"], file=sys.stderr)
        if not a.no_cpu_baseline and world == 1:
            if pinned and full_mask:
                os.sched_setaffinity(0, full_mask)  # the CPU baseline gets the whole machine
            ow = OracleWorld(refs)
            out["cpu_baseline"] = cpu_baseline(ow, refs, qs, a.cpu_sample or 4)
            if picked:
                n_chk, n_same, diffs = verify_against_oracle(ow, qs, picked)
                out["verify"] = {"checked": n_chk, "identical": n_same, "differences": diffs,
                                 "what": "randomly picked queries of the timed run re-done by the CPU oracle at the "
                                         "full reference count: family, aligned columns + case bits, head/tail/quality"}
                verify_failed = n_same != n_chk
    pl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stderr.flush()
        try:  # (what native libraries -- RCCL's "Librccl path" -- left in the C stdio buffer comes out BEFORE the line)
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)  # the ONE JSON line, after RCCL has said whatever it says
        if verify_failed:
            raise SystemExit("bench.py --verify: results differ from the oracle: %s" % out["verify"]["differences"])


if __name__ == "__main__":
    main()
