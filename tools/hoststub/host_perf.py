#!/usr/bin/env python3
"""CPU cost of the HOST side of the boundary per query, measured without a GPU.

Loads tools/hoststub/libsina_hip.so -- a stub that answers the C-ABI calls with made-up results in no time
(fake_hip.cpp) -- in place of the real library and pushes bench-shaped queries through
sina_host_pipeline_run.  What is left is the host stages' own work: trays, famfinder's cascade and family
text, the aligner's glue, the sink.  Prints CPU microseconds per query (all threads, user + kernel) and,
with SINA_HOST_PROFILE=1, the per-phase table.  Results are NOT alignments: nothing here checks anything.

    make -C tools/hoststub && python3 tools/hoststub/host_perf.py [--queries 36864] [--refs 20000] [--threads 4]
"""
import argparse
import ctypes as C
import os
import resource
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=36864)
ap.add_argument("--refs", type=int, default=20000)
ap.add_argument("--length", type=int, default=1500)
ap.add_argument("--batch", type=int, default=9216)
ap.add_argument("--inflight", type=int, default=4)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--window", type=int, default=0, help="cut queries to this many bases (V4: 250)")
a = ap.parse_args()

C.CDLL(os.path.join(HERE, "libsina_hip.so"), mode=C.RTLD_GLOBAL)  # (its soname satisfies the host library's NEEDED)
from sina_amd import capi, pipeline, synth  # noqa: E402

capi.load = lambda: None  # (the stub is in; the real library must not be)
pipeline.HOST_LIB_PATH = os.path.join(HERE, "libsina_host.so")

refs = synth.make_refs(a.refs, length=a.length, width=50000, seed=1)
qs = synth.make_queries(refs, a.queries, seed=2, window=(1.0 / 3.0, a.window) if a.window else None)
st = pipeline.Store(":mem:hostperf", refs, device=0)
pl = pipeline.Pipeline(st, host_threads=a.threads or None)
pl.run(qs.mask, qs.off, batch=a.batch, inflight=a.inflight)  # warm-up: object caches, arenas
if os.environ.get("SINA_HOST_PROFILE"):
    pl.profile(reset=True)
for r in range(a.repeat):
    if r == a.repeat - 1 and os.environ.get("SINA_HOST_PROFILE"):
        pl.profile(reset=True)  # (the table below: the last run alone)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    pl.run(qs.mask, qs.off, batch=a.batch, inflight=a.inflight)
    t1 = time.perf_counter()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    print("run %d: %.2f us CPU per query (%.2f kernel-mode), wall %.3f s = %.0f queries/s, minor faults %d"
          % (r, 1e6 * cpu / a.queries, 1e6 * (ru1.ru_stime - ru0.ru_stime) / a.queries, t1 - t0,
             a.queries / (t1 - t0), ru1.ru_minflt - ru0.ru_minflt))
if os.environ.get("SINA_HOST_PROFILE"):
    print(pl.profile(reset=True))
n_ok = sum(pl.result(q)["status"] == 0 for q in range(0, a.queries, 997))
print("spot check: %d of %d sampled results aligned" % (n_ok, len(range(0, a.queries, 997))))
