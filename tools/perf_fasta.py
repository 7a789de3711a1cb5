"""FASTA I/O speed at alignment width 50 000.
usage: tools/perf_fasta.py [n_sequences]            reader -> writer alone (no GPU): sequences/s and MB/s
       tools/perf_fasta.py align [n_queries] [refs]  unaligned FASTA in -> famfinder -> aligner -> aligned FASTA out on the
                                                     GPU, through the concurrent driver and the one-batch-at-a-time one"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sina_amd import synth, pipeline
tmp = tempfile.mkdtemp(prefix="perf_fasta_")
path, out = os.path.join(tmp, "in.fasta"), os.path.join(tmp, "out.fasta")
if len(sys.argv) > 1 and sys.argv[1] == "align":
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    nr = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
    refs = synth.make_refs(nr, length=1500, width=50000, seed=2)
    qs = synth.make_queries(refs, nq, seed=3)
    with open(path, "w") as f:
        for i in range(qs.n):
            f.write(">query%d\n%s\n" % (i, synth.bases_string(qs.seq(i))))
    st = pipeline.Store(":mem:perf-fasta", refs)
    st.build_index(10, False)
    pipeline.run_fasta(st, path, out, batch=4096)   # (set-up: contexts, scratch, trace-back planes)
    for name, kw in (("concurrent stages", {}), ("one batch at a time", {"serial": True}), ("concurrent stages", {}),
                     ("concurrent stages, --fasta-write-dots", {"fasta": {"fasta-write-dots": True}})):
        t = time.time()
        got = pipeline.run_fasta(st, path, out, batch=4096, **kw)
        dt = time.time() - t
        print("%-40s %d sequences in %.2f s -> %.0f seq/s, %.0f MB/s out" % (
            name, got["written"], dt, got["written"] / dt, os.path.getsize(out) / 1e6 / dt))
    st.close()
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    refs = synth.make_refs(n, length=1500, width=50000, seed=2)
    with open(path, "w") as f:
        for i in range(refs.n):
            f.write(">seq%d\n%s\n" % (i, synth.aligned_string(refs.seq(i), refs.width)))
    for rep in range(3):
        t = time.time()
        got, sk = pipeline.fasta_roundtrip(path, out)
        dt = time.time() - t
        print("read + write: %d sequences in %.2f s -> %.0f seq/s, %.0f MB/s in, %.0f MB/s out" % (
            got, dt, got / dt, os.path.getsize(path) / 1e6 / dt, os.path.getsize(out) / 1e6 / dt))
for f in (path, out):
    if os.path.exists(f):
        os.remove(f)
os.rmdir(tmp)
