"""FASTA reader -> writer alone (no GPU): sequences/s and MB/s at alignment width 50 000.
usage: tools/perf_fasta.py [n_sequences]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sina_amd import synth, pipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
refs = synth.make_refs(n, length=1500, width=50000, seed=2)
path, out = "/tmp/perf_fasta_in.fasta", "/tmp/perf_fasta_out.fasta"
with open(path, "w") as f:
    for i in range(refs.n):
        f.write(">seq%d\n%s\n" % (i, synth.aligned_string(refs.seq(i), refs.width)))
for rep in range(3):
    t = time.time()
    got, sk = pipeline.fasta_roundtrip(path, out)
    dt = time.time() - t
    print("read + write: %d sequences in %.2f s -> %.0f seq/s, %.0f MB/s in, %.0f MB/s out" % (
        got, dt, got / dt, os.path.getsize(path) / 1e6 / dt, os.path.getsize(out) / 1e6 / dt))
os.remove(path); os.remove(out)
