"""--insertion=forbid through the pipeline under a forced DP geometry (SINA_HIP_TEST=geom=T,B): crash hunt.
usage: SINA_HIP_TEST=geom=T,B tools/dbg_forbid.py [n_queries] [window]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sina_amd import pipeline, synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 24
win = int(sys.argv[2]) if len(sys.argv) > 2 else 120
refs = synth.make_refs(500, length=320, width=3200, seed=51, amb_rate=0.01, lower_rate=0.02)
st = pipeline.Store(":mem:dbg", refs)
st.build_index(10, False)
qs = synth.make_queries(refs, nq, seed=53, window=(0.3, win) if win else None, ins=0.02, dele=0.02, lower_rate=0.05)
pl = pipeline.Pipeline(st, famfinder={"fs-min-len": 100, "fs-full-len": 250}, aligner={"insertion": os.environ.get("DBG_INSERTION", "forbid")})
q0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
q1 = int(sys.argv[4]) if len(sys.argv) > 4 else nq
import numpy as np
off = (qs.off[q0:q1 + 1] - qs.off[q0]).astype(np.uint64)
if os.environ.get("DBG_ABL"):
    from sina_amd import capi
    capi.load().sina_hip_debug_dp_ablate(int(os.environ["DBG_ABL"]))
pl.run(qs.mask[qs.off[q0]:qs.off[q1]], off, batch=q1 - q0, inflight=1)
print("ok", os.environ.get("SINA_HIP_TEST"), q0, q1, sum(1 for q in range(q1 - q0) if pl.result(q)["status"] == 0), "aligned")
pl.close(); st.close()
