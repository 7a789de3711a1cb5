import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from sina_amd import synth, capi
from tests import util
nq=int(sys.argv[1]) if len(sys.argv)>1 else 256
refs=synth.make_refs(2000, length=1500, width=50000, seed=2)
qs=synth.make_queries(refs, nq, seed=3)
cs=util.cseqs_from_refs(refs)
idx=po.Index(cs,k=10)
t=time.time()
graphs=[];qms=[]
for qi in range(nq):
    q=util.query_cseq(qs,qi)
    ids,sc,_=idx.famfinder(q)
    graphs.append(util.graph_dict([cs[i] for i in ids]))
    qms.append((q.packed()>>24).astype(np.uint8))
print("prep",time.time()-t)
qoff=np.zeros(nq+1,np.uint64); qoff[1:]=np.cumsum([len(m) for m in qms])
ctx=capi.Context(0)
gb=ctx.graph_batch(graphs, refs.width)
qm=np.concatenate(qms)
for lds in (os.environ.get('SINA_HIP_DP_LDS_KB','128'),):
    for rep in range(3):
        t=time.time(); out,pos=ctx.align_graphs(gb,qm,qoff); dt=time.time()-t
        st=ctx.stats()
        print("lds",lds,"wall %.3fs dp %.2f ms bt %.2f ms cells %.3g  -> %.1f Gcell/s  %.1f GB/s(8B/cell)  q/s(dp) %.0f"%(dt,st['dp_ms'],st['backtrack_ms'],st['dp_cells'],st['dp_cells']/st['dp_ms']/1e6, 8*st['dp_cells']/st['dp_ms']/1e6, nq/st['dp_ms']*1e3))
