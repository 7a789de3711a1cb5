"""usage: tools/host_trace.py <trace file>  -- what the batches in flight were doing (SINA_HOST_TRACE)."""
import sys, collections
ev = []
for l in open(sys.argv[1]):
    tid, name, t0, t1 = l.split()
    ev.append((int(tid), name, float(t0), float(t1)))
top = [e for e in ev if e[1].startswith(("drv.", "ff.find_batch", "ff.match", "ff.post", "al."))]
# steady window: middle 60 % of the trace
lo = min(e[2] for e in top); hi = max(e[3] for e in top)
a, b = lo + 0.3 * (hi - lo), lo + 0.9 * (hi - lo)
tot = collections.Counter()
for tid, name, t0, t1 in top:
    s, e = max(t0, a), min(t1, b)
    if e > s:
        tot[name] += e - s
threads = sorted({e[0] for e in top})
print("window %.3f s, %d worker threads" % (b - a, len(threads)))
for n, v in tot.most_common():
    print("%-28s %7.3f s  = %.2f workers busy" % (n, v, v / (b - a)))
# fraction of time NO worker is inside a GPU call
gpu = sorted((max(t0, a), min(t1, b)) for tid, n, t0, t1 in ev if "C-ABI" in n and min(t1, b) > max(t0, a))
cur = None; busy = 0
for s, e in gpu:
    if cur is None: cur = [s, e]
    elif s <= cur[1]: cur[1] = max(cur[1], e)
    else: busy += cur[1] - cur[0]; cur = [s, e]
if cur: busy += cur[1] - cur[0]
print("some worker inside a GPU call: %.1f%% of the window" % (100 * busy / (b - a)))
