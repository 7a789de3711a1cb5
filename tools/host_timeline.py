"""Start-up and tail of a bench run from a SINA_HOST_TRACE file: what the host threads do between the start of the
timed region and the first GPU call, and between the last GPU call's return and the end.
   SINA_HOST_TRACE=gpurun_out/trace.txt python bench.py --no-cpu-baseline ...; python tools/host_timeline.py gpurun_out/trace.txt"""
import sys
ev = []
for ln in open(sys.argv[1]):
    p = ln.split()
    tid, name, t0, t1 = p[0], " ".join(p[1:-2]), float(p[-2]), float(p[-1])
    ev.append((t0, t1, tid, name))
ev.sort()
starts = [e for e in ev if e[3] == "MARK:timed-start"]
ends = [e for e in ev if e[3] == "MARK:timed-end"]
if not starts or not ends:
    sys.exit("no timed-region marks in the trace")
ts, te = starts[-1][0], ends[-1][0]
inside = [e for e in ev if ts <= e[0] <= te and not e[3].startswith("MARK:timed")]
for e in inside:
    if e[3].startswith("MARK:"):
        print("  +%7.2f ms  %s" % (1e3 * (e[0] - ts), e[3]))
print("timed region %.1f ms" % (1e3 * (te - ts)))
gpu = [e for e in inside if "C-ABI" in e[3]]
first_gpu, last_gpu_end = min(e[0] for e in gpu), max(e[1] for e in gpu)
print("first GPU call %.1f ms after the start; last GPU call returns %.1f ms before the end" % (
    1e3 * (first_gpu - ts), 1e3 * (te - last_gpu_end)))
print("--- until the first GPU call:")
for t0, t1, tid, name in inside:
    if t0 <= first_gpu:
        print("  +%7.2f ms  %-28s %7.2f ms  (thread %s)" % (1e3 * (t0 - ts), name, 1e3 * (t1 - t0), tid))
print("--- after the last GPU call returned:")
for t0, t1, tid, name in inside:
    if t1 >= last_gpu_end:
        print("  %7.2f ms before the end  %-28s %7.2f ms  (thread %s)" % (1e3 * (te - t0), name, 1e3 * (t1 - t0), tid))
