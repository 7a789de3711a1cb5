"""usage: tools/heavy_gaps.py <kernel_trace.csv>  -- how busy the device-filling kernels (k-mer count /
select, DAG build, DP: the store's FIFO stream) keep the GPU between the first and the last DP launch
of a bench run, and what sits in front of the idle gaps."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
HEAVY = ("mesh_dp_", "family_graph_kernel", "kmer_count_kernel", "kmer_select_kernel")
def short(n):
    for k in HEAVY + ("backtrack_kernel", "copyBuffer", "fillBuffer"):
        if k in n:
            return k
    return n[:24]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
hv = [x for x in iv if x[2] in HEAVY]
# the timed region: the run's longest stretch without a gap of more than 12 ms between device-filling kernels
# (tools/kt_union.py picks the same window), from its second DP launch to its last
runs, cur, reach = [], [hv[0]], hv[0][1]
for x in hv[1:]:
    if x[0] - reach > 12e6:
        runs.append(cur)
        cur = []
    cur.append(x)
    reach = max(reach, x[1])
runs.append(cur)
best = max(runs, key=lambda r: sum(1 for x in r if x[2] == "mesh_dp_"))
dp = [x for x in best if x[2] == "mesh_dp_"]
t0, t1 = dp[1][0], dp[-1][1]
hv = [x for x in hv if x[0] >= t0 and x[1] <= t1]
busy, reach = 0, None   # (union: chained kernels overlap)
for s0, e0, _ in sorted(hv):
    if reach is None or s0 > reach:
        busy += e0 - s0
        reach = e0
    elif e0 > reach:
        busy += e0 - reach
        reach = e0
print("window %.1f ms: heavy kernels busy %.1f ms (%.1f %%)" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
import collections
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in hv:
    tot[n] += e - s; cnt[n] += 1
for n in HEAVY:
    if cnt[n]:
        print("  %-22s %4d launches  avg %7.2f ms  total %8.1f ms (%.1f %%)" % (n, cnt[n], tot[n] / cnt[n] / 1e6, tot[n] / 1e6, 100.0 * tot[n] / (t1 - t0)))
gaps = collections.Counter(); gcnt = collections.Counter()
reach = None
for (s0, e0, n0), (s1, e1, n1) in zip(hv, hv[1:]):
    reach = e0 if reach is None else max(reach, e0)
    g = s1 - reach
    if g > 20000:
        gaps[n0 + " -> " + n1] += g; gcnt[n0 + " -> " + n1] += 1
print("idle between heavy kernels (> 20 us), by neighbours:")
for k, v in gaps.most_common(10):
    print("  %-50s %3d x  avg %6.2f ms  total %7.1f ms" % (k, gcnt[k], v / gcnt[k] / 1e6, v / 1e6))
