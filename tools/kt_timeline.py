"""usage: tools/kt_timeline.py <kernel_trace.csv>  -- GPU busy fraction (union of kernel intervals),
per-kernel totals, and the largest idle gaps, from a rocprofv3 --kernel-trace CSV."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady part: from the first to the last mesh_dp_kernel
dp = [x for x in iv if "mesh_dp" in x[2]]
t0, t1 = dp[0][0], dp[-1][1]
iv = [x for x in iv if x[1] > t0 and x[0] < t1]
busy = 0
cur_s, cur_e = None, None
gaps = []
for s, e, _ in iv:
    s, e = max(s, t0), min(e, t1)
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
print("window %.1f ms, busy %.1f ms (%.1f%%)" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
import collections
tot = collections.Counter()
for s, e, n in iv:
    tot[n.split("(")[0][-40:]] += e - s
for n, v in tot.most_common(8):
    print("  %-42s %8.1f ms" % (n, v / 1e6))
gaps.sort(reverse=True)
print("largest idle gaps (ms @ offset ms):", ", ".join("%.2f@%.0f" % (g / 1e6, o / 1e6) for g, o in gaps[:12]))
print("total idle in gaps > 0.2 ms: %.1f ms" % (sum(g for g, _ in gaps if g > 2e5) / 1e6))
