"""usage: tools/kt_union.py <kernel_trace.csv>  -- per device-filling kernel of a bench run: launches, average
start-to-end duration (what `rocprofv3 --stats` prints), and the time during which at least one launch of that
kernel was resident (the union of its intervals).  Launches are chained (csrc/ctx.h, heavy_launch): a kernel
starts when the launch before it has dispatched its last workgroup, so consecutive launches overlap while the
older one drains; the union counts that shared time once.  Also: the time during which ANY of them was resident
and how much of the window between the first and the last that is."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
HEAVY = ("mesh_dp_", "family_graph_kernel", "kmer_count_kernel", "kmer_select_kernel")


def short(n):
    for k in HEAVY:
        if k in n:
            return k
    return None


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    return tot + (cur_e - cur_s if cur_e is not None else 0)


iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows
            if short(r["Kernel_Name"]))
dp = [x for x in iv if x[2] == "mesh_dp_"]
# the timed region: the longest stretch of the run in which no two consecutive device-filling kernels are more than
# 12 ms apart (between bench.py's set-up pass, warm-up, timed steps and isolated step the host counts results for
# hundreds of milliseconds); the stretch's first and last DP launch bound the window
runs, cur = [], [iv[0]]
reach = iv[0][1]
for x in iv[1:]:
    if x[0] - reach > 12e6:
        runs.append(cur)
        cur = []
    cur.append(x)
    reach = max(reach, x[1])
runs.append(cur)
best = max(runs, key=lambda r: sum(1 for x in r if x[2] == "mesh_dp_"))
bdp = [x for x in best if x[2] == "mesh_dp_"]
t0, t1 = bdp[1][0], bdp[-1][1]   # (from the second DP launch of the stretch: the pipeline is full by then)
iv = [x for x in iv if x[0] >= t0 and x[1] <= t1]
print("window: the run's longest gap-free stretch, DP launch 2 .. %d of its %d (%d in the trace), %.1f ms" % (
    len(bdp), len(bdp), len(dp), (t1 - t0) / 1e6))
for k in HEAVY:
    mine = [(s, e) for s, e, n in iv if n == k]
    if not mine:
        continue
    tot = sum(e - s for s, e in mine)
    u = union(mine)
    print("  %-22s %4d launches  avg start-to-end %7.2f ms  resident (union) %8.1f ms = %7.2f ms per launch  overlap with itself %5.1f %%" % (
        k, len(mine), tot / len(mine) / 1e6, u / 1e6, u / len(mine) / 1e6, 100.0 * (tot - u) / tot))
allu = union([(s, e) for s, e, _ in iv])
print("  any of them resident %.1f ms = %.1f %% of the window; sum of their durations %.1f ms (%.1f %% counted twice)" % (
    allu / 1e6, 100.0 * allu / (t1 - t0), sum(e - s for s, e, _ in iv) / 1e6,
    100.0 * (sum(e - s for s, e, _ in iv) - allu) / max(1, allu)))
