"""usage: tools/bt_delay.py <kernel_trace.csv>  -- how long after the end of a DP launch its backtrack walk starts, what the walk
and the assembly take, and how much of each DP launch they run beside (steady part of a bench run)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
H = ("mesh_dp_", "family_graph_kernel", "kmer_count_kernel", "kmer_select_kernel", "backtrack_kernel", "assemble_kernel")
def short(n):
    for k in H:
        if k in n:
            return k
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows if short(r["Kernel_Name"]))
dps = [x for x in iv if x[2] == "mesh_dp_"]
bts = [x for x in iv if x[2] == "backtrack_kernel"]
lo, hi = dps[len(dps) // 3][0], dps[-2][1]
delays, durs, ov = [], [], []
for b in bts:
    if not (lo <= b[0] <= hi):
        continue
    ended = [d for d in dps if d[1] <= b[0]]
    if ended:
        delays.append((b[0] - max(d[1] for d in ended)) / 1e6)
    durs.append((b[1] - b[0]) / 1e6)
for d in dps:
    if lo <= d[0] and d[1] <= hi:
        ov.append(sum(max(0, min(e, d[1]) - max(s, d[0])) for s, e, k in iv if k in ("backtrack_kernel", "assemble_kernel")) / 1e6)
delays.sort()
print("backtrack starts after its DP launch's end: median %.2f ms, max %.2f ms, > 2 ms in %d of %d; walk lasts %.2f ms on average (max %.2f); "
      "walk + assembly run beside a DP launch for %.2f ms per launch; DP start-to-end %.2f ms" % (
          delays[len(delays) // 2], delays[-1], sum(1 for x in delays if x > 2), len(delays), sum(durs) / len(durs), max(durs),
          sum(ov) / max(1, len(ov)), sum((d[1] - d[0]) / 1e6 for d in dps if lo <= d[0] and d[1] <= hi) / max(1, len(ov))))
