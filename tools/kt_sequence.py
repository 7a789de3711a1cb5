"""usage: tools/kt_sequence.py <kernel_trace.csv> [first_ms] [last_ms]  -- the pipeline's kernels in start order
(name, start, end, duration in ms from the first DP launch), to see what runs beside and behind what."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
dp = [x for x in iv if "mesh_dp" in x[2]]
t0 = dp[len(dp) // 2][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 120.0
short = {"mesh_dp": "DP", "family_graph": "graph", "kmer_count": "kmer", "chain_scout": "scout", "backtrack": "walk", "assemble": "asm"}
for s, e, n in iv:
    a, b = (s - t0) / 1e6, (e - t0) / 1e6
    if b < lo or a > hi:
        continue
    tag = next((v for k, v in short.items() if k in n), None)
    if tag:
        print("%-6s %8.2f .. %8.2f  (%6.2f ms)" % (tag, a, b, b - a))
