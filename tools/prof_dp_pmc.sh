#!/bin/bash
# usage: tools/prof_dp_pmc.sh <tag> [nq]  -- SQ counter passes over the DP kernel alone (tools/perf_dp.py);
# counters only, one small group per pass (never combined with trace domains other than kernel-trace).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; NQ=${2:-512}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
# (QUICK=1: the two groups that decide what bounds the kernel)
GRPS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC")
[ -n "$QUICK" ] && GRPS=("${GRPS[@]:0:2}")
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout 400 rocprofv3 --output-format csv --pmc $grp -d $OUT/p$i -o pmc -- python3 tools/perf_dp.py $NQ > $OUT/run$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.Counter(); n = collections.Counter()
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mesh_dp_" not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
out = []
for k in sorted(agg):
    out.append("%-24s %16.0f per launch (%d launches)" % (k, agg[k] / n[k], n[k]))
g = lambda k: agg[k] / max(1, n[k])
if g("SQ_WAVE_CYCLES"):
    out.append("")
    out.append("VALU busy / wave-cycles      %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
    out.append("any-inst busy / wave-cycles  %.3f" % (g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
    out.append("wait-inst-any / wave-cycles  %.3f" % (g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
    out.append("wait-any / wave-cycles       %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
    out.append("LDS wait / wave-cycles       %.3f" % (g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")))
    out.append("VALU insts : SALU : LDS : SMEM : branch : VMEM = %.0f : %.0f : %.0f : %.0f : %.0f : %.0f" % (
        g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_SMEM"), g("SQ_INSTS_BRANCH"),
        g("SQ_INSTS_VMEM_WR") + g("SQ_INSTS_VMEM_RD")))
print("\n".join(out))
open(d + "/summary.txt", "w").write("\n".join(out) + "\n")
# VALU wave-instructions per mesh cell, for bench.py's roofline_valu (cells per launch: the run's own log)
import json, os, re
sys.path.insert(0, os.getcwd())
try:
    import bench
    cells = None
    swept = 1.0  # (the certified row skip: fraction of the nominal cells the kernel computed, second sweeps included)
    for f in sorted(glob.glob(d + "/run*.log")):
        for l in open(f):
            m = re.search(r" cells ([0-9.e+]+) ", l)
            if m:
                cells = float(m.group(1))
            m = re.search(r"cells swept ([0-9.]+)", l)
            if m:
                swept = float(m.group(1))
    if cells:
        cells *= swept
    if cells and g("SQ_INSTS_VALU"):
        json.dump({"kernel_source_rev": bench.kernel_source_rev(), "valu_wave_instructions_per_launch": g("SQ_INSTS_VALU"),
                   "cells_per_launch": cells, "cells_computed_frac": swept, "valu_wave_instructions_per_cell": g("SQ_INSTS_VALU") / cells,
                   "what": "tools/prof_dp_pmc.sh: SQ_INSTS_VALU of the DP kernel alone per COMPUTED cell, tools/perf_dp.py workload"},
                  open(d + "/dp_valu.json", "w"), indent=1)
except Exception as e:
    print("no dp_valu.json:", e)
PY
