#!/bin/bash
# usage: tools/prof_kmer_pmc.sh <tag> [nq]  -- SQ counter passes over the k-mer count kernel alone (tools/perf_kmer.py, 100 000 references);
# counters only, one small group per pass (never combined with trace domains other than kernel-trace).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; NQ=${2:-9216}
OUT=gpurun_out/pmck_$TAG
mkdir -p $OUT
i=0
GRPS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
      "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
      "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
      "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
      "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" \
      "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT")
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout 400 rocprofv3 --output-format csv --pmc $grp -d $OUT/p$i -o pmc -- python3 tools/perf_kmer.py $NQ 100000 > $OUT/run$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.Counter(); n = collections.Counter()
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kmer_count_kernel" not in r["Kernel_Name"]:
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
    out.append("LDS busy / wave-cycles       %.3f" % (g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES")))
open(d + "/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
