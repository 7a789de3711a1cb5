cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_BRANCH -d gpurun_out/pmc_c -o pmc -- python3 tools/perf_dp.py 1024 > gpurun_out/pmc_c.log 2>&1
python3 - <<'PY'
import csv,glob,collections
agg=collections.Counter();n=collections.Counter()
for f in glob.glob("gpurun_out/pmc_c/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "mesh_dp_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]);n[r["Counter_Name"]]+=1
for k in agg: print(k, agg[k]/n[k], n[k])
PY
grep "^abl" gpurun_out/pmc_c.log | tail -1
