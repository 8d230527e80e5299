#!/bin/bash
# rocprofv3 passes over tools/mv_prof.py (Path B split iteration): kernel trace, SQ counters, FETCH_SIZE, WRITE_SIZE,
# each pass a run of its own. Usage: tools/prof_mv.sh <tag>
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/mvprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/mv_prof.py"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $P > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -o p -- $P > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -o p -- $P > $OUT/pmc_sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -o p -- $P > $OUT/pmc_sq3.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $P > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $P > $OUT/pmc_write.log 2>&1 || exit 1
python3 - <<PY > $OUT/summary.txt
import csv, collections, glob
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob("$OUT/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %6s  avg %12.1f ns  total %14s ns  %6s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["TotalDurationNs"], r["Percentage"]))
print("== PMC counters, mean per launch ==")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in agg.items():
    if "copyBuffer" in k or "fillBuffer" in k: continue
    print(k)
    for c, v in sorted(dd.items()):
        print("   %-22s %16.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/summary.txt
