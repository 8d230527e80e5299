#!/bin/bash
# two rocprofv3 --pmc passes over tools/mv_prof.py; prints the counters of the first-half kernel. Usage: tools/mv_pmc_quick.sh <tag>
set -u
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/mvpmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/mv_prof.py"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/p1 -o p -- $P > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/p2 -o p -- $P > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY
import csv, collections
for d in ("p1", "p2"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$OUT/%s/p_counter_collection.csv" % d)):
        if "mv_chain_run_kernel<1" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print("   %-22s %14.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
