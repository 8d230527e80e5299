#!/bin/bash
# rocprofv3 PMC pass over tools/mv_prof.py (Path B split iteration). Usage: tools/prof_mv.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/mvpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/mv_prof.py > $OUT/pmc_sq.log 2>&1 || exit 1
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/pmc_sq/p_counter_collection.csv")):
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s %14.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
