#!/bin/bash
# rocprofv3 kernel trace of tools/mv_prof.py only (per-kernel times of the split iteration). Usage: tools/mv_trace.sh <tag>
set -u
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/mvtrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/mv_prof.py > $OUT/trace.log 2>&1 || exit 1
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")):
    print("%-60s calls %6s  avg %12.1f ns  %6s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
