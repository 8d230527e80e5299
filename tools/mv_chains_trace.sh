#!/bin/bash
# first-half kernel time of the split iteration against the number of chains (1 = alone, 256 = one per CU, 1024 = four per CU)
set -u
cd /tmp && export TMPDIR=/tmp
for c in 1 64 256 512 768 1024 2048; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/mvchains_$c
  mkdir -p $OUT
  CHAINS=$c timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/mv_prof.py > $OUT/trace.log 2>&1 || exit 1
  python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")):
    if "mv_chain_run_kernel<1" in r["Name"] or "mv_errors" in r["Name"] or "mv_chain_run_kernel<2" in r["Name"]:
        print("chains %5d  %-40s avg %10.1f us" % ($c, r["Name"][:40], float(r["AverageNs"]) / 1e3))
PY
done
