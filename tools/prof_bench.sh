#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command itself. Usage: tools/prof_bench.sh <tag>
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/benchprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $OUT/bench.log 2> $OUT/bench.err || exit 1
