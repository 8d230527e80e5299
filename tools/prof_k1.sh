#!/bin/bash
# rocprofv3 passes over tools/k1_prof.py: kernel trace + PMC (each pass its own run). Usage: tools/prof_k1.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/k1prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/k1_prof.py > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/k1_prof.py > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/pmc_sq2 -o p -- python3 $GRAFT_REPO_ROOT/tools/k1_prof.py > $OUT/pmc_sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $GRAFT_REPO_ROOT/tools/k1_prof.py > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $GRAFT_REPO_ROOT/tools/k1_prof.py > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -30
