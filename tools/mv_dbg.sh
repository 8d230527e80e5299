cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do echo "dbg $m"; MCMC_DBG=$m timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg$m -o t -- python3 $GRAFT_REPO_ROOT/tools/mv_prof.py > /dev/null 2>&1 || exit 1; grep "errors_kernel" $GRAFT_REPO_ROOT/gpurun_out/dbg$m/t_kernel_stats.csv; done
