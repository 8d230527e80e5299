for n in 2048 4096 8192 16384 32768 65536; do echo "ewaves $n"; MCMC_MV_EWAVES=$n timeout -k 10 120 python tools/mv_prof.py || exit 1; done
