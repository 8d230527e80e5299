for n in 256 512 1024 2048 4096; do echo "chains $n"; CHAINS=$n REPEAT=2 timeout -k 10 120 python tools/mv_prof.py | grep "again" | tail -1 || exit 1; done
