"""Same-box A/B timing of library builds.  Fresh GPU boxes differ by a per cent or two, so kernels are compared inside
one gpurun call: every variant is a directory tools/ab/trees/<name>/ holding a copy of mcmc_amd/ (with its own
libmcmc_amd.so) and include/, "work" is the working tree; the variants are timed in turn, twice over.
  python tools/ab/time_multi.py base work          (TIME_SCRIPT=time_k3.py | time_k1.py | time_mv.py, default time_k3.py)
A baseline tree of the last commit:  mkdir -p tools/ab/trees/base && git archive HEAD mcmc_amd include | tar -x -C
tools/ab/trees/base && cp mcmc_amd/libmcmc_amd.so tools/ab/trees/base/mcmc_amd/   (before rebuilding the working tree)."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
names = sys.argv[1:]
for rnd in range(2):
    for name in names:
        d = root if name == "work" else os.path.join(root, "tools/ab/trees", name)
        r = subprocess.run([sys.executable, os.path.join(root, "tools/ab/" + os.environ.get("TIME_SCRIPT", "time_k3.py"))], cwd=d, capture_output=True, text=True, timeout=200)
        for line in r.stdout.strip().splitlines(): print("%-10s %s" % (name, line.split(" ", 1)[1]), flush=True)
        if r.returncode: print(name, "FAILED", r.stderr[-300:], flush=True)
