"""Path B at C4: time of the split iteration and, with rocprofv3 off, nothing else; prints ms per 10 steps (best of REPEAT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
m.vertex_moments()
n = int(os.environ.get("CHAINS", 1024))
for opt in ("split_growth", "fuse_halves", "mvg", "growth_ahead", "growth_ahead_waves"):
    if os.environ.get((opt.upper())) is not None:
        m.set_option("groups" if opt == "mvg" else opt, int(os.environ[(opt.upper())]))
m.chains_init(n, seed=1)
m.chains_run(4)
best = 1e9
for _ in range(int(os.environ.get("REPEAT", 6))):
    m.chains_run(20)
    best = min(best, m.last_run_ms())
print("%s %s: %d chains, 20 iterations best %.3f ms -> %.1f us per iteration, %.2f M proposals/s" % (os.environ.get("MCMC_AMD_LIB", "product").split("/")[-1], {k: os.environ[k] for k in ("SPLIT_GROWTH", "FUSE_HALVES", "MVG", "GROWTH_AHEAD", "GROWTH_AHEAD_WAVES") if k in os.environ}, n, best, best * 50, n * 20 / best / 1e3))
