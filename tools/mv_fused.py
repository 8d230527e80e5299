"""Path B, fused kernel (the form used below 32 chains): time per proposal at C4 for a few chains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
m.vertex_moments()
for mode, name in ((0, "fused"), (1, "split")):
    m.set_option("split_step", mode)
    for n in (1, 8, 31, 64):
        m.chains_init(n, seed=1)
        m.chains_run(5)
        m.chains_run(20)
        print("%d chains, %s: %.1f us per iteration" % (n, name, m.last_run_ms() / 20 * 1e3))
