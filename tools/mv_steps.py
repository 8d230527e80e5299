"""Path B at C4: time per iteration as the run goes on (split proposals add segments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
for seed, mns in ((1, 1024), (1, 200), (4, 80), (4, 1024)):
    m.chains_init(1024, seed=seed, max_new_segments=mns)
    out = []
    for k in range(8):
        m.chains_run(10)
        out.append(m.last_run_ms() / 10 * 1e3)
    print("seed", seed, "max_new", mns, "us per iteration, runs of 10:", " ".join("%.0f" % x for x in out))
