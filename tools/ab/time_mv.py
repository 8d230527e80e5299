"""run from a tree root: Path B at C4, us per iteration (split iteration), and the K1 roofline launch"""
import os, sys
sys.path.insert(0, os.getcwd())
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
res = []
for seed, mns in ((1, 1024), (4, 80)):
    m.chains_init(1024, seed=seed, max_new_segments=mns)
    out = []
    for k in range(8):
        m.chains_run(10)
        out.append(m.last_run_ms() / 10 * 1e3)
    res.append("%.1f" % (sum(out[3:]) / len(out[3:])))
print(os.path.basename(os.getcwd()), "path B us per iteration", " ".join(res))
