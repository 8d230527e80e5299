"""Path B at C4: iteration time against the number of chain groups (each group runs its iterations on its own stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
m.vertex_moments()
ref = {}
for n in [int(x) for x in os.environ.get("CHAINS", "1024 4096").split()]:
    for G in [int(x) for x in os.environ.get("MVG", "1 2 3 4").split()]:
        m.set_option("groups", G)
        m.chains_init(n, seed=4, max_new_segments=80)
        m.chains_run(4)
        ms = []
        for _ in range(3):
            m.chains_run(10); ms.append(m.last_run_ms())
        h = m.scalars()["trace_hash"]
        ref.setdefault(n, h)
        print("chains %5d groups %d: %.1f us per iteration, %.2f M proposals/s; same chains: %s"
              % (n, G, np.mean(ms[1:]) * 100, n * 10 / np.mean(ms[1:]) / 1e3, np.array_equal(h, ref[n])), flush=True)
