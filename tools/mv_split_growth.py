"""Path B at C4: the first half as one kernel against growth kernel + plane-fit kernel (option split_growth)."""
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
    for sg in (0, 1):
        m.set_option("split_growth", sg)
        m.chains_init(n, seed=4, max_new_segments=80)
        m.chains_run(4)
        ms = []
        for _ in range(3):
            m.chains_run(10); ms.append(m.last_run_ms())
        s = m.scalars()
        h = s["trace_hash"]
        ref.setdefault(n, (h, s["n_segments"].copy(), m.labels(1).copy()))
        same = np.array_equal(h, ref[n][0]) and np.array_equal(s["n_segments"], ref[n][1]) and np.array_equal(m.labels(1), ref[n][2])
        print("chains %5d split_growth %d: %.1f us per iteration, %.2f M proposals/s; status %s; same chains: %s"
              % (n, sg, np.mean(ms[1:]) * 100, n * 10 / np.mean(ms[1:]) / 1e3, np.unique(s["status"]).tolist(), same), flush=True)
