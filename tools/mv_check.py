"""Diagnostic build -DMC_MV_ERR_CHECK: the shapes test's case with the out-of-range indices of the L1 kernels counted, not used."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
m.vertex_moments()
m.set_option("split_step", 1)
lib = host.load()
for n in (37, 1030):
    m.chains_init(n, seed=3, max_new_segments=8, split_prob=0.2)
    m.chains_set_trace(6)
    m.chains_run(6)
    buf = np.zeros(16, dtype=np.uint64)
    assert lib.mcmc_mv_debug_check(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    print(n, "violations", buf[:8].tolist(), "last values", [hex(int(v)) for v in buf[8:]])
