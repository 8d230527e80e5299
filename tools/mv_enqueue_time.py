"""Path B at C4: host time to enqueue a run of 20 iterations against the device time of the run (is a run launch bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
m.vertex_moments()
for n in (1, 64, 256, 1024):
    m.chains_init(n, seed=1)
    m.chains_run(4)
    best_e = best_g = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); m.chains_run(20, sync=False); t1 = time.perf_counter(); m.chains_run(20)   # the second run waits for both
        best_e = min(best_e, t1 - t0); best_g = min(best_g, m.last_run_ms())
    print(f"{n} chains: host enqueue of 20 iterations {best_e*1e3:.3f} ms ({best_e*1e6/20:.1f} us per iteration), device {best_g:.3f} ms ({best_g*50:.1f} us per iteration)")
