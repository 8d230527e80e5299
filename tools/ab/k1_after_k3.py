"""K1 launch times right after a few seconds of the chain kernel (the order bench.py runs them in) against a fresh start."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcmc_amd import host
from mcmc_amd.scene import make_scene
scs = [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=s) for s in (1, 100, 101, 102)]
smax = max(s.n_superpixels for s in scs)
def k1_blocks(tag):
    ctx = host.Context(width=640, height=480)
    ctx.frames_alloc(256, smax, 8)
    for f in range(256):
        sc = scs[f % 4]; ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
    ctx.set_timing(True)
    out = []
    for blk in range(10):
        ms = []
        for _ in range(100):
            ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
        out.append("%.1f" % (np.mean(ms) * 1e3))
    print(tag, "K1 us, blocks of 100 launches:", " ".join(out), flush=True)
    ctx.close()
k1_blocks("fresh process:")
sc = scs[0]
c = host.Context()
c.frames_alloc(1, sc.n_superpixels, sc.col.size)
host.load_scene(c, 0, sc)
c.frames_moments(); c.frames_edge_probs()
c.chains_init(1024, seed=1)
c.reserve_visits(20000)
t0 = time.time()
while time.time() - t0 < 8.0:
    c.chains_run(0, 1000); c.synchronize()
c.close()
k1_blocks("after 8 s of the chain kernel:")
time.sleep(3.0)
k1_blocks("after 3 s idle:")
