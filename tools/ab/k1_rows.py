"""K1 launch time against the frame height (fixed cost of a launch against the cost per row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcmc_amd import host
from mcmc_amd.scene import make_scene
for h in (480, 360, 240, 120):
    scs = [make_scene(width=640, height=h, n_planes=8, n_superpixels=500, seed=s) for s in (1, 100, 101, 102)]
    smax = max(s.n_superpixels for s in scs)
    ctx = host.Context(width=640, height=h)
    ctx.frames_alloc(256, smax, 8)
    for f in range(256):
        sc = scs[f % 4]; ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
    ctx.set_timing(True)
    for _ in range(100): ctx.frames_moments()
    for dbg in (0, 4, 1):
        ctx.set_option("k1_debug", dbg)
        for _ in range(20): ctx.frames_moments()
        ms = []
        for _ in range(50):
            ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
        print("H %d S %d k1_debug %d: mean %.2f min %.2f us" % (h, smax, dbg, np.mean(ms) * 1e3, np.min(ms) * 1e3), flush=True)
    del ctx
