"""K1 on the roofline workload with 2, 3 (default), 4 and 5 rows in flight per wave (tuning option k1_debug)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcmc_amd import host
from mcmc_amd.scene import make_scene
scs = [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=s) for s in (1, 100, 101, 102)]
smax = max(s.n_superpixels for s in scs)
ctx = host.Context(width=640, height=480)
ctx.frames_alloc(256, smax, 8)
for f in range(256):
    sc = scs[f % 4]; ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
ctx.set_timing(True)
for _ in range(150): ctx.frames_moments()
ref = ctx.moments(5)
for dbg in (0, 2, 4, 5, 0, 1):
    ctx.set_option("k1_debug", dbg)
    for _ in range(20): ctx.frames_moments()
    ms = []
    for _ in range(50):
        ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
    ok = dbg == 1 or np.array_equal(ctx.moments(5), ref)
    print("k1_debug %d: mean %.2f min %.2f us   same moments: %s" % (dbg, np.mean(ms) * 1e3, np.min(ms) * 1e3, ok), flush=True)
