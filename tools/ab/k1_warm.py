"""K1 on the roofline workload: launch time as the process goes on (clock ramp), mean of every 25 launches."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from mcmc_amd import host
from mcmc_amd.scene import make_scene
scs = [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=s) for s in (1, 100, 101, 102)]
ctx = host.Context(width=640, height=480)
ctx.frames_alloc(256, max(s.n_superpixels for s in scs), 8)
for f in range(256):
    sc = scs[f % 4]; ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
ctx.set_timing(True)
for blk in range(16):
    ms = []
    for _ in range(25):
        ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
    print("launches %3d..%3d: mean %.2f min %.2f us" % (blk * 25, blk * 25 + 24, np.mean(ms) * 1e3, np.min(ms) * 1e3), flush=True)
    if blk == 7: time.sleep(2.0); print("(slept 2 s)")
