"""run from a tree root: K1 on 256 frames of 640x480 (the roofline workload), us per launch"""
import os, sys
sys.path.insert(0, os.getcwd())
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
for _ in range(150): ctx.frames_moments()      # clock warm-up (k1_warm.py)
ms = []
for _ in range(50):
    ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
print(os.path.basename(os.getcwd()), "K1 us per 256-frame launch: mean %.2f min %.2f" % (np.mean(ms) * 1e3, np.min(ms) * 1e3))
ctx.set_option("k1_debug", 1)
sm = []
for _ in range(5): ctx.frames_moments()
for _ in range(20):
    ctx.frames_moments(); sm.append(ctx.last_kernel_ms(host.K_MOMENTS))
print(os.path.basename(os.getcwd()), "K1 loads only (k1_debug): mean %.2f min %.2f" % (np.mean(sm) * 1e3, np.min(sm) * 1e3))
