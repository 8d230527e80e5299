"""K1 on one 1920x1080 frame (C3): the table does not fit LDS, 256 bands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcmc_amd import host
from mcmc_amd.scene import make_scene
sc = make_scene(width=1920, height=1080, n_planes=24, n_superpixels=4000, seed=2)
ctx = host.Context(width=1920, height=1080)
ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
host.load_scene(ctx, 0, sc)
ctx.set_timing(True)
for _ in range(50): ctx.frames_moments()
ms = []
for _ in range(100):
    ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
print("C3 K1: mean %.1f us  min %.1f us" % (np.mean(ms) * 1e3, np.min(ms) * 1e3))
