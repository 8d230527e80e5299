"""K1 launch time at 640x480 (15 waves per workgroup: one SIMD holds three, the others four) and 512x480 (16 waves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcmc_amd import host
from mcmc_amd.scene import make_scene
for w, h in ((640, 480), (512, 480), (512, 600), (640, 480)):
    scs = [make_scene(width=w, height=h, n_planes=8, n_superpixels=int(500 * w * h / 307200), seed=s) for s in (1, 100, 101, 102)]
    smax = max(s.n_superpixels for s in scs)
    ctx = host.Context(width=w, height=h)
    ctx.frames_alloc(256, smax, 8)
    for f in range(256):
        sc = scs[f % 4]; ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
    ctx.set_timing(True)
    for _ in range(100): ctx.frames_moments()
    for dbg in (0, 1):
        ctx.set_option("k1_debug", dbg)
        for _ in range(20): ctx.frames_moments()
        ms = []
        for _ in range(50):
            ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
        us = np.mean(ms) * 1e3
        print("%dx%d S %d k1_debug %d: mean %.2f min %.2f us  %.3f ns per kilopixel" % (w, h, smax, dbg, us, np.min(ms) * 1e3, us * 1e3 / (256 * w * h / 1e3)), flush=True)
    del ctx
