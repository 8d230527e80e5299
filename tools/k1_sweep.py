"""Tuning sweep of the moment kernel on the C5-shaped batch (not part of the product)."""
import sys, os, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene

F = int(os.environ.get("F", 256))
scs = [make_scene(seed=s) for s in (1, 100, 101, 102)]
W, H = 640, 480
ctx = host.Context()
ctx.frames_alloc(F, max(s.n_superpixels for s in scs), 8)
alg = 0
for f in range(F):
    sc = scs[f % 4]
    ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
    alg += W * H * 4 + sc.n_superpixels * 80
ctx.set_timing(True)
ref = None
combos = [tuple(map(int, c.split(":"))) for c in sys.argv[1:]] or [(8, 320, 1, 0)]
for combo in combos:
    ctx.set_option("k1_px", combo[5] if len(combo) > 5 else 0)
    bands, threads, lds = combo[:3]
    dbg = combo[3] if len(combo) > 3 else 0
    ctx.set_option("k1_debug", dbg)
    ctx.set_option("k1_bands", bands); ctx.set_option("k1_threads", threads); ctx.set_option("k1_use_lds", lds)
    for _ in range(3):
        ctx.frames_moments()
    ms = []
    for _ in range(20):
        ctx.frames_moments(); ms.append(ctx.last_kernel_ms(0))
    m = ctx.moments(F - 1)
    if ref is None:
        ref = m
    ok = np.array_equal(ref, m)
    print(f"bands={bands:3d} threads={threads:4d} lds={lds} dbg={dbg} avg={np.mean(ms)*1e3:8.1f}us min={np.min(ms)*1e3:8.1f}us "
          f"{alg/np.mean(ms)/1e6:7.0f} GB/s ok={ok}", flush=True)
