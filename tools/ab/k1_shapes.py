"""K1 launch shapes on the roofline workload (256 frames of 640x480): options k1_bands x k1_threads, us per launch."""
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
ref = None
shapes = ((0, 0), (2, 480), (2, 960), (1, 480), (2, 640), (3, 320), (4, 240), (1, 800), (1, 640))
if os.environ.get("SHAPES"):      # e.g. SHAPES=1x960,1x800,1x640,1x960,1x800,1x640
    shapes = tuple(tuple(int(v) for v in x.split("x")) for x in os.environ["SHAPES"].split(","))
for bands, threads in shapes:
    ctx.set_option("k1_bands", bands); ctx.set_option("k1_threads", threads)
    for _ in range(4): ctx.frames_moments()
    ms = []
    for _ in range(20):
        ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
    m = ctx.moments(255)
    if ref is None: ref = m
    print("bands %d threads %4d: mean %.2f min %.2f us  %s" % (bands, threads, np.mean(ms) * 1e3, np.min(ms) * 1e3, "same moments" if np.array_equal(m, ref) else "MOMENTS DIFFER"), flush=True)
