"""K1 on the roofline workload: the library named by MCMC_AMD_LIB (or the product), mean / min of 3 x 50 launches."""
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
out = []
for rep in range(3):
    ms = []
    for _ in range(50):
        ctx.frames_moments(); ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
    out.append("%.2f/%.2f" % (np.mean(ms) * 1e3, np.min(ms) * 1e3))
print(os.environ.get("MCMC_AMD_LIB", "product").split("/")[-1], " mean/min us:", "  ".join(out), " checksum", int(ctx.moments(7).sum()) & 0xFFFFFFFF)
