"""K1 on the C5-shaped batch, a few launches: the program rocprofv3 wraps (kernel trace / PMC passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene

F = int(os.environ.get("F", 256)); REPS = int(os.environ.get("REPS", 5))
scs = [make_scene(seed=s) for s in (1, 100, 101, 102)]
ctx = host.Context()
ctx.frames_alloc(F, max(s.n_superpixels for s in scs), 8)
alg = 0
for f in range(F):
    sc = scs[f % 4]
    ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
    alg += 640 * 480 * 4 + sc.n_superpixels * 80
for k, v in (("k1_bands", "BANDS"), ("k1_threads", "THREADS"), ("k1_debug", "DEBUG")):
    if v in os.environ:
        ctx.set_option(k, int(os.environ[v]))
ctx.set_timing(True)
ms = []
for _ in range(REPS):
    ctx.frames_moments(); ms.append(ctx.last_kernel_ms(0))
print("algorithmic bytes per launch", alg, "ms", ms)
ctx.close()
