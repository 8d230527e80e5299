"""K3 per-step time of the plain and the two-wave form, at C2 (synthetic scene) and config 1 (the bundled frame)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene


def run(ctx, n, spec, iters=2000):
    ctx.set_option("k3_spec", spec)
    ctx.chains_init(n, seed=1)
    ctx.chains_run(1000, 0)
    ctx.set_timing(True)
    ctx.chains_run(0, iters)
    ms = ctx.last_kernel_ms(3)
    s = ctx.scalars()
    return ms * 1e3 / iters, ctx.last_variant() if hasattr(ctx, "last_variant") else -1, s["n_accept"].sum() / (n * (1000 + iters))


sc = make_scene(seed=1)
ctx = host.Context()
ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
host.load_scene(ctx, 0, sc)
ctx.frames_moments(); ctx.frames_edge_probs()
for spec in (-1, 1, 0):
    print("C2  k3_spec %2d: %.2f us per step, variant %s, accept %.3f" % ((spec,) + run(ctx, 1024, spec)))
ctx.close()
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_frame.npz"))
o = host.OverSegmenter(640, 480)
S, e2 = o.run(z["colors"], z["disp"])
ctx = host.Context(width=640, height=480)
ctx.frames_alloc(1, S, e2)
d_disp, d_label = o.device_buffers()
ctx.frame_upload_device(0, d_disp, d_label, S)
rp, cl = o.graph()
ctx.frame_set_graph(0, rp, cl)
ctx.frames_moments(); ctx.frames_edge_probs()
for spec in (-1, 1, 0):
    print("C1  k3_spec %2d: %.2f us per step, variant %s, accept %.3f" % ((spec,) + run(ctx, 1024, spec)))
