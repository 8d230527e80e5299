"""C5 share (32 frames x 64 chains at 640x480) with the chain kernel's forms: dense (default there), two-wave, plain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene
seeds = [int(x) for x in os.environ.get("C5_SEEDS", "1,100,101,102").split(",")]       # the bench's four scenes
scs = [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=s) for s in seeds]
smax = max(s.n_superpixels for s in scs); emax = max(s.col.size for s in scs)
ctx = host.Context()
ctx.frames_alloc(32, smax, emax)
for f in range(32):
    host.load_scene(ctx, f, scs[f % 4])
ctx.frames_moments(); ctx.frames_edge_probs()
ctx.set_timing(True)
for spec, dense in ((0, 0), (1, -1), (3, -1), (2, -1), (-1, 0), (-1, -1)):
    ctx.set_option("k3_spec", spec); ctx.set_option("k3_dense", dense)
    ctx.chains_init(2048, chains_per_frame=64, seed=5)
    ctx.chains_run(500, 0)
    ctx.chains_run(0, 1000)
    ms = ctx.last_kernel_ms(3)
    print("k3_spec %2d k3_dense %2d: variant %d, %.2f us per step, %.1f M proposals/s" % (spec, dense, ctx.last_variant(), ms, 2048 * 1000 / ms / 1e3))

ctx.set_option("k3_spec", 0); ctx.set_option("k3_dense", 0)
ctx.chains_init(2048, chains_per_frame=64, seed=5)
for k in range(5):
    ctx.chains_run(0, 1000)
    print("measured launch %d: %.2f us per step, visit capacity %d" % (k, ctx.last_kernel_ms(3), ctx.visit_capacity()))

# the bench's own sequence: moments + edge probabilities before every launch, no burn-in, no sync in between
ctx.chains_init(2048, chains_per_frame=64, seed=5)
for k in range(4):
    ctx.frames_moments(); ctx.frames_edge_probs(); ctx.chains_run(0, 1000, sync=False)
    ctx.synchronize()
    print("bench sequence launch %d: %.2f us per step" % (k, ctx.last_kernel_ms(3)))
