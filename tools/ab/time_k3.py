"""run from a tree root: C2 one-wave and the C5 share (bench scenes), chain kernel us per step"""
import os, sys
sys.path.insert(0, os.getcwd())
from mcmc_amd import host
from mcmc_amd.scene import make_scene
seeds = [1, 100, 101, 102]
scs = [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=s) for s in seeds]
smax = max(s.n_superpixels for s in scs); emax = max(s.col.size for s in scs)
def opt(ctx, k, v):
    try: ctx.set_option(k, v)
    except Exception as e: pass
ctx = host.Context()
ctx.frames_alloc(32, smax, emax)
for f in range(32): host.load_scene(ctx, f, scs[f % 4])
ctx.frames_moments(); ctx.frames_edge_probs(); ctx.set_timing(True)
opt(ctx, "k3_spec", 0)
ctx.chains_init(2048, chains_per_frame=64, seed=5)
r = []
for k in range(4):
    ctx.chains_run(0, 1000); r.append(ctx.last_kernel_ms(3))
print(os.path.basename(os.getcwd()), "c5 share, variant", ctx.last_variant(), " ".join("%.2f" % x for x in r))
ctx.close()
ctx = host.Context()
ctx.frames_alloc(1, scs[0].n_superpixels, scs[0].col.size)
host.load_scene(ctx, 0, scs[0]); ctx.frames_moments(); ctx.frames_edge_probs(); ctx.set_timing(True)
opt(ctx, "k3_spec", 0)
ctx.chains_init(1024, chains_per_frame=1024, seed=1)
r = []
for k in range(4):
    ctx.chains_run(0, 2000); r.append(ctx.last_kernel_ms(3) / 2)
print(os.path.basename(os.getcwd()), "c2, 1024 chains, variant", ctx.last_variant(), " ".join("%.2f" % x for x in r))
