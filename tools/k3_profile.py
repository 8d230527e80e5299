"""Per-phase cycle breakdown of the chain kernel (diagnostic build)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene
sc = make_scene(seed=1)
ctx = host.Context()
ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
host.load_scene(ctx, 0, sc)
ctx.frames_moments(); ctx.frames_edge_probs()
n = int(os.environ.get("CHAINS", 1024)); iters = int(os.environ.get("ITERS", 500))
ctx.chains_init(n, seed=1)
ctx.chains_run(0, 2000)            # reach steady state
ctx.set_timing(True)
ctx.chains_run(0, iters)
base_ms = ctx.last_kernel_ms(3)
ctx.set_option("k3_profile", 1)
ctx.chains_run(0, iters)
ms = ctx.last_kernel_ms(3)
pp = ctx.chain_profile().astype(np.float64) / (n * iters)
p = pp[:12]
print("mean component size %.1f  stream positions/step %.1f  refit members/step %.1f  bfs passes/step %.2f" % (pp[12], pp[13], pp[14], pp[15]))
names = ["pick", "bfs", "select", "products+sums", "reductions", "two fits", "ratios", "(unused)", "mh", "flush_stats",
         "apply", "bookkeeping"]
print(f"plain {base_ms*1e3/iters:.2f} us/step   profiled {ms*1e3/iters:.2f} us/step   cycles/step {p.sum():.0f}")
for k, v in zip(names, p):
    print(f"  {k:16s} {v:9.0f} cycles  {100*v/p.sum():5.1f}%")
s = ctx.scalars()
print("accept", s["n_accept"].sum() / (n * (2000 + 2 * iters)), "cases", s["n_case"].sum(axis=0) / s["n_case"].sum())
