"""K3 at C2 (1024 chains, 594 superpixels): time per step of the chain kernel's forms (tuning aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene
sc = make_scene(seed=1)
n = int(os.environ.get("CHAINS", 1024)); iters = int(os.environ.get("ITERS", 1000))
ref = None
for name, opts in (("plain", {"k3_spec": -1, "k3_dense": -1}), ("speculative", {"k3_spec": 1, "k3_dense": -1}), ("dense", {"k3_spec": -1, "k3_dense": 1})):
    ctx = host.Context()
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    host.load_scene(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.chains_init(n, seed=1)
    ctx.reserve_visits(4 * iters + 2000)
    ctx.chains_run(0, 2000)
    ctx.set_timing(True)
    ms = []
    for _ in range(3):
        ctx.chains_run(0, iters); ms.append(ctx.last_kernel_ms(3))
    h = ctx.scalars()["trace_hash"]
    ref = h if ref is None else ref
    print(f"{name:12s} variant {ctx.last_variant()}  {np.mean(ms)*1e3/iters:6.2f} us/step  {n*iters/np.mean(ms)/1e3:7.1f} M proposals/s  same chains: {np.array_equal(h, ref)}", flush=True)
    ctx.close()
