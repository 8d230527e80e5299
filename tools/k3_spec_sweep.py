"""K3 at C2 (1024 chains): time per step of the speculative form with 2, 3, 4 waves per chain against the plain form;
checks that every form gives the same chains (trace hashes).  SPECS="-1 2 3 4" CHAINS=1024 ITERS=1000."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene
sc = make_scene(seed=1)
n = int(os.environ.get("CHAINS", 1024)); iters = int(os.environ.get("ITERS", 1000))
ref = None
for spec in [int(x) for x in os.environ.get("SPECS", "-1 2 3 4").split()]:
    ctx = host.Context()
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    host.load_scene(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.set_option("k3_spec", spec); ctx.set_option("k3_dense", -1)
    ctx.chains_init(n, seed=1)
    ctx.reserve_visits(5 * iters + 2000)
    ctx.chains_run(0, 2000)
    ctx.set_timing(True)
    ms = []
    for _ in range(4):
        ctx.chains_run(0, iters); ms.append(ctx.last_kernel_ms(3))
    s = ctx.scalars()
    h = s["trace_hash"]
    ref = h if ref is None else ref
    print(f"k3_spec {spec:2d} variant {ctx.last_variant():4d}  {np.mean(ms[1:])*1e3/iters:6.2f} us/step  {n*iters/np.mean(ms[1:])/1e3:7.1f} M proposals/s"
          f"  status {np.unique(s['status']).tolist()}  same chains: {np.array_equal(h, ref)}", flush=True)
    ctx.close()
