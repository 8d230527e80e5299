"""Speculation statistics of the chain kernel (a build with -DMC_SPEC_STATS: MCMC_BUILD_DEFS=-DMC_SPEC_STATS
MCMC_BUILD_TAG=stats python mcmc_amd/build.py; run with MCMC_AMD_LIB=mcmc_amd/libmcmc_amd_stats.so)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene
sc = make_scene(seed=1)
n = int(os.environ.get("CHAINS", 1024)); iters = int(os.environ.get("ITERS", 1000))
for spec in [int(x) for x in os.environ.get("SPECS", "2 3 4").split()]:
    ctx = host.Context()
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    host.load_scene(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.set_option("k3_spec", spec); ctx.set_option("k3_dense", -1)
    ctx.chains_init(n, seed=1)
    ctx.reserve_visits(5 * iters + 2000)
    ctx.chains_run(0, 2000)
    ctx.set_timing(True)
    ctx.set_option("k3_profile", 1)
    ctx.chains_run(0, iters)
    ms = ctx.last_kernel_ms(3)
    p = ctx.chain_profile().astype(np.float64)
    waves = n * max(spec, 1)
    tot = p[6]
    print(f"k3_spec {spec} variant {ctx.last_variant()} {ms*1e3/iters:.2f} us/step | per wave: wait {100*p[0]/tot:.1f}% eval {100*p[1]/tot:.1f}% commit {100*p[5]/tot:.1f}% of {tot/waves:.0f} cycles"
          f" | evaluations/step {p[2]/(n*iters):.3f}  kept by label check/step {p[3]/(n*iters):.4f}  redone/step {p[4]/(n*iters):.4f}"
          f" | cycles per evaluation {p[1]/p[2]:.0f}", flush=True)
    ctx.close()
