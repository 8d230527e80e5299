"""BASELINE configs 3 and 5 on one GPU (throughput only; parity of these shapes is in tests/test_gpu_parity.py).
C3: 1920x1080, ~4000 superpixels, 512 chains (one GPU's share of 4096).  C5: 32 frames x 64 chains (one GPU's share
of 256 frames), moments + edge probabilities + chains."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene

def run_c3():
    t0 = time.time()
    sc = make_scene(width=1920, height=1080, n_planes=24, n_superpixels=4000, seed=2)
    print("C3 scene: %d superpixels, %d directed edges (generated in %.0f s)" % (sc.n_superpixels, sc.col.size, time.time() - t0), flush=True)
    ctx = host.Context(width=1920, height=1080)
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    host.load_scene(ctx, 0, sc)
    ctx.set_timing(True)
    ctx.frames_moments(); ctx.frames_edge_probs(); ctx.frames_moments(); ctx.frames_edge_probs()
    print("C3 K1 %.3f ms (%.0f GB/s on one frame), K2 %.3f ms" % (ctx.last_kernel_ms(0), (1920 * 1080 * 4 + sc.n_superpixels * 80) / ctx.last_kernel_ms(0) / 1e6, ctx.last_kernel_ms(2)))
    ctx.chains_init(512, seed=2)
    ctx.chains_run(200, 0)
    iters = int(os.environ.get("ITERS", 500))
    ctx.chains_run(0, iters)
    ms = ctx.last_kernel_ms(3)
    s = ctx.scalars()
    print("C3 chains: 512 x %d steps in %.1f ms -> %.1f us per proposal per chain, %.2f M proposals/s; status %s"
          % (iters, ms, ms * 1e3 / iters, 512 * iters / ms / 1e3, np.unique(s["status"])), flush=True)
    ctx.close()

def run_c5():
    F, cpf = 32, 64
    scs = [make_scene(seed=100 + f) for f in range(8)]
    smax = max(s.n_superpixels for s in scs); emax = max(s.col.size for s in scs)
    ctx = host.Context()
    ctx.frames_alloc(F, smax, emax)
    for f in range(F):
        host.load_scene(ctx, f, scs[f % 8])
    ctx.set_timing(True)
    ctx.frames_moments(); ctx.frames_edge_probs(); ctx.frames_moments(); ctx.frames_edge_probs()
    print("C5 (32 frames): K1 %.3f ms, K2 %.3f ms" % (ctx.last_kernel_ms(0), ctx.last_kernel_ms(2)))
    if os.environ.get("DENSE"):
        ctx.set_option("k3_dense", int(os.environ["DENSE"]))
    ctx.chains_init(F * cpf, chains_per_frame=cpf, seed=5)
    ctx.chains_run(1000, 0)
    iters = int(os.environ.get("ITERS5", 2000))
    ctx.chains_run(0, iters)
    ms = ctx.last_kernel_ms(3)
    s = ctx.scalars()
    print("C5 chains: %d x %d steps in %.1f ms -> %.1f us per proposal per chain, %.1f M proposals/s; status %s"
          % (F * cpf, iters, ms, ms * 1e3 / iters, F * cpf * iters / ms / 1e3, np.unique(s["status"])), flush=True)
    ctx.close()

run_c5()
if not os.environ.get("ONLY_C5"):
    run_c3()
