"""Config 1 on the GPU: stage times of the over-segmentation front end on the reference's bundled frame,
then the sampler on the resulting superpixel graph."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_frame.npz"))
disp, col = z["disp"], z["colors"]
o = host.OverSegmenter(640, 480)
o.run(col, disp)
t0 = time.perf_counter()
for _ in range(3):
    S, e2 = o.run(col, disp)
wall = (time.perf_counter() - t0) / 3 * 1e3
print("C1: %d superpixels, %d directed edges; stage ms %s; wall %.1f ms per frame" % (S, e2, o.last_ms(), wall))
# frames are independent: mcmc_overseg_batch_run hands them out to worker threads, one handle (own stream, own pinned
# buffers) each; 64 copies of the bundled frame, both tie orders
nf = int(os.environ.get("FRAMES", 64))
cols = np.ascontiguousarray(np.broadcast_to(col, (nf,) + col.shape)); disps = np.ascontiguousarray(np.broadcast_to(disp, (nf,) + disp.shape))
for tie in (1, 0):
    for nw in [int(x) for x in os.environ.get("WORKERS", "8 16").split()]:
        b = host.OverSegmenterBatch(640, 480, n_workers=nw)
        b.set_option("tie_order", tie)
        b.run(cols[:nw], disps[:nw])
        t0 = time.perf_counter()
        Sb, _ = b.run(cols, disps)
        wall_mt = time.perf_counter() - t0
        assert (Sb == Sb[0]).all()
        print("tie_order %d, batch of %d frames on %2d worker threads: %.1f ms -> %.2f ms per frame, %.0f frames/s"
              % (tie, nf, nw, wall_mt * 1e3, wall_mt * 1e3 / nf, nf / wall_mt), flush=True)
        b.close()
ctx = host.Context(width=640, height=480)
ctx.frames_alloc(1, S, e2)
d_disp, d_label = o.device_buffers()
ctx.frame_upload_device(0, d_disp, d_label, S)
rp, cl = o.graph()
ctx.frame_set_graph(0, rp, cl)
ctx.set_timing(True)
ctx.frames_moments(); ctx.frames_edge_probs(); ctx.frames_moments(); ctx.frames_edge_probs()   # second launch: no first-use overhead
fit, valid, _ = ctx.local_fit(0)
print("valid superpixels %d of %d; K1 %.3f ms, K2 %.3f ms" % (valid.sum(), S, ctx.last_kernel_ms(0), ctx.last_kernel_ms(2)))
n = int(os.environ.get("CHAINS", 1024)); iters = int(os.environ.get("ITERS", 2000))
ctx.chains_init(n, seed=1)
ctx.chains_run(1000, 0)
ctx.chains_run(0, iters)
ms = ctx.last_kernel_ms(3)
s = ctx.scalars()
lab = ctx.labels(2, True)
nseg = [int(l[l != 0xFFFF].max()) + 1 for l in lab[:16]]
print("chains: %d x %d steps in %.1f ms -> %.2f us per proposal per chain, %.1f M proposals/s; accept %.3f; segments (most visited) %s"
      % (n, iters, ms, ms * 1e3 / iters, n * iters / ms / 1e3, s["n_accept"].sum() / (n * (1000 + iters)), nseg[:8]))
print("status", np.unique(s["status"]))
