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
# frames are independent: one handle (own stream, own pinned buffers) per host thread; the C call releases the GIL
import threading
nthr = int(os.environ.get("THREADS", 8)); per = 4
hs = [host.OverSegmenter(640, 480) for _ in range(nthr)]
for h in hs:
    h.run(col, disp)
def work(h):
    for _ in range(per):
        h.run(col, disp)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(h,)) for h in hs]
[t.start() for t in th]; [t.join() for t in th]
wall_mt = time.perf_counter() - t0
print("%d host threads, one handle each: %d frames in %.1f ms -> %.2f ms per frame, %.0f frames/s"
      % (nthr, nthr * per, wall_mt * 1e3, wall_mt * 1e3 / (nthr * per), nthr * per / wall_mt))
for h in hs:
    h.close()
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
