"""Debug aid: per-superpixel differences of the moment kernel against numpy on one golden frame."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "small_frames.npz"))
ci = int(os.environ.get("CASE", 0))
disp, label, S, want = g[f"c{ci}_disp"], g[f"c{ci}_label"], int(g[f"c{ci}_S"]), g[f"c{ci}_moments"]
H, W = disp.shape
for ver in (8, 16):
    ctx = host.Context(width=W, height=H)
    ctx.frames_alloc(1, S, 8)
    ctx.set_option("k1_px", ver)
    ctx.frame_upload(0, disp, label, S)
    ctx.frames_moments()
    got = ctx.moments(0)
    bad = np.argwhere(got != want)
    print("k1_px", ver, "mismatching cells", len(bad), "superpixels", sorted(set(bad[:, 0].tolist())))
    for s in sorted(set(bad[:, 0].tolist()))[:6]:
        ys, xs = np.nonzero(label == s)
        print("  sp", s, "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "diff", (got[s] - want[s]).tolist())
    ctx.close()
