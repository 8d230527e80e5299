"""Tuning sweep: frame stride padding and label-plane skew against the moment kernel's time (C5 batch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene
F = int(os.environ.get("F", 256))
scs = [make_scene(seed=s) for s in (1, 100, 101, 102)]
ref = None
combos = [tuple(map(int, c.split(":"))) for c in sys.argv[1:]] or [(0, 0)]
for pad, skew in combos:
    ctx = host.Context()
    ctx.set_option("stride_pad", pad); ctx.set_option("label_skew", skew)
    ctx.frames_alloc(F, max(s.n_superpixels for s in scs), 8)
    for f in range(F):
        sc = scs[f % 4]
        ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
    ctx.set_timing(True)
    for _ in range(3):
        ctx.frames_moments()
    ms = []
    for _ in range(20):
        ctx.frames_moments(); ms.append(ctx.last_kernel_ms(0))
    m = ctx.moments(F - 1)
    ref = m if ref is None else ref
    print(f"pad={pad:6d} px skew={skew:8d} B  avg={np.mean(ms)*1e3:7.1f} us  min={np.min(ms)*1e3:7.1f} us  ok={np.array_equal(m, ref)}", flush=True)
    ctx.close()
