import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_scene
from tests import oracle_lib as O
for (W,H,S,K,lds) in [(640,480,500,8,0),(1920,1080,4000,24,-1),(1920,1080,600,8,-1),(1920,1080,600,8,0)]:
    sc = make_scene(width=W, height=H, n_planes=K, n_superpixels=S, seed=2)
    ctx = host.Context(width=W, height=H)
    ctx.frames_alloc(1, sc.n_superpixels, 8)
    ctx.frame_upload(0, sc.disp, sc.label, sc.n_superpixels)
    ctx.set_option("k1_use_lds", lds)
    ctx.frames_moments()
    got = ctx.moments(0); want = O.moments(sc.disp, sc.label, sc.n_superpixels)
    bad = np.nonzero((got != want).any(axis=1))[0]
    print(W,H,sc.n_superpixels,"lds",lds,"bad rows", bad.size, bad[:10])
    if bad.size:
        i = bad[0]; print(" got ", got[i]); print(" want", want[i]); print(" diff", got[i]-want[i])
        print(" colsum diff", (got-want).sum(axis=0))
    ctx.close()
