"""Generates tests/golden/small_frames.npz.

The reference holds no golden vectors for this path, so these
are minted from tests/literal_ref.py -- a numpy emulation of the reference's
gtsam call sequence in raw pixel coordinates -- NOT from oracle/ and not from
the HIP code.  Inputs are small synthetic frames; expected outputs are
moments (plain numpy integer sums), local plane parameters, segment
log-probabilities for several member sets, and edge probabilities.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mcmc_amd.scene import make_scene  # noqa: E402
from tests import literal_ref as LR  # noqa: E402

CASES = [
    dict(width=96, height=64, n_planes=3, n_superpixels=24, seed=11, noise=1.0),
    dict(width=128, height=96, n_planes=4, n_superpixels=40, seed=12, noise=0.5),
    dict(width=64, height=48, n_planes=2, n_superpixels=12, seed=13, noise=2.0),
]


def numpy_moments(disp, label, S):
    H, W = disp.shape
    v, u = np.mgrid[0:H, 0:W]
    u = u.astype(np.int64); v = v.astype(np.int64); b = disp.astype(np.int64)
    out = np.zeros((S, 10), np.int64)
    m = label < S
    lab = label[m].astype(np.int64)
    for k, arr in enumerate([np.ones_like(u), u, v, u * u, u * v, v * v, b, u * b, v * b, b * b]):
        out[:, k] = np.bincount(lab, weights=None, minlength=S) if k == 0 else \
            np.array([arr[m][lab == s].sum() for s in range(S)], np.int64)
    return out


def main():
    out = {}
    rng = np.random.default_rng(2024)
    for ci, kw in enumerate(CASES):
        sc = make_scene(**kw)
        S = sc.n_superpixels
        dens = LR.frame_densities(sc.disp, sc.label, S)
        theta = np.array([d[2] for d in dens])
        lerr = np.array([d[3] for d in dens])
        # member sets: singletons, same-plane groups, random subsets, everything
        sets = [[0], [S - 1], list(range(S))]
        for p in np.unique(sc.plane_of_sp):
            sets.append(np.nonzero(sc.plane_of_sp == p)[0].tolist())
        for _ in range(6):
            k = int(rng.integers(2, max(3, S // 2)))
            sets.append(sorted(rng.choice(S, size=k, replace=False).tolist()))
        probL, th, err = [], [], []
        for ms in sets:
            p, t, e = LR.segment_probL([(dens[i][0], dens[i][1]) for i in ms])
            probL.append(p); th.append(t); err.append(e)
        q = np.zeros(sc.col.size)
        src = np.repeat(np.arange(S), np.diff(sc.row_ptr.astype(np.int64)))
        for e_, (i, j) in enumerate(zip(src, sc.col)):
            q[e_] = LR.edge_probability((dens[i][0], dens[i][1]), (dens[j][0], dens[j][1]))
        pre = f"c{ci}_"
        out[pre + "disp"] = sc.disp
        out[pre + "label"] = sc.label
        out[pre + "S"] = np.int64(S)
        out[pre + "row_ptr"] = sc.row_ptr
        out[pre + "col"] = sc.col
        out[pre + "moments"] = numpy_moments(sc.disp, sc.label, S)
        out[pre + "theta_raw"] = theta
        out[pre + "local_err"] = lerr
        out[pre + "set_ptr"] = np.cumsum([0] + [len(s) for s in sets]).astype(np.int64)
        out[pre + "set_idx"] = np.concatenate([np.array(s, np.int64) for s in sets])
        out[pre + "set_probL"] = np.array(probL)
        out[pre + "set_theta_raw"] = np.array(th)
        out[pre + "set_err"] = np.array(err)
        out[pre + "q"] = q
    out["n_cases"] = np.int64(len(CASES))
    path = os.path.join(ROOT, "tests", "golden", "small_frames.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
