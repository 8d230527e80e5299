"""Literal numpy/float64 emulation of the reference's gtsam call sequence.

The reference delegates its arithmetic to gtsam 2.x (absent here, un-pinned;
SURVEY.md section 8c).  This module restates, call by call, what those gtsam calls
compute, in raw (un-centred) pixel coordinates exactly as the reference does,
using dense QR like gtsam's elimination.  It is independent of oracle/ and of
the HIP code (different algorithm: QR on stacked factors vs. normal equations
on centred integer moments) and is what pins both.

* ``local_density``  -- solve(): include/MyMath.h:120-154 (Diagonal::Variances
  whitening by 1/sigma, QR elimination -> GaussianDensity (R, d); error =
  graph.error(mean) = 1/2 |whitened residual|^2).
* ``fuse``           -- fit(): include/MyMath.h:189-275 (one JacobianFactor
  [R_i | d_i] per superpixel, eliminate, sum of factor errors at the mean).
* ``segment_probL``  -- Segment::computeProbability: src/libraries/Graph.cpp:116-138.
* ``edge_probability`` -- Problem::computeEdgeProbability: src/libraries/Problem.cpp:420-445
  with addLogProbs: include/MyMath.h:284-296.
"""
from __future__ import annotations

import numpy as np

K_DISPARITY_VARIANCE = 7.0  # include/MyMath.h:28


def superpixel_rows(disp, label, s):
    v, u = np.nonzero(label == s)
    b = disp[v, u].astype(np.float64)
    A = np.stack([u.astype(np.float64), v.astype(np.float64), np.ones(u.size)], axis=1)
    return A, b


def local_density(A, b, var=K_DISPARITY_VARIANCE):
    sig = np.sqrt(var)
    Ab = np.hstack([A / sig, (b / sig)[:, None]])
    R = np.linalg.qr(Ab, mode="r")
    R3, d3 = R[:3, :3], R[:3, 3]
    err = 0.5 * (R[3, 3] ** 2 if R.shape[0] > 3 else 0.0)
    theta = np.linalg.solve(R3, d3)
    return R3, d3, theta, err


def fuse(densities):
    stack = np.vstack([np.hstack([R, d[:, None]]) for (R, d) in densities])
    Rf = np.linalg.qr(stack, mode="r")
    R3, d3 = Rf[:3, :3], Rf[:3, 3]
    theta = np.linalg.solve(R3, d3)
    total = 0.0
    for (R, d) in densities:
        r = R @ theta - d
        total += 0.5 * float(r @ r)
    return R3, d3, theta, total


def segment_probL(densities):
    R3, d3, theta, err = fuse(densities)
    cov = np.linalg.inv(R3.T @ R3)
    det = np.linalg.det(2.0 * np.pi * cov)
    p = -err + np.log(np.sqrt(det))
    if np.isnan(p):
        p = 0.75  # Graph.cpp:134
    return p, theta, err


def add_log_probs(l1, l2):
    lB, lS = (l1, l2) if l1 > l2 else (l2, l1)
    return lB + np.log(1.0 + np.exp(lS - lB))


def edge_probability(dens_i, dens_j):
    T, _, _ = segment_probL([dens_i, dens_j])
    Si, _, _ = segment_probL([dens_i])
    Sj, _, _ = segment_probL([dens_j])
    Sep = Si + Sj
    return float(np.exp(T - add_log_probs(T, Sep)))


def frame_densities(disp, label, S, var=K_DISPARITY_VARIANCE):
    out = []
    for s in range(S):
        A, b = superpixel_rows(disp, label, s)
        R, d, theta, err = local_density(A, b, var)
        out.append((R, d, theta, err))
    return out
