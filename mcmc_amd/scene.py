"""Synthetic planar RGB-D scenes and the superpixel graph the sampler consumes.

The reference renders synthetic scenes with OpenGL and adds Gaussian disparity
noise (reference: src/libraries/Simulation.cpp:700-749; disparity model
``delta = a*u + b*v + c``, include/MyMath.h:92-93).  There is no GL context
here, so scenes are generated directly in disparity space:

* K planes ``delta_k(u, v) = a_k*u + b_k*v + c_k``; plane regions are the Voronoi
  cells of K uniform sites.
* Superpixels are the Voronoi cells of jittered-grid sites, split where they
  straddle a plane boundary, so every superpixel lies on one plane.
* ``disp = clamp(round(delta + N(0, noise)), 1, 2047)`` as int16.
* Cells that would fail the reference's filter (fewer than 10 pixels, a single
  row/column; OverSegmentation.cpp:198-222) are left unlabelled (0xFFFF), as
  the reference drops them.

``build_adjacency`` restates the neighbour rule of createSuperPixels
(OverSegmentation.cpp:173-192): each pixel is compared with its LEFT, TOP,
TOP-LEFT and TOP-RIGHT neighbour through a ring buffer of W+1 ids.  The
TOP-LEFT slot is overwritten by the current pixel before it is read
(``previousIDs[k % (W+1)] = id`` at :174, and ``(k-W-1) % (W+1) == k % (W+1)``),
so that comparison never fires; the quirk is kept.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

NOLABEL = 0xFFFF


@dataclass
class Scene:
    width: int
    height: int
    disp: np.ndarray      # int16 [H, W]
    label: np.ndarray     # uint16 [H, W], NOLABEL = no superpixel
    rgb: np.ndarray       # uint8 [H, W, 3]
    n_superpixels: int
    plane_of_sp: np.ndarray   # int32 [S] ground-truth plane index
    planes: np.ndarray        # float64 [K, 3] (a, b, c) in raw pixel coordinates
    row_ptr: np.ndarray   # uint32 [S+1]
    col: np.ndarray       # uint32 [E2], ascending within a row


def _nearest_site(h, w, sites):
    from scipy.spatial import cKDTree
    yy, xx = np.mgrid[0:h, 0:w]
    pts = np.stack([xx.ravel(), yy.ravel()], axis=1).astype(np.float64)
    _, idx = cKDTree(sites).query(pts)
    return idx.reshape(h, w)


def build_adjacency(label: np.ndarray, n_sp: int):
    """CSR adjacency (symmetric, ascending ids) by the createSuperPixels rule."""
    lab = label.astype(np.int64)
    pairs = []

    def add(a, b):
        m = (a != b) & (a < n_sp) & (b < n_sp)
        if m.any():
            pairs.append(np.stack([a[m], b[m]], axis=1))

    add(lab[:, 1:].ravel(), lab[:, :-1].ravel())           # LEFT
    add(lab[1:, :].ravel(), lab[:-1, :].ravel())           # TOP
    add(lab[1:, :-1].ravel(), lab[:-1, 1:].ravel())        # TOP-RIGHT
    # TOP-LEFT never fires in the reference (ring-buffer slot already overwritten).
    if pairs:
        p = np.concatenate(pairs, axis=0)
        p = np.concatenate([p, p[:, ::-1]], axis=0)
        key = np.unique(p[:, 0] * n_sp + p[:, 1])
        src, dst = key // n_sp, key % n_sp
    else:
        src = dst = np.zeros(0, dtype=np.int64)
    row_ptr = np.zeros(n_sp + 1, dtype=np.uint32)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.uint32)
    return row_ptr, dst.astype(np.uint32)


def make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=1,
               noise=1.0, jitter=0.35) -> Scene:
    rng = np.random.Generator(np.random.Philox(seed))
    # planes
    psites = np.stack([rng.uniform(0, width, n_planes), rng.uniform(0, height, n_planes)], axis=1)
    planes = np.stack([rng.uniform(-0.05, 0.05, n_planes), rng.uniform(-0.05, 0.05, n_planes),
                       rng.uniform(200.0, 900.0, n_planes)], axis=1)
    plane_map = _nearest_site(height, width, psites)
    # superpixel sites on a jittered grid
    aspect = width / height
    gy = max(1, int(round(np.sqrt(n_superpixels / aspect))))
    gx = max(1, int(round(n_superpixels / gy)))
    cw, ch = width / gx, height / gy
    cx, cy = np.meshgrid((np.arange(gx) + 0.5) * cw, (np.arange(gy) + 0.5) * ch)
    sites = np.stack([cx.ravel() + rng.uniform(-jitter, jitter, gx * gy) * cw,
                      cy.ravel() + rng.uniform(-jitter, jitter, gx * gy) * ch], axis=1)
    site_map = _nearest_site(height, width, sites)
    # split cells at plane boundaries, renumber in raster first-appearance order
    combo = site_map.astype(np.int64) * n_planes + plane_map
    uniq, first, inv = np.unique(combo.ravel(), return_index=True, return_inverse=True)
    order = np.argsort(first)
    remap = np.empty_like(order)
    remap[order] = np.arange(order.size)
    lab = remap[inv].reshape(height, width)
    # drop cells the reference's filter would drop
    n_all = order.size
    ys, xs = np.mgrid[0:height, 0:width]
    cnt = np.bincount(lab.ravel(), minlength=n_all)
    xmin = np.full(n_all, width); xmax = np.full(n_all, -1)
    ymin = np.full(n_all, height); ymax = np.full(n_all, -1)
    np.minimum.at(xmin, lab.ravel(), xs.ravel()); np.maximum.at(xmax, lab.ravel(), xs.ravel())
    np.minimum.at(ymin, lab.ravel(), ys.ravel()); np.maximum.at(ymax, lab.ravel(), ys.ravel())
    keep = (cnt >= 10) & (xmax > xmin) & (ymax > ymin)
    new_id = np.full(n_all, NOLABEL, dtype=np.int64)
    new_id[keep] = np.arange(int(keep.sum()))
    label = new_id[lab].astype(np.uint16)
    n_sp = int(keep.sum())
    plane_of_cell = (uniq % n_planes)[order]
    plane_of_sp = plane_of_cell[keep].astype(np.int32)
    # disparity
    a = planes[plane_map, 0]; b = planes[plane_map, 1]; c = planes[plane_map, 2]
    delta = a * xs + b * ys + c + rng.normal(0.0, noise, (height, width))
    disp = np.clip(np.rint(delta), 1, 2047).astype(np.int16)
    base = rng.integers(40, 216, size=(n_planes, 3))
    rgb = np.clip(base[plane_map] + rng.normal(0, 8, (height, width, 3)), 0, 255).astype(np.uint8)
    row_ptr, col = build_adjacency(label, n_sp)
    return Scene(width, height, disp, label, rgb, n_sp, plane_of_sp, planes, row_ptr, col)


# ---------------------------------------------------------------------------
# Path B: a world graph (SPGraph) over several synthetic views
# ---------------------------------------------------------------------------
@dataclass
class MultiviewScene:
    n_vertices: int
    pt_ptr: np.ndarray        # uint32 [V+1]
    points: np.ndarray        # float32 [P, 4]  x y z pad (PCL PointXYZ stride)
    row_ptr: np.ndarray       # uint32 [V+1]
    col: np.ndarray           # uint32 [E2] ascending within a row
    q: np.ndarray             # float64 [E2] SPEdge::BernoulliProbability (symmetric)
    view_of_vertex: np.ndarray
    plane_of_vertex: np.ndarray
    theta: np.ndarray         # float32 [V, 7] SPNode::getFullModel(): plane (4) + centroid (3)


DEFAULT_WEIGHTS = np.array([5.0, 5.0, 5.0, 10.0, 0.5, 0.5, 0.5], np.float32)  # SPNode::getDefaultWeights, SPGraph.cpp:76-82


def pmerge_np(t1, t2, w=DEFAULT_WEIGHTS, temperature=8.0):
    """pMerge of GraphUtils.cpp:11-20 in numpy f64 (used to synthesise edge probabilities)."""
    e = w.astype(np.float64) * np.abs(t2.astype(np.float64) - t1.astype(np.float64))
    return np.exp(-(e * e).sum(axis=-1) * temperature / 2.0)


def make_multiview(n_views=3, width=160, height=120, n_planes=4, n_superpixels=40, seed=7,
                   focal=580.0, baseline=0.10, noise=0.002, in_view_threshold=0.01,
                   cross_view_threshold=0.2, shift=0.05) -> MultiviewScene:
    """`n_views` views of one scene of `n_planes` world planes seen from cameras
    translated by `shift` metres along x.  Each view is over-segmented like
    make_scene; every superpixel becomes a vertex with its XYZ points (world frame,
    z = focal*baseline/disparity as in MyMath.h:94-115), a fitted plane and a
    centroid.  In-view edges follow the 8-neighbour adjacency with
    p = pMerge > in_view_threshold (GraphUtils.cpp:52-138, SimpleSeg.cpp:406);
    cross-view edges join every pair with pMerge > cross_view_threshold
    (mergeNewScanGraph GraphUtils.cpp:140-202, SimpleSeg.cpp:409)."""
    rng = np.random.Generator(np.random.Philox(seed))
    # world planes n.X = d in front of the cameras, roughly fronto-parallel
    normals = np.stack([rng.uniform(-0.35, 0.35, n_planes), rng.uniform(-0.35, 0.35, n_planes),
                        np.ones(n_planes)], axis=1)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    dists = rng.uniform(1.0, 3.0, n_planes)
    psites = np.stack([rng.uniform(0, width, n_planes), rng.uniform(0, height, n_planes)], axis=1)
    cx, cy = width / 2.0, height / 2.0
    pts_all, ptr, view_of, plane_of, thetas, adj = [], [0], [], [], [], []
    base = 0
    for k in range(n_views):
        cam = np.array([shift * k, 0.0, 0.0])
        # region map: the plane sites move by the parallax of a 2 m deep point
        sites_k = psites - np.array([focal * shift * k / 2.0, 0.0])
        plane_map = _nearest_site(height, width, sites_k)
        sc = make_scene(width, height, n_planes, n_superpixels, seed=seed * 31 + k)
        # re-split superpixels by this view's plane map
        combo = sc.label.astype(np.int64) * n_planes + plane_map
        combo[sc.label == NOLABEL] = -1
        uniq, inv = np.unique(combo.ravel(), return_inverse=True)
        lab = inv.reshape(height, width).astype(np.int64)
        if uniq[0] == -1:
            lab -= 1
            uniq = uniq[1:]
        cnt = np.bincount(lab[lab >= 0].ravel(), minlength=uniq.size)
        keep = cnt >= 10
        new_id = np.full(uniq.size, -1, np.int64); new_id[keep] = np.arange(int(keep.sum()))
        lab = np.where(lab >= 0, new_id[np.maximum(lab, 0)], -1)
        nv = int(keep.sum())
        ys, xs = np.mgrid[0:height, 0:width]
        ray = np.stack([(xs - cx) / focal, (ys - cy) / focal, np.ones_like(xs, dtype=np.float64)], axis=-1)
        n = normals[plane_map]; d = dists[plane_map]
        z = (d - n @ cam) / np.einsum("hwc,hwc->hw", n, ray)
        z = z + rng.normal(0.0, noise, z.shape)
        X = cam + ray * z[..., None]
        for s in range(nv):
            m = lab == s
            P = X[m].astype(np.float32)
            pts_all.append(np.concatenate([P, np.ones((P.shape[0], 1), np.float32)], axis=1))
            ptr.append(ptr[-1] + P.shape[0])
            view_of.append(k)
            plane_of.append(int(np.bincount(plane_map[m]).argmax()))
            c = P.astype(np.float64).mean(axis=0)
            _, _, vt = np.linalg.svd(P.astype(np.float64) - c, full_matrices=False)
            nn = vt[2]
            if nn[2] < 0:
                nn = -nn
            thetas.append(np.concatenate([nn, [-(nn @ c)], c]).astype(np.float32))
        lab16 = np.where(lab >= 0, lab, NOLABEL).astype(np.uint16)
        rp, cl = build_adjacency(lab16, nv)
        src = np.repeat(np.arange(nv), np.diff(rp.astype(np.int64)))
        adj.append(np.stack([src + base, cl.astype(np.int64) + base], axis=1))
        base += nv
    V = base
    theta = np.stack(thetas)
    view_of = np.array(view_of, np.int32)
    pairs, probs = [], []
    a = np.concatenate(adj, axis=0)
    a = a[a[:, 0] < a[:, 1]]
    p = pmerge_np(theta[a[:, 0]], theta[a[:, 1]])
    sel = p > in_view_threshold
    pairs.append(a[sel]); probs.append(p[sel])
    ii, jj = np.nonzero(view_of[:, None] < view_of[None, :])
    p = pmerge_np(theta[ii], theta[jj])
    sel = p > cross_view_threshold
    pairs.append(np.stack([ii[sel], jj[sel]], axis=1)); probs.append(p[sel])
    e = np.concatenate(pairs, axis=0); pe = np.concatenate(probs)
    both = np.concatenate([e, e[:, ::-1]], axis=0); pb = np.concatenate([pe, pe])
    order = np.lexsort((both[:, 1], both[:, 0]))
    both = both[order]; pb = pb[order]
    row_ptr = np.zeros(V + 1, np.uint32)
    np.add.at(row_ptr, both[:, 0] + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.uint32)
    return MultiviewScene(V, np.array(ptr, np.uint32), np.ascontiguousarray(np.concatenate(pts_all, axis=0)),
                          row_ptr, both[:, 1].astype(np.uint32), pb.astype(np.float64), view_of,
                          np.array(plane_of, np.int32), theta)
