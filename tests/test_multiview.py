"""Path B (multiview) tests: SURVEY.md section 8a rows B1-B4.

CPU part: oracle/multiview_oracle.c against what pins it -- numpy's f64 SVD of the
centred cloud (the formula of MathUtils::fitPlaneSVD, MathUtils.cpp:22-54), the
reference's own property tests (TestMathUtils.cpp:44-60, TestGraphUtils.cpp:55-71),
and invariants of the restated loop.  The sampler trajectory itself is "parity
unpinned" (the reference seeds from time()).
GPU part (-m gpu): the HIP path through the C ABI, bit-exact against the oracle.
"""
import math

import numpy as np
import pytest

from mcmc_amd.scene import DEFAULT_WEIGHTS, make_multiview, pmerge_np
from tests import oracle_lib as O

REL = 1e-6


@pytest.fixture(scope="module")
def mv_small():
    return make_multiview(n_views=3, width=96, height=72, n_planes=3, n_superpixels=24, seed=5)


def _numpy_fit(points_list):
    """fitPlaneSVD + computeProbability in numpy f64, literally (MathUtils.cpp:22-54,
    SPGraph.cpp:89-104, MultiviewSegment.cpp:71-96)."""
    cloud = np.concatenate(points_list, axis=0).astype(np.float64)[:, :3].T   # 3 x N
    centroid = cloud.mean(axis=1)
    U, S, _ = np.linalg.svd(cloud - centroid[:, None], full_matrices=True)
    n = U[:, 2]
    d = n @ centroid
    cov = U @ np.diag(S * S) @ U.T / 2
    err = sum(np.abs(p.astype(np.float64)[:, :3] @ n + d).mean() for p in points_list)
    prob = math.exp(-err / 2.0) * math.sqrt(2.0 * math.pi * np.linalg.det(cov))
    return n, d, np.linalg.det(cov), err, prob


def test_det_sqrt_against_libm():
    rng = np.random.default_rng(1)
    xs = np.concatenate([np.exp(rng.uniform(-700, 700, 3000)), rng.uniform(0.25, 4.0, 3000),
                         [1.0, 2.0, 4.0, 0.25, 1e-310, 5e-324, 1.7976931348623157e308, 0.0]])
    for x in xs:
        got, want = O.lib().orc_det_sqrt(float(x)), math.sqrt(x)
        assert abs(got - want) <= 2.3e-16 * want, (x, got, want)
    assert math.isnan(O.lib().orc_det_sqrt(-1.0)) and O.lib().orc_det_sqrt(math.inf) == math.inf


def test_det_rsqrt_against_libm():
    # the division-free 1 / sqrt(x) of the Jacobi rotations: arguments there are 1 + t^2 and 1 + theta^2 < 2^55
    rng = np.random.default_rng(2)
    xs = np.concatenate([np.exp(rng.uniform(0.0, 80.0, 3000)), rng.uniform(1.0, 2.0, 3000), rng.uniform(2.0, 4.0, 1000),
                         np.exp(rng.uniform(-700, 700, 1000)), [1.0, 2.0, 4.0, 1.0 + 2.0 ** -52, 2.0 - 2.0 ** -52, 2.0 ** 54 + 4.0]])
    worst = 0.0
    for x in xs:
        got, want = O.lib().orc_det_rsqrt(float(x)), 1.0 / math.sqrt(x)
        worst = max(worst, abs(got - want) / want)
    assert worst <= 4.5 * 2.0 ** -53, worst                  # measured: 4.0 (two units in the last place below 1)


def test_plane_svd_reference_property():
    # TestMathUtils.cpp:44-60: noisy plane rotated 45 degrees about x, r = 1, step 0.1, sigma 0.01
    rng = np.random.default_rng(3)
    g = np.arange(-1.0, 1.0 + 1e-6, 0.1, dtype=np.float32)
    x, y = np.meshgrid(g, g, indexing="ij")
    P = np.stack([x.ravel(), y.ravel(), np.zeros(x.size, np.float32)], axis=1) + rng.normal(0, 0.01, (x.size, 3))
    c, s = math.cos(math.pi / 4), math.sin(math.pi / 4)
    R = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    P = (P @ R.T).astype(np.float32)
    pts = np.concatenate([P, np.ones((P.shape[0], 1), np.float32)], axis=1)
    pt_ptr = np.array([0, pts.shape[0]], np.uint32)
    vm = O.mv_vertex_moments(pts, pt_ptr)
    pl, prob, err = O.mv_fit_segment(pts, pt_ptr, vm, [0])
    assert abs(np.linalg.norm(pl[:4]) - 1.0) < 1e-5          # CPPUNIT_ASSERT :57 (4-vector norm, d ~ 0)
    assert abs(pl[:3] @ R[:, 2]) > 0.9                        # :58
    n, d, det, e_np, p_np = _numpy_fit([pts])
    sgn = np.sign(n @ pl[:3])
    assert np.allclose(pl[:3], sgn * n, atol=1e-9) and abs(pl[3] - sgn * d) < 1e-9
    assert abs(pl[4] - det) <= REL * det and abs(err - e_np) <= REL * e_np and abs(prob - p_np) <= REL * p_np


def test_segment_fit_against_numpy_svd(mv_small):
    mv = mv_small
    vm = O.mv_vertex_moments(mv.points, mv.pt_ptr)
    # per-vertex moments are plain sums
    for v in (0, 5, mv.n_vertices - 1):
        P = mv.points[mv.pt_ptr[v]:mv.pt_ptr[v + 1], :3].astype(np.float64)
        want = [P.shape[0], *P.sum(0), (P[:, 0] ** 2).sum(), (P[:, 0] * P[:, 1]).sum(), (P[:, 0] * P[:, 2]).sum(),
                (P[:, 1] ** 2).sum(), (P[:, 1] * P[:, 2]).sum(), (P[:, 2] ** 2).sum()]
        assert np.allclose(vm[v], want, rtol=1e-12)
    rng = np.random.default_rng(0)
    sets = [[0], [1, 2], list(range(0, mv.n_vertices, 3)),
            list(np.flatnonzero(mv.plane_of_vertex == 0)), list(range(mv.n_vertices))]
    sets += [sorted(rng.choice(mv.n_vertices, 6, replace=False).tolist()) for _ in range(4)]
    for ms in sets:
        pl, prob, err = O.mv_fit_segment(mv.points, mv.pt_ptr, vm, ms)
        n, d, det, e_np, p_np = _numpy_fit([mv.points[mv.pt_ptr[v]:mv.pt_ptr[v + 1]] for v in ms])
        sgn = np.sign(n @ pl[:3])
        assert abs(np.linalg.norm(pl[:3]) - 1.0) < 1e-12
        assert np.allclose(pl[:3], sgn * n, atol=1e-7), (ms, pl, n)
        assert abs(pl[3] - sgn * d) <= 1e-7
        assert abs(pl[4] - det) <= REL * abs(det)
        assert abs(err - e_np) <= REL * e_np
        assert abs(prob - p_np) <= REL * p_np + 1e-300


def _hard_clouds():
    rng = np.random.default_rng(8)
    g = np.arange(-1.0, 1.01, 0.25)
    x, y = [a.ravel() for a in np.meshgrid(g, g, indexing="ij")]
    z0 = np.zeros_like(x)
    Rz = lambda t: np.array([[math.cos(t), -math.sin(t), 0], [math.sin(t), math.cos(t), 0], [0, 0, 1.0]])
    Rx = lambda t: np.array([[1.0, 0, 0], [0, math.cos(t), -math.sin(t)], [0, math.sin(t), math.cos(t)]])
    R = Rz(0.7) @ Rx(1.1)
    clouds = {
        "exact plane z = 0": np.stack([x, y, z0], 1),
        "exact plane, tilted": np.stack([x, y, z0], 1) @ R.T,
        "thin slab 1e-6": (np.stack([x, y, 1e-6 * rng.normal(size=x.size)], 1)) @ R.T,
        "spread 1e4 : 1 : 1e-3": (np.stack([1e4 * x, y, 1e-3 * rng.normal(size=x.size)], 1)) @ R.T,
        "far from the origin": np.stack([x, y, 0.01 * rng.normal(size=x.size)], 1) @ R.T + np.array([300.0, -200.0, 500.0]),
        "small scale 1e-3": 1e-3 * (np.stack([x, y, 0.01 * rng.normal(size=x.size)], 1) @ R.T),
        "nearly isotropic": rng.normal(size=(200, 3)) * np.array([1.0, 1.0 + 1e-9, 0.5]),
    }
    degenerate = {"a line": np.stack([g, 2 * g, -g], 1), "one point nine times": np.tile([[1.0, 2.0, 3.0]], (9, 1))}
    return clouds, degenerate


def test_plane_fit_hard_clouds():
    """The Jacobi sweeps of the plane fit stop by a relative criterion: clouds whose scatter has zero, tiny, equal or
    widely spread eigenvalues against numpy's SVD (normal up to sign where the smallest eigenvalue is simple)."""
    clouds, degenerate = _hard_clouds()
    for name, P in clouds.items():
        pts = np.concatenate([P.astype(np.float32), np.ones((P.shape[0], 1), np.float32)], axis=1)
        pt_ptr = np.array([0, pts.shape[0]], np.uint32)
        vm = O.mv_vertex_moments(pts, pt_ptr)
        pl, prob, err = O.mv_fit_segment(pts, pt_ptr, vm, [0])
        assert np.isfinite(pl).all() and abs(np.linalg.norm(pl[:3]) - 1.0) < 1e-12, (name, pl)
        n, d, det, e_np, p_np = _numpy_fit([pts])
        sgn = np.sign(n @ pl[:3])
        Pd = pts[:, :3].astype(np.float64)
        # the fitted normal leaves no more scatter than numpy's (the smallest eigenvalue, to rounding of the scatter matrix)
        C = (Pd - Pd.mean(0)).T @ (Pd - Pd.mean(0))
        lam_got, lam_np = pl[:3] @ C @ pl[:3], n @ C @ n
        assert lam_got <= lam_np + 1e-12 * np.trace(C), (name, lam_got, lam_np)
        if name != "nearly isotropic":
            assert np.allclose(pl[:3], sgn * n, atol=2e-6), (name, pl[:3], n)
    # degenerate: a line, and one point many times -- finite unit normal, zero determinant, probability 0
    for P in degenerate.values():
        pts = np.concatenate([P.astype(np.float32), np.ones((P.shape[0], 1), np.float32)], axis=1)
        pt_ptr = np.array([0, pts.shape[0]], np.uint32)
        pl, prob, err = O.mv_fit_segment(pts, pt_ptr, O.mv_vertex_moments(pts, pt_ptr), [0])
        assert np.isfinite(pl).all() and abs(np.linalg.norm(pl[:3]) - 1.0) < 1e-12
        assert abs(pl[4]) < 1e-9 and prob >= 0.0


def test_pmerge_reference_properties():
    # TestGraphUtils.cpp:55-71
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, 50).astype(np.float32)
    w = np.ones(50, np.float32)
    assert O.mv_pmerge(a, a, w) > 0.999999
    b = rng.uniform(-1, 1, 50).astype(np.float32)
    assert O.mv_pmerge(a, b, w) == O.mv_pmerge(b, a, w)
    t1 = rng.uniform(-1, 1, (20, 7)).astype(np.float32); t2 = t1 + rng.normal(0, 0.05, (20, 7)).astype(np.float32)
    want = pmerge_np(t1, t2)
    got = np.array([O.mv_pmerge(t1[i], t2[i], DEFAULT_WEIGHTS) for i in range(20)])
    assert np.allclose(got, want, rtol=1e-12)


def test_oracle_loop_invariants(mv_small):
    mv = mv_small
    ch = O.MvChains(mv, 4, seed=3)
    cur0 = ch.labels(0).copy()
    # initial partition: connected components numbered by smallest vertex, every edge inside
    # a component is allowed to be off, but no on-edge may cross components: ids are dense
    for k in range(4):
        ids = cur0[k]
        first = [int(np.flatnonzero(ids == s)[0]) for s in range(ids.max() + 1)]
        assert first == sorted(first)
    f, a = ch.run(60)
    s = ch.scalars()
    assert (s["status"] == 0).all()
    assert np.array_equal(ch.labels(0), cur0)                     # currentSegmentID is never updated
    assert (s["n_segments"] == cur0.max(axis=1) + 1 + s["n_split"]).all()   # every split appends a segment
    assert (s["n_accept"] == (f & 1).sum(axis=1)).all() and (s["n_split"] == ((f >> 1) & 1).sum(axis=1)).all()
    assert ((f & 4) == 0).all() or (a[(f & 4) != 0] == 0).all()
    # a split proposal leaves SB2 == SA2, so a == P(SB1)/P(SA1); all ratios finite or nan, never negative
    assert (np.isnan(a) | (a >= 0)).all()
    # determinism, chain offset, continuation
    ch2 = O.MvChains(mv, 2, seed=3, chain_offset=2)
    f2, a2 = ch2.run(60)
    assert np.array_equal(f2, f[2:]) and np.array_equal(a2.view(np.uint64), a[2:].view(np.uint64))
    ch3 = O.MvChains(mv, 4, seed=3)
    ch3.run(25); ch3.run(35)
    assert np.array_equal(ch3.scalars()["trace_hash"], s["trace_hash"])
    assert np.array_equal(ch3.labels(1), ch.labels(1))
    # the vertex sets stay a partition whose new segments are exactly the split components
    own = ch.labels(1)
    for k in range(4):
        assert own[k].max() < s["n_segments"][k]


def test_capacity_overflow_freezes_chain(mv_small):
    ch = O.MvChains(mv_small, 2, seed=1, max_new_segments=1, split_prob=1.0)
    f, _ = ch.run(5)
    assert (f[:, 0] & 2).all() and (f[:, 1:] == 0xFF).all()
    assert (ch.scalars()["status"] == -6).all()


# ------------------------------------------------------------------ GPU parity
@pytest.fixture(scope="module")
def gpu_mv():
    from mcmc_amd import host
    m = host.Multiview(device=0)
    yield m
    m.close()


def _load(gm, mv):
    gm.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)


@pytest.mark.gpu
def test_gpu_vertex_moments_and_fit_bit_exact(gpu_mv, mv_small):
    mv = mv_small
    _load(gpu_mv, mv)
    vm = O.mv_vertex_moments(mv.points, mv.pt_ptr)
    assert np.array_equal(gpu_mv.vertex_moments().view(np.uint64), vm.view(np.uint64))
    rng = np.random.default_rng(4)
    sets = [[0], [3, 4], list(range(mv.n_vertices))] + [sorted(rng.choice(mv.n_vertices, 9, replace=False).tolist())
                                                         for _ in range(5)]
    for ms in sets:
        pl, prob, err = gpu_mv.fit_segment(ms)
        opl, oprob, oerr = O.mv_fit_segment(mv.points, mv.pt_ptr, vm, ms)
        assert np.array_equal(pl.view(np.uint64), opl.view(np.uint64)), (ms, pl, opl)
        assert np.float64(prob).view(np.uint64) == np.float64(oprob).view(np.uint64)
        assert np.float64(err).view(np.uint64) == np.float64(oerr).view(np.uint64)


@pytest.mark.gpu
def test_gpu_plane_fit_hard_clouds_bit_exact(gpu_mv):
    """The clouds of test_plane_fit_hard_clouds, one vertex each: the device's fit against the oracle's, bit for bit
    (sweeps that stop early, rotations left out, the short form of a small angle, zero scatter)."""
    clouds, degenerate = _hard_clouds()
    Ps = list(clouds.values()) + list(degenerate.values())
    pts = np.concatenate([np.concatenate([P.astype(np.float32), np.ones((P.shape[0], 1), np.float32)], axis=1) for P in Ps])
    pt_ptr = np.concatenate([[0], np.cumsum([P.shape[0] for P in Ps])]).astype(np.uint32)
    V = len(Ps)
    rows = [[w for w in (v - 1, v + 1) if 0 <= w < V] for v in range(V)]            # a path graph
    row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint32)
    col = np.concatenate(rows).astype(np.uint32)
    gpu_mv.set_graph(pt_ptr, pts, row_ptr, col, np.full(col.size, 0.5))
    vm = O.mv_vertex_moments(pts, pt_ptr)
    assert np.array_equal(gpu_mv.vertex_moments().view(np.uint64), vm.view(np.uint64))
    for ms in [[v] for v in range(V)] + [[0, 1], [2, 4, 5], list(range(V))]:
        pl, prob, err = gpu_mv.fit_segment(ms)
        opl, oprob, oerr = O.mv_fit_segment(pts, pt_ptr, vm, ms)
        assert np.array_equal(pl.view(np.uint64), opl.view(np.uint64)), (ms, pl, opl)
        assert np.float64(prob).view(np.uint64) == np.float64(oprob).view(np.uint64)
        assert np.float64(err).view(np.uint64) == np.float64(oerr).view(np.uint64)


@pytest.mark.gpu
def test_gpu_pmerge_bit_exact(gpu_mv):
    rng = np.random.default_rng(6)
    t1 = rng.uniform(-1, 1, (300, 7)).astype(np.float32); t2 = t1 + rng.normal(0, 0.1, (300, 7)).astype(np.float32)
    got = gpu_mv.pmerge(t1, t2, DEFAULT_WEIGHTS)
    want = np.array([O.mv_pmerge(t1[i], t2[i], DEFAULT_WEIGHTS) for i in range(300)])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.array_equal(gpu_mv.pmerge(t1, t1, DEFAULT_WEIGHTS), np.ones(300))


@pytest.mark.gpu
@pytest.mark.parametrize("split_prob", [0.10, 0.5])
def test_gpu_chains_bit_exact(gpu_mv, mv_small, split_prob):
    mv = mv_small
    _load(gpu_mv, mv)
    n, steps = 6, 80
    gpu_mv.chains_init(n, seed=9, chain_offset=3, max_new_segments=64, split_prob=split_prob)
    gpu_mv.chains_set_trace(steps)
    orc = O.MvChains(mv, n, seed=9, chain_offset=3, max_new_segments=64, split_prob=split_prob)
    assert np.array_equal(gpu_mv.labels(0), orc.labels(0))       # initial partition
    s0 = gpu_mv.scalars()
    for k in range(n):
        assert np.array_equal(gpu_mv.planes(k, s0["n_segments"][k]).view(np.uint64),
                              orc.planes(k, s0["n_segments"][k]).view(np.uint64))
    gpu_mv.chains_run(steps)
    of, oa = orc.run(steps)
    gf, ga = gpu_mv.trace()
    assert np.array_equal(gf, of)                                 # accept / split / skip sequence
    assert np.array_equal(ga.view(np.uint64), oa.view(np.uint64)) # ratios bit for bit
    gs, os_ = gpu_mv.scalars(), orc.scalars()
    for key in gs:
        assert np.array_equal(gs[key], os_[key]), key
    assert np.array_equal(gpu_mv.labels(1), orc.labels(1))
    # a second launch continues the chains
    gpu_mv.chains_run(steps)
    orc.run(steps)
    assert np.array_equal(gpu_mv.scalars()["trace_hash"], orc.scalars()["trace_hash"])


@pytest.mark.gpu
def test_gpu_edge_trial_thresholds(gpu_mv, mv_small):
    """The kernel tries an edge as `draw <= threshold` (integers) where the oracle compares draw / 4294967295.0
    with BernoulliProbability (MultiviewSegmentation.cpp:23): probabilities at and next to representable
    draws, 0, 1, above 1 and denormal-small must give the same chains."""
    import dataclasses
    mv = mv_small
    rng = np.random.default_rng(12)
    src = np.repeat(np.arange(mv.n_vertices), np.diff(mv.row_ptr.astype(np.int64)))
    key = np.minimum(src, mv.col).astype(np.int64) * mv.n_vertices + np.maximum(src, mv.col)
    _, und = np.unique(key, return_inverse=True)                   # both directions of an edge share a value
    x = rng.integers(0, 2**32, und.max() + 1, dtype=np.uint64).astype(np.float64)
    exact = x / 4294967295.0
    pool = np.stack([exact, np.nextafter(exact, 0.0), np.nextafter(exact, 2.0), np.zeros_like(x), np.ones_like(x),
                     np.full_like(x, 1.5), np.full_like(x, 5e-324), rng.uniform(0, 1, x.size)])
    q = pool[rng.integers(0, pool.shape[0], x.size), np.arange(x.size)][und]
    q = np.maximum(q, 0.0)
    mv2 = dataclasses.replace(mv, q=q)
    _load(gpu_mv, mv2)
    n, steps = 5, 60
    gpu_mv.chains_init(n, seed=21, max_new_segments=64, split_prob=0.2)
    gpu_mv.chains_set_trace(steps)
    orc = O.MvChains(mv2, n, seed=21, max_new_segments=64, split_prob=0.2)
    assert np.array_equal(gpu_mv.labels(0), orc.labels(0))
    gpu_mv.chains_run(steps)
    of, oa = orc.run(steps)
    gf, ga = gpu_mv.trace()
    assert np.array_equal(gf, of) and np.array_equal(ga.view(np.uint64), oa.view(np.uint64))
    assert np.array_equal(gpu_mv.scalars()["trace_hash"], orc.scalars()["trace_hash"])
    # not a probability: refused (a negative or NaN value could never succeed; the reference's are exp(-d/T))
    for bad in (-1e-9, np.nan):
        qb = q.copy(); qb[3] = bad
        with pytest.raises(Exception):
            gpu_mv.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, qb)
    _load(gpu_mv, mv)


@pytest.mark.gpu
@pytest.mark.parametrize("n_chains", [37, 1030])
def test_gpu_matrix_l1_pass_shapes(gpu_mv, n_chains):
    """The L1 pass on the matrix cores at its edges: vertices of 1 to ~40 points (fewer than one block of 16, ragged
    last blocks), chain counts that are no multiple of 16 / 64 and that span more than one block of 1024 chains
    (second chain block of the pair lists and the queue), items with 1..4 groups of 16 pairs.  Split iteration
    = oracle, bit for bit, for every chain."""
    mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
    cnt = np.diff(mv.pt_ptr.astype(np.int64))
    assert cnt.min() < 16 and (cnt % 16 != 0).any() and cnt.max() > 16
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    gpu_mv.set_option("split_step", 1)
    steps = 6
    gpu_mv.chains_init(n_chains, seed=3, max_new_segments=8, split_prob=0.2)
    gpu_mv.chains_set_trace(steps)
    gpu_mv.chains_run(steps)
    gf, ga = gpu_mv.trace()
    gs = gpu_mv.scalars()
    gpu_mv.set_option("split_step", -1)
    orc = O.MvChains(mv, n_chains, seed=3, max_new_segments=8, split_prob=0.2)
    of, oa = orc.run(steps)
    assert np.array_equal(gf, of) and np.array_equal(ga.view(np.uint64), oa.view(np.uint64))
    s = orc.scalars()
    for k in ("n_accept", "n_split", "n_skip", "trace_hash", "n_segments", "status"):
        assert np.array_equal(gs[k], s[k]), k


@pytest.mark.gpu
def test_gpu_l1_queue_longer_than_the_launch(gpu_mv, monkeypatch):
    """A wave of the L1 kernel takes further items (item += waves of the launch) when the queue is longer than the
    launch -- 16384 waves, so only beyond ~2000 chains per group at BASELINE config 4.  Forced here with a launch of 64
    waves (tuning variable MCMC_MV_EWAVES, read at every run): same chains as with the full launch and as the oracle."""
    mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    n, steps = 300, 5
    got = []
    for ew in (None, "64"):
        if ew is None:
            monkeypatch.delenv("MCMC_MV_EWAVES", raising=False)
        else:
            monkeypatch.setenv("MCMC_MV_EWAVES", ew)
        gpu_mv.chains_init(n, seed=9, max_new_segments=8, split_prob=0.3)
        gpu_mv.chains_set_trace(steps)
        gpu_mv.chains_run(steps)
        f, a = gpu_mv.trace()
        got.append((f.copy(), a.copy(), gpu_mv.scalars()["trace_hash"].copy()))
    monkeypatch.delenv("MCMC_MV_EWAVES", raising=False)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1].view(np.uint64), got[1][1].view(np.uint64))
    assert np.array_equal(got[0][2], got[1][2])
    orc = O.MvChains(mv, 6, seed=9, max_new_segments=8, split_prob=0.3)
    of, oa = orc.run(steps)
    assert np.array_equal(got[1][0][:6], of) and np.array_equal(got[1][1][:6].view(np.uint64), oa.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("n_sp,n_chains,split_prob,steps", [(40, 700, 0.2, 4), (4, 1100, 1.0, 3)])
def test_gpu_matrix_l1_pass_large_vertices(gpu_mv, n_sp, n_chains, split_prob, steps):
    """The L1 pass with vertices of thousands of points: an item is then one vertex under 16, 32 or 48 pairs instead of 64
    (mv_err_groups_per_item keeps an item near 128 matrix instructions) and the point loads run 16 blocks ahead; in the
    second case 31 vertices, every proposal a split and 775 pairs per vertex and step on average, so the vertices of
    16 pairs per item make more items than the 64 lanes of the wave that queues them.  Split iteration = oracle, bit
    for bit, for every chain."""
    mv = make_multiview(n_views=2, width=320, height=240, n_planes=3, n_superpixels=n_sp, seed=13)
    cnt = np.diff(mv.pt_ptr.astype(np.int64))
    nblk = (cnt + 15) // 16
    groups = np.clip(128 // np.maximum(nblk, 1), 1, 4)
    assert (groups == 1).any() and cnt.max() > 2048 and (n_sp != 40 or set(groups.tolist()) >= {1, 2, 3})
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    gpu_mv.set_option("split_step", 1)
    gpu_mv.chains_init(n_chains, seed=5, max_new_segments=8, split_prob=split_prob)
    gpu_mv.chains_set_trace(steps)
    gpu_mv.chains_run(steps)
    gf, ga = gpu_mv.trace()
    gs = gpu_mv.scalars()
    gpu_mv.set_option("split_step", -1)
    orc = O.MvChains(mv, n_chains, seed=5, max_new_segments=8, split_prob=split_prob)
    of, oa = orc.run(steps)
    assert np.array_equal(gf, of) and np.array_equal(ga.view(np.uint64), oa.view(np.uint64))
    s = orc.scalars()
    for k in ("n_accept", "n_split", "n_skip", "trace_hash", "n_segments", "status"):
        assert np.array_equal(gs[k], s[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("split_prob,max_new", [(0.3, 16), (1.0, 2)])
def test_gpu_growth_kernel_apart_changes_nothing(gpu_mv, split_prob, max_new):
    """Option "split_growth": the first half of an iteration as a growth kernel (eight waves per chain) followed by a
    plane-fit kernel (automatic up to 256 chains).  Same chains as the single kernel and as the oracle, split proposals
    (the vertices move in the second of the two kernels) and the freeze at max_new_segments included."""
    mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    n, steps = 70, 9
    got = {}
    for sg in (0, 1):
        gpu_mv.set_option("split_growth", sg)
        gpu_mv.chains_init(n, seed=6, chain_offset=2, max_new_segments=max_new, split_prob=split_prob)
        gpu_mv.chains_set_trace(steps)
        gpu_mv.chains_run(steps)
        f, a = gpu_mv.trace()
        gpu_mv.chains_run(3)
        s = gpu_mv.scalars()
        got[sg] = (f.copy(), a.copy(), {k: v.copy() for k, v in s.items()}, gpu_mv.labels(1).copy())
    gpu_mv.set_option("split_growth", -1)
    assert np.array_equal(got[1][0], got[0][0]) and np.array_equal(got[1][1].view(np.uint64), got[0][1].view(np.uint64))
    for k in got[0][2]:
        assert np.array_equal(got[1][2][k], got[0][2][k]), k
    assert np.array_equal(got[1][3], got[0][3])
    if max_new == 2:
        assert (got[1][2]["status"] == -6).any()                 # some chains ran out of room for new segments
    orc = O.MvChains(mv, 5, seed=6, chain_offset=2, max_new_segments=max_new, split_prob=split_prob)
    of, oa = orc.run(steps)
    assert np.array_equal(got[1][0][:5], of) and np.array_equal(got[1][1][:5].view(np.uint64), oa.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("split_prob,max_new", [(0.2, 8), (0.9, 2)])
def test_gpu_fused_halves_change_nothing(gpu_mv, split_prob, max_new):
    """Option "fuse_halves" (the default of the split iteration): the second half of a step and the first half of the next
    run as one kernel, so a run of n steps is first half, n - 1 fused kernels, second half.  Same chains as with the
    halves apart and as the oracle, over runs of one step (no fused kernel at all), two and more, split proposals and
    the freeze at max_new_segments included."""
    mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    gpu_mv.set_option("split_growth", 0)
    n, runs = 70, (1, 2, 6)
    got = {}
    for fh in (0, 1):
        gpu_mv.set_option("fuse_halves", fh)
        gpu_mv.chains_init(n, seed=6, chain_offset=2, max_new_segments=max_new, split_prob=split_prob)
        fs, as_ = [], []
        for steps in runs:
            gpu_mv.chains_set_trace(steps)
            gpu_mv.chains_run(steps)
            f, a = gpu_mv.trace()
            fs.append(f.copy()); as_.append(a.copy())
        s = gpu_mv.scalars()
        got[fh] = (np.concatenate(fs, axis=1), np.concatenate(as_, axis=1), {k: v.copy() for k, v in s.items()}, gpu_mv.labels(1).copy())
    gpu_mv.set_option("fuse_halves", -1)
    gpu_mv.set_option("split_growth", -1)
    assert np.array_equal(got[1][0], got[0][0]) and np.array_equal(got[1][1].view(np.uint64), got[0][1].view(np.uint64))
    for k in got[0][2]:
        assert np.array_equal(got[1][2][k], got[0][2][k]), k
    assert np.array_equal(got[1][3], got[0][3])
    if max_new == 2:
        assert (got[1][2]["status"] == -6).any()                 # some chains ran out of room for new segments
    orc = O.MvChains(mv, 5, seed=6, chain_offset=2, max_new_segments=max_new, split_prob=split_prob)
    of, oa = orc.run(sum(runs))
    assert np.array_equal(got[1][0][:5], of) and np.array_equal(got[1][1][:5].view(np.uint64), oa.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("split_prob,max_new", [(0.2, 8), (0.9, 2)])
def test_gpu_growth_a_step_ahead_changes_nothing(gpu_mv, split_prob, max_new):
    """Option "growth_ahead" (the default up to 3072 chains): the growth of step t + 1 (a kernel of its own on a second stream; it needs nothing of the
    steps before it) runs while the main stream finishes step t - 1, fits the planes of step t and evaluates its
    residuals.  What the growth kernel cannot know a step ahead -- the start vertex's segment after a split, the id of
    a new segment, whether there is room for one -- is filled in by the kernel that uses the set.  Same chains as
    without it and as the oracle over runs of 1, 2, 3 and 6 steps, splits and the freeze at max_new_segments included."""
    mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    n, runs = 70, (1, 2, 3, 6)
    got = {}
    for ah, waves in ((0, 0), (1, 4), (2, 8)):                   # off; on with four / eight growth waves per chain
        gpu_mv.set_option("growth_ahead", min(ah, 1))
        gpu_mv.set_option("growth_ahead_waves", waves)
        gpu_mv.chains_init(n, seed=6, chain_offset=2, max_new_segments=max_new, split_prob=split_prob)
        fs, as_ = [], []
        for steps in runs:
            gpu_mv.chains_set_trace(steps)
            gpu_mv.chains_run(steps)
            f, a = gpu_mv.trace()
            fs.append(f.copy()); as_.append(a.copy())
        s = gpu_mv.scalars()
        got[ah] = (np.concatenate(fs, axis=1), np.concatenate(as_, axis=1), {k: v.copy() for k, v in s.items()}, gpu_mv.labels(1).copy())
    # a run leaves the growth of the next run's first step behind; another kind of run or an option in between discards it
    gpu_mv.set_option("growth_ahead", 1)
    gpu_mv.set_option("growth_ahead_waves", 0)
    gpu_mv.chains_init(n, seed=6, chain_offset=2, max_new_segments=max_new, split_prob=split_prob)
    gpu_mv.chains_run(2)
    gpu_mv.chains_run(0)                                         # launches nothing
    gpu_mv.chains_run(1)                                         # starts from the growth the run of two steps left
    gpu_mv.set_option("growth_ahead", 0)
    gpu_mv.chains_run(2)
    gpu_mv.set_option("growth_ahead", 1)
    gpu_mv.chains_run(3)
    gpu_mv.chains_set_trace(4)
    gpu_mv.chains_run(4)                                         # the last four of the twelve steps, traced
    f, a = gpu_mv.trace()
    assert np.array_equal(f, got[0][0][:, 8:]) and np.array_equal(a.view(np.uint64), got[0][1][:, 8:].view(np.uint64))
    s = gpu_mv.scalars()
    for k in got[0][2]:
        if k != "trace_hash":
            assert np.array_equal(s[k], got[0][2][k]), k
    assert np.array_equal(gpu_mv.labels(1), got[0][3])
    gpu_mv.set_option("growth_ahead", -1)
    for ah in (1, 2):
        assert np.array_equal(got[ah][0], got[0][0]) and np.array_equal(got[ah][1].view(np.uint64), got[0][1].view(np.uint64))
        for k in got[0][2]:
            assert np.array_equal(got[ah][2][k], got[0][2][k]), k
        assert np.array_equal(got[ah][3], got[0][3])
    if max_new == 2:
        assert (got[1][2]["status"] == -6).any()                 # some chains ran out of room for new segments
    orc = O.MvChains(mv, 5, seed=6, chain_offset=2, max_new_segments=max_new, split_prob=split_prob)
    of, oa = orc.run(sum(runs))
    assert np.array_equal(got[1][0][:5], of) and np.array_equal(got[1][1][:5].view(np.uint64), oa.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_gpu_growth_ahead_with_many_handles():
    """Twelve mcmc_mv handles in one process, each with its main and its growth stream -- six times as many streams as
    the device has hardware queues: every handle's runs finish promptly and give the chains a single handle gives.
    (With a low-priority growth stream, five handles in a process stalled for 90 s at the first run:
    tools/mv_streams_probe.py, which also puts two dozen torch streams beside them.)"""
    import time
    from mcmc_amd import host
    mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
    handles = []
    try:
        for k in range(12):
            m = host.Multiview(device=0)
            _load(m, mv)
            m.vertex_moments()
            handles.append(m)
        want = None
        for rep in range(2):
            for k, m in enumerate(handles):
                m.chains_init(300, seed=21, max_new_segments=8, split_prob=0.2)
                m.chains_set_trace(5)
                t0 = time.time()
                m.chains_run(5)
                assert time.time() - t0 < 20.0
                f, a = m.trace()
                if want is None:
                    want = (f.copy(), a.copy())
                assert np.array_equal(f, want[0]) and np.array_equal(a.view(np.uint64), want[1].view(np.uint64))
        orc = O.MvChains(mv, 4, seed=21, max_new_segments=8, split_prob=0.2)
        of, oa = orc.run(5)
        assert np.array_equal(want[0][:4], of) and np.array_equal(want[1][:4].view(np.uint64), oa.view(np.uint64))
    finally:
        for m in handles:
            m.close()


@pytest.mark.gpu
def test_gpu_chain_groups_change_nothing(gpu_mv):
    """Option "groups": the chains in G groups, each on a stream of its own with its share of the exchange buffers
    (automatic from 1024 chains on).  Chains do not interact, so every G gives the same chains; 200 chains in three
    unequal groups, against one group and, for the first chains, the oracle."""
    mv = make_multiview(n_views=2, width=48, height=36, n_planes=3, n_superpixels=150, seed=11)
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    n, steps = 200, 7
    got = {}
    for G in (1, 3, 2):
        gpu_mv.set_option("groups", G)
        gpu_mv.chains_init(n, seed=5, chain_offset=9, max_new_segments=16, split_prob=0.2)
        gpu_mv.chains_set_trace(steps)
        gpu_mv.chains_run(steps)
        f, a = gpu_mv.trace()
        gpu_mv.chains_run(steps)                                  # and a second launch on the same buffers
        s = gpu_mv.scalars()
        got[G] = (f.copy(), a.copy(), {k: v.copy() for k, v in s.items()}, gpu_mv.labels(1).copy())
    gpu_mv.set_option("groups", 0)
    for G in (3, 2):
        assert np.array_equal(got[G][0], got[1][0]) and np.array_equal(got[G][1].view(np.uint64), got[1][1].view(np.uint64))
        for k in got[1][2]:
            assert np.array_equal(got[G][2][k], got[1][2][k]), (G, k)
        assert np.array_equal(got[G][3], got[1][3])
    orc = O.MvChains(mv, 4, seed=5, chain_offset=9, max_new_segments=16, split_prob=0.2)
    of, oa = orc.run(steps)
    assert np.array_equal(got[3][0][:4], of) and np.array_equal(got[3][1][:4].view(np.uint64), oa.view(np.uint64))


@pytest.mark.gpu
def test_gpu_chains_capacity_overflow(gpu_mv, mv_small):
    _load(gpu_mv, mv_small)
    gpu_mv.chains_init(2, seed=1, max_new_segments=1, split_prob=1.0)
    gpu_mv.chains_set_trace(5)
    gpu_mv.chains_run(5)
    f, _ = gpu_mv.trace()
    assert (f[:, 0] & 2).all() and (f[:, 1:] == 0xFF).all()
    assert (gpu_mv.scalars()["status"] == -6).all()


@pytest.mark.gpu
def test_gpu_chains_c4_size_bit_exact(gpu_mv):
    # BASELINE configs[3] at full size (3 views x 640x480, ~500 superpixels each), a few chains
    mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
    _load(gpu_mv, mv)
    n, steps = 3, 12
    gpu_mv.chains_init(n, seed=2, max_new_segments=32)
    gpu_mv.chains_set_trace(steps)
    orc = O.MvChains(mv, n, seed=2, max_new_segments=32)
    assert np.array_equal(gpu_mv.labels(0), orc.labels(0))
    gpu_mv.chains_run(steps)
    of, oa = orc.run(steps)
    gf, ga = gpu_mv.trace()
    assert np.array_equal(gf, of) and np.array_equal(ga.view(np.uint64), oa.view(np.uint64))
    assert np.array_equal(gpu_mv.scalars()["trace_hash"], orc.scalars()["trace_hash"])
    # size-independent property: a split proposal leaves SB2 == SA2, so its ratio is P(SB1)/P(SA1) and the
    # vertex sets stay a partition
    own = gpu_mv.labels(1)
    assert (own.max(axis=1) < gpu_mv.scalars()["n_segments"]).all()


@pytest.mark.gpu
def test_gpu_chains_c4_all_chains_properties(gpu_mv):
    """BASELINE configs[3] at the size bench.py runs: 1024 chains x 20 steps.  The oracle is too slow for all of them, so
    every chain is checked through what the loop must preserve (MultiviewSegmentation.cpp:85-110, 207-257;
    PatternedMatcher.cpp:58-130) and the first three chains against the oracle bit for bit."""
    mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
    _load(gpu_mv, mv)
    n, steps = 1024, 20
    gpu_mv.chains_init(n, seed=4, max_new_segments=4 * steps)
    gpu_mv.chains_set_trace(steps)
    cur0 = gpu_mv.labels(0).copy()
    own0 = gpu_mv.labels(1).copy()
    s0 = gpu_mv.scalars()
    assert np.array_equal(cur0, own0)                             # before the first step the two labelings agree
    assert (s0["n_segments"] == cur0.max(axis=1).astype(np.int64) + 1).all()
    gpu_mv.chains_run(steps)
    f, a = gpu_mv.trace()
    s = gpu_mv.scalars()
    assert f.shape == (n, steps) and (s["status"] == 0).all() and (f != 0xFF).all()
    # the counters are the trace, summed: accepted, split and skipped proposals
    assert (s["n_accept"].astype(np.int64) == (f & 1).sum(axis=1)).all()
    assert (s["n_split"].astype(np.int64) == ((f >> 1) & 1).sum(axis=1)).all()
    assert (s["n_skip"].astype(np.int64) == ((f >> 2) & 1).sum(axis=1)).all()
    assert (a[(f & 4) != 0] == 0).all()                           # a skipped proposal has no ratio
    assert (np.isnan(a) | (a >= 0)).all()
    # the vertex sets are a partition: every vertex in exactly one segment, ids below the count; every split appended
    # exactly one segment and it is not empty; currentSegmentID never changes
    own = gpu_mv.labels(1)
    assert np.array_equal(gpu_mv.labels(0), cur0)
    assert (s["n_segments"].astype(np.int64) == s0["n_segments"].astype(np.int64) + s["n_split"].astype(np.int64)).all()
    assert (own.max(axis=1) < s["n_segments"]).all()
    for k in range(0, n, 97):
        used = np.unique(own[k])
        new = used[used >= s0["n_segments"][k]]
        # a later split may empty an earlier new segment's vertices only by moving them to a newer one
        assert new.size <= s["n_split"][k] and (new < s["n_segments"][k]).all()
        assert np.bincount(own[k], minlength=int(s["n_segments"][k])).sum() == mv.n_vertices
    # the chains differ from one another and the run is deterministic in (seed, chain id)
    assert np.unique(s["trace_hash"]).size > n // 2
    orc = O.MvChains(mv, 3, seed=4, max_new_segments=4 * steps)
    of, oa = orc.run(steps)
    assert np.array_equal(f[:3], of) and np.array_equal(a[:3].view(np.uint64), oa.view(np.uint64))
    assert np.array_equal(s["trace_hash"][:3], orc.scalars()["trace_hash"])
    assert np.array_equal(own[:3], orc.labels(1))


# ------------------------------------------------------------------ world graph
def _world_inputs(seed=3, n0=30, n1=25, n2=20, dim=7):
    rng = np.random.default_rng(seed)
    V = n0 + n1 + n2
    base = rng.uniform(-1, 1, (6, dim)).astype(np.float32)                 # six underlying planes
    theta = (base[rng.integers(0, 6, V)] + rng.normal(0, 0.03, (V, dim))).astype(np.float32)
    # in-view neighbour graphs: a ring plus a few chords inside each scan
    pairs = set()
    for lo, n in ((0, n0), (n0, n1), (n0 + n1, n2)):
        for k in range(n):
            pairs.add((lo + k, lo + (k + 1) % n)); pairs.add((lo + k, lo + (k + 5) % n))
    nb = [set() for _ in range(V)]
    for a, b in pairs:
        if a != b:
            nb[a].add(b); nb[b].add(a)
    rp = np.zeros(V + 1, np.uint32); col = []
    for i in range(V):
        col += sorted(nb[i]); rp[i + 1] = len(col)
    return theta, rp, np.array(col, np.uint32), n0


def test_world_graph_against_numpy():
    theta, irp, icol, n0 = _world_inputs()
    V = theta.shape[0]
    rp, col, prob = O.mv_world_graph(theta, DEFAULT_WEIGHTS, n0, irp, icol, in_thr=0.01, thr=0.2)
    P = pmerge_np(theta[:, None, :], theta[None, :, :])                     # V x V
    inview = np.zeros((V, V), bool)
    src = np.repeat(np.arange(V), np.diff(irp.astype(np.int64)))
    inview[src, icol] = True
    later = np.maximum.outer(np.arange(V), np.arange(V)) >= n0             # a vertex of a later scan is involved
    want = ((later & (P > 0.2)) | (inview & (P > 0.01))) & ~np.eye(V, dtype=bool)
    got = np.zeros((V, V), bool)
    gsrc = np.repeat(np.arange(V), np.diff(rp.astype(np.int64)))
    got[gsrc, col] = True
    margin = np.minimum(np.abs(P - 0.2), np.abs(P - 0.01)) > 1e-9          # away from the thresholds (libm vs det_exp)
    assert np.array_equal(got[margin], want[margin])
    assert np.array_equal(got, got.T)                                       # symmetric, like boost's undirectedS
    assert np.allclose(prob, P[gsrc, col], rtol=1e-12)
    for i in range(V):
        r = col[rp[i]:rp[i + 1]]
        assert (np.diff(r.astype(np.int64)) > 0).all() and i not in r       # ascending, no self loops (:166)
    # first-scan vertices are joined to each other only through in-view edges (getPlanarAdjacencyGraph)
    assert not (got[:n0, :n0] & ~inview[:n0, :n0]).any()
    # later scans are joined to everything, their own scan included (mergeNewScanGraph iterates `original`)
    assert (got[n0:, n0:] & ~inview[n0:, n0:]).any()


@pytest.mark.gpu
def test_gpu_world_graph_bit_exact(gpu_mv):
    for seed, n0 in ((3, 30), (4, 0), (5, 75)):
        theta, irp, icol, _ = _world_inputs(seed)
        want = O.mv_world_graph(theta, DEFAULT_WEIGHTS, n0, irp, icol, in_thr=0.01, thr=0.2)
        got = gpu_mv.build_world_graph(theta, DEFAULT_WEIGHTS, n0, irp, icol, 0.01, 0.2, 8.0)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(got[2].view(np.uint64), want[2].view(np.uint64))
    theta, irp, icol, n0 = _world_inputs(6)
    want = O.mv_world_graph(theta, DEFAULT_WEIGHTS, n0)                     # no in-view graph
    got = gpu_mv.build_world_graph(theta, DEFAULT_WEIGHTS, n0)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.gpu
def test_gpu_split_step_equals_fused_and_oracle(gpu_mv, mv_small):
    """The three-kernel iteration (residuals of all chains evaluated vertex by vertex) gives the bits of
    the fused kernel and of the oracle, capacity freeze included."""
    mv = mv_small
    _load(gpu_mv, mv)
    gpu_mv.vertex_moments()
    res = {}
    for mode in (0, 1):
        gpu_mv.set_option("split_step", mode)
        gpu_mv.chains_init(6, seed=7, max_new_segments=3, split_prob=0.3)
        gpu_mv.chains_set_trace(40)
        gpu_mv.chains_run(25); f1, a1 = gpu_mv.trace()
        gpu_mv.chains_run(15); f2, a2 = gpu_mv.trace()
        res[mode] = (np.concatenate([f1, f2], 1), np.concatenate([a1, a2], 1), gpu_mv.scalars(), gpu_mv.labels(1))
    gpu_mv.set_option("split_step", -1)
    ch = O.MvChains(mv, 6, seed=7, max_new_segments=3, split_prob=0.3)
    f, a = ch.run(40)
    s = ch.scalars()
    for mode in (0, 1):
        gf, ga, gs, gl = res[mode]
        assert np.array_equal(gf, f) and np.array_equal(ga.view(np.uint64), a.view(np.uint64)), mode
        for k in ("n_accept", "n_split", "n_skip", "trace_hash", "n_segments", "status"):
            assert np.array_equal(gs[k], s[k]), (mode, k)
        assert np.array_equal(gl, ch.labels(1)), mode
    assert (s["status"] == -6).any()          # the case covers a frozen chain
