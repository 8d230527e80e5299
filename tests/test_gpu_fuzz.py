"""Randomised differential test of the HIP path against the oracle (Path A): frame shapes (widths that take the
strip kernel, its ragged form and the generic kernel), superpixel counts, noise levels, chain counts that do not
fill a workgroup, runs split into several launches.  Everything is compared bit for bit.  Fixed seeds: the
cases are the same on every run."""
import os

import numpy as np
import pytest

from mcmc_amd import host
from mcmc_amd.scene import make_scene
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _cases():
    # FUZZ_N / FUZZ_SEED widen the sweep for a one-off run; the defaults are the committed cases
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", 20240611)))
    out = []
    for k in range(int(os.environ.get("FUZZ_N", 14))):
        W = int(rng.choice([16, 24, 40, 56, 64, 72, 96, 100, 120, 128, 160, 37, 250, 320]))
        H = int(rng.integers(12, 130))
        S = int(rng.integers(3, max(4, min(160, W * H // 60))))
        out.append(dict(W=W, H=H, S=S, K=int(rng.integers(1, 5)), seed=int(rng.integers(1, 10_000)),
                        noise=float(rng.choice([0.3, 1.0, 3.0])), chains=int(rng.integers(1, 23)),
                        burn=int(rng.integers(0, 60)), parts=[int(x) for x in rng.integers(1, 90, size=int(rng.integers(1, 4)))]))
    return out


@pytest.mark.parametrize("c", _cases(), ids=lambda c: f"{c['W']}x{c['H']}-S{c['S']}-n{c['chains']}")
def test_random_frame_and_chains_bit_exact(c, gpu_ctx_factory):
    sc = make_scene(width=c["W"], height=c["H"], n_planes=c["K"], n_superpixels=c["S"], seed=c["seed"], noise=c["noise"])
    S = sc.n_superpixels
    if S < 2 or sc.col.size == 0:
        pytest.skip("degenerate scene")
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height)
    ctx.frames_alloc(1, S, sc.col.size)
    host.load_scene(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    mom = O.moments(sc.disp, sc.label, S)
    assert np.array_equal(ctx.moments(0), mom)
    fit, valid, lerr = O.local_fit(mom, sc.width, sc.height)
    gfit, gvalid, glerr = ctx.local_fit(0)
    assert np.array_equal(bits(gfit), bits(fit)) and np.array_equal(gvalid, valid) and np.array_equal(bits(glerr), bits(lerr))
    st, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    assert st == 0 and np.array_equal(bits(ctx.edge_probs(0)), bits(q))
    n = c["chains"]
    if n % 8 == 0:
        ctx.set_option("k3_dense", 1)          # eight chains per workgroup, records behind L2
    ctx.chains_init(n, seed=c["seed"], chain_offset=3)
    orc = O.Chains(fit, valid, sc.row_ptr, sc.col, q, n, c["seed"], 3)
    burn = c["burn"]
    for part in c["parts"]:
        ctx.chains_set_trace(burn + part)
        ctx.chains_run(burn, part)
        st, tr = orc.run(burn, part, trace=True)
        assert st == 0
        assert np.array_equal(ctx.trace(), tr)
        burn = 0
    gs, os_ = ctx.scalars(), orc.scalars()
    for k in ("state_hash", "trace_hash", "n_accept", "n_case", "status", "top_count", "top_hash", "overflow"):
        assert np.array_equal(gs[k], os_[k]), k
    assert np.array_equal(bits(gs["energy"]), bits(os_["energy"]))
    assert np.array_equal(bits(gs["best_energy"]), bits(os_["best_energy"]))
    for which in (0, 1, 2):
        assert np.array_equal(ctx.labels(which, True)[:, :S], orc.labels(which, True)), which
    assert np.array_equal(bits(ctx.recompute_energy()), bits(orc.recompute_energy()))
    assert ctx.visits() == orc.visits()
    assert np.array_equal(ctx.marginals(0)[:S], orc.marginals())


def _mv_cases():
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", 77)))
    out = []
    for k in range(int(os.environ.get("FUZZ_N", 8))):
        out.append(dict(views=int(rng.integers(1, 4)), W=int(rng.choice([64, 96, 160])), H=int(rng.choice([48, 80, 120])),
                        K=int(rng.integers(1, 5)), S=int(rng.integers(6, 60)), seed=int(rng.integers(1, 10_000)),
                        chains=int(rng.choice([1, 3, 17, 33, 40, 70])), steps=int(rng.integers(1, 40)),
                        split_prob=float(rng.choice([0.0, 0.1, 0.5, 1.0])), max_new=int(rng.choice([0, 2, 64])),
                        split_step=int(rng.choice([-1, 0, 1]))))
    return out


@pytest.mark.parametrize("c", _mv_cases(), ids=lambda c: f"v{c['views']}-S{c['S']}-n{c['chains']}-p{c['split_prob']}-m{c['split_step']}")
def test_random_multiview_chains_bit_exact(c):
    """Path B: fused and split iteration, chain counts on both sides of the automatic switch, split
    probabilities 0 and 1, segment tables that fill up (max_new_segments 0 / 2)."""
    from mcmc_amd.scene import make_multiview
    mv = make_multiview(n_views=c["views"], width=c["W"], height=c["H"], n_planes=c["K"], n_superpixels=c["S"], seed=c["seed"])
    if mv.n_vertices < 2 or mv.col.size == 0:
        pytest.skip("degenerate scene")
    gm = host.Multiview(device=0)
    try:
        gm.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
        gm.set_option("split_step", c["split_step"])
        n, steps = c["chains"], c["steps"]
        gm.chains_init(n, seed=c["seed"], chain_offset=5, max_new_segments=c["max_new"], split_prob=c["split_prob"])
        gm.chains_set_trace(steps)
        orc = O.MvChains(mv, n, seed=c["seed"], chain_offset=5, max_new_segments=c["max_new"], split_prob=c["split_prob"])
        assert np.array_equal(gm.labels(0), orc.labels(0))
        for rep in range(2):
            gm.chains_run(steps)
            of, oa = orc.run(steps)
            gf, ga = gm.trace()
            assert np.array_equal(gf, of), rep
            assert np.array_equal(ga.view(np.uint64), oa.view(np.uint64)), rep
            gs, os_ = gm.scalars(), orc.scalars()
            for key in gs:
                assert np.array_equal(gs[key], os_[key]), (rep, key)
            assert np.array_equal(gm.labels(1), orc.labels(1)), rep
        ns = gm.scalars()["n_segments"]
        for k in range(min(n, 4)):
            assert np.array_equal(gm.planes(k, ns[k]).view(np.uint64), orc.planes(k, ns[k]).view(np.uint64))
    finally:
        gm.close()


def _fe_cases():
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", 991)))
    out = []
    for k in range(int(os.environ.get("FUZZ_N", 8))):
        w = int(rng.choice([1, 3, 5, 7, 9, 11, 6]))
        Wc, Hc = int(rng.choice([8, 17, 40, 56, 90, 128])), int(rng.integers(6, 70))
        if w >= min(Wc, Hc):          # the boundary rejects windows that do not fit the frame
            w = 3
        out.append(dict(W=Wc, H=Hc, seed=int(rng.integers(0, 1000)),
                        zero_frac=float(rng.choice([0.0, 0.1, 0.5])),
                        kw=dict(w=w, sstd=float(rng.choice([1.5, 2.5, 5.0])), dstd=(80.0 if w % 2 == 0 else float(rng.choice([1.0, 3.0, 20.0]))),
                                c=int(rng.choice([5, 40, 300, 2000])), cf=float(rng.choice([0.0, 0.5, 1.0])),
                                df=float(rng.choice([0.5, 1.0, 3.0])), distf=float(rng.choice([0.0, 1.0, 2.0])))))
    return out


@pytest.mark.parametrize("c", _fe_cases(), ids=lambda c: f"{c['W']}x{c['H']}-w{c['kw']['w']}-c{c['kw']['c']}")
def test_random_front_end_bit_exact(c):
    """Over-segmentation front end: frame shapes from 8 pixels wide, windows 1..11 (even ones included), holes in
    the disparity map, thresholds from many tiny superpixels to a few large ones."""
    from tests.test_frontend import _check_against_oracle
    rng = np.random.default_rng(c["seed"])
    H, W = c["H"], c["W"]
    yy, xx = np.mgrid[0:H, 0:W]
    disp = (250 + 0.4 * xx - 0.2 * yy + 30 * (xx > W // 2) + rng.integers(-2, 3, (H, W))).astype(np.int16)
    disp[rng.random((H, W)) < c["zero_frac"]] = 0
    col = np.clip(100 + 70 * (yy > H // 2)[:, :, None] + rng.normal(0, 10, (H, W, 3)), 0, 255).astype(np.uint8)
    o = host.OverSegmenter(W, H, device=0)
    try:
        _check_against_oracle(o, disp, col, **c["kw"])
    finally:
        o.close()
