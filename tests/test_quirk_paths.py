"""The reference's error and quirk paths, GPU against the oracle (`NaN -> 0.75` at
Graph.cpp:134, "log probability is inf" at Graph.cpp:136-137, "Target ratio is nan" at Problem.cpp:216-217,
"edge probability is nan" at Problem.cpp:443).  They are reached here the way a caller can reach them: by data (a
superpixel whose pixels lie on one diagonal line passes the A1 filter -- isLinear only looks for one row or one
column, OverSegmentation.cpp:435-455 -- but has a singular normal matrix) and by a variance of +inf or NaN."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
W, H = 64, 48


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def same_f64(a, b):
    """bit-equal, except that any NaN equals any NaN (x86 and gfx950 differ in the sign of a generated NaN)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(bits(a)[~na], bits(b)[~nb])


def _line_frame(seed=0):
    """Superpixels 0 and 1: 11 and 10 pixels on one diagonal through the frame centre (singular fit); 2: a diagonal
    elsewhere (rank deficient up to rounding); 3..6: ordinary blocks on a noisy plane.  Everyone neighbours everyone."""
    rng = np.random.default_rng(seed)
    v, u = np.mgrid[0:H, 0:W]
    disp = np.clip(np.rint(300 + 0.5 * u - 0.3 * v + rng.normal(0, 1.0, (H, W))), 1, 2047).astype(np.int16)
    label = np.full((H, W), 0xFFFF, np.uint16)
    cx, cy = W // 2, H // 2
    for t in range(11):
        label[cy - 5 + t, cx - 5 + t] = 0
    for t in range(10):
        label[cy + 6 + t, cx + 6 + t] = 1
    for t in range(12):
        label[20 + t, 3 + t] = 2
    label[2:8, 2:10] = 3; label[2:8, 12:20] = 4; label[2:8, 22:30] = 5; label[10:16, 40:50] = 6
    S = 7
    rp = np.zeros(S + 1, np.uint32); col = []
    for s in range(S):
        col += [t for t in range(S) if t != s]; rp[s + 1] = len(col)
    return disp, label, S, rp, np.array(col, np.uint32)


def _setup(gpu_ctx_factory, var, seed=0):
    disp, label, S, rp, col = _line_frame(seed)
    ctx = gpu_ctx_factory(width=W, height=H, disparity_variance=var)
    ctx.frames_alloc(1, S, col.size)
    ctx.frame_upload(0, disp, label, S)
    ctx.frame_set_graph(0, rp, col)
    ctx.frames_moments(); ctx.frames_edge_probs()
    fit, valid, lerr = O.local_fit(O.moments(disp, label, S), W, H, var)
    return ctx, (disp, label, S, rp, col), (fit, valid, lerr)


def test_nan_probability_becomes_075_by_data(gpu_ctx_factory):
    # Graph.cpp:134 with the reference's own variance: the diagonal superpixels' fits are 0/0
    ctx, (disp, label, S, rp, col), (fit, valid, _) = _setup(gpu_ctx_factory, 7.0)
    gfit, gvalid, _ = ctx.local_fit(0)
    assert gvalid.tolist() == valid.tolist() == [1] * S
    assert same_f64(gfit, fit)
    for members in ([0], [1], [0, 1], [2], [0, 3], [3, 4], list(range(S))):
        p, th, err = ctx.segment_probability(0, members)
        st, op, oth, oerr = O.segment_probability(fit, members)
        assert st == 0 and same_f64([p, err], [op, oerr]) and same_f64(th, oth), members
    p0, _, e0 = ctx.segment_probability(0, [0])
    assert p0 == 0.75 and np.isnan(e0)                       # the literal of Graph.cpp:134
    st, q = O.edge_probs(fit, valid, rp, col)
    assert st == 0 and np.array_equal(bits(ctx.edge_probs(0)), bits(q))
    n = 8
    ctx.chains_init(n, seed=3); ctx.chains_set_trace(500)
    ctx.chains_run(100, 400)
    orc = O.Chains(fit, valid, rp, col, q, n, 3)
    st, tr = orc.run(100, 400, trace=True)
    assert np.array_equal(ctx.trace(), tr)
    g, o = ctx.scalars(), orc.scalars()
    assert np.array_equal(g["status"], o["status"])
    live = g["status"] == 0
    assert live.any()
    assert np.array_equal(g["trace_hash"], o["trace_hash"]) and same_f64(g["energy"], o["energy"])
    assert np.array_equal(ctx.labels()[:, :S], orc.labels())
    # a chain that meets a NaN ratio (a 0.75 segment against a NaN-free one cannot give one; an inf can) stays frozen
    assert set(np.unique(g["status"]).tolist()) <= {0, -4, -5}


def test_nan_variance_every_segment_at_075(gpu_ctx_factory):
    # variance = NaN: every error term is NaN, so every segment's probabilityL is the 0.75 of Graph.cpp:134; the
    # chains run (targetRatio is a difference of 0.75s) and are the oracle's bit for bit
    ctx, (disp, label, S, rp, col), (fit, valid, _) = _setup(gpu_ctx_factory, float("nan"))
    for members in ([3], [3, 4], list(range(S))):
        assert ctx.segment_probability(0, members)[0] == 0.75
    st, q = O.edge_probs(fit, valid, rp, col, float("nan"))
    gq = ctx.edge_probs(0)
    assert st == 0 and np.array_equal(bits(gq), bits(q)) and np.all(gq == gq[0])   # 1 / (1 + e^0.75) everywhere
    n = 6
    ctx.chains_init(n, seed=11); ctx.chains_set_trace(300)
    ctx.chains_run(50, 250)
    orc = O.Chains(fit, valid, rp, col, q, n, 11, var=float("nan"))
    st, tr = orc.run(50, 250, trace=True)
    assert st == 0 and np.array_equal(ctx.trace(), tr) and (tr != 0xFF).all()
    g, o = ctx.scalars(), orc.scalars()
    assert (g["status"] == 0).all() and np.array_equal(g["trace_hash"], o["trace_hash"])
    assert np.array_equal(bits(g["energy"]), bits(o["energy"]))
    assert np.array_equal(ctx.labels()[:, :S], orc.labels())
    assert ctx.visits() == orc.visits()


def test_inf_variance_inf_energy_and_nan_edge_probability(gpu_ctx_factory):
    from mcmc_amd import host
    inf = float("inf")
    ctx, (disp, label, S, rp, col), (fit, valid, _) = _setup(gpu_ctx_factory, inf)
    # Graph.cpp:136-137: log sqrt det(2 pi Sigma) = +inf
    with pytest.raises(host.McmcError) as e:
        ctx.segment_probability(0, [3, 4])
    assert e.value.code == -5
    assert O.segment_probability(fit, [3, 4], inf)[0] == -5
    # Problem.cpp:443: inf - inf in the edge probability
    st, q = O.edge_probs(fit, valid, rp, col, inf)
    assert st == -4 and np.isnan(q[rp[3]:]).any()          # every edge between two ordinary superpixels
    gq = np.zeros(col.size)
    rc = ctx.L.mcmc_frame_get_edge_probs(ctx.h, 0, gq.ctypes.data_as(__import__("ctypes").c_void_p))
    assert rc == -4 and same_f64(gq, q)                      # the values are handed over with the error code
    with pytest.raises(host.McmcError) as e:
        ctx.edge_probs(0)
    assert e.value.code == -4
    # the chains: the initial segment holds the diagonal superpixels too, so its probabilityL is NaN -> 0.75 and
    # initializeState succeeds; the first proposal that builds a segment of ordinary superpixels alone meets
    # +inf: the reference throws out of run(), here the chain freezes with MCMC_E_INF_ENERGY (or MCMC_E_NAN_RATIO
    # when the ratio is inf - inf, Problem.cpp:216-217) and its trace reads 0xFF from there on
    n = 16
    ctx.chains_init(n, seed=2); ctx.chains_set_trace(60)
    ctx.chains_run(20, 40)
    gtr = ctx.trace()
    orc = O.Chains(fit, valid, rp, col, q, n, 2, var=inf)
    st, tr = orc.run(20, 40, trace=True)
    assert st in (-4, -5) and np.array_equal(gtr, tr)
    g, o = ctx.scalars(), orc.scalars()
    assert np.array_equal(g["status"], o["status"]) and set(g["status"].tolist()) <= {0, -4, -5}
    frozen = g["status"] != 0
    assert frozen.any() and (gtr[frozen, -1] == 0xFF).all()
    for k in np.flatnonzero(frozen):                         # frozen for good: 0xFF from its first error to the end
        first = int(np.argmax(gtr[k] == 0xFF))
        assert (gtr[k, first:] == 0xFF).all() and (gtr[k, :first] != 0xFF).all()
    for k in ("n_accept", "n_case", "trace_hash", "state_hash"):
        assert np.array_equal(g[k], o[k]), k
    assert np.array_equal(ctx.labels()[:, :S], orc.labels())
    ctx.chains_run(0, 10)                                    # and a further run leaves a frozen chain where it is
    g2 = ctx.scalars()
    assert np.array_equal(g2["n_case"][frozen], g["n_case"][frozen]) and np.array_equal(g2["status"][frozen], g["status"][frozen])
    # the frame's flag does not outlive the data that raised it: the same frame without edges is clean again
    ctx.frame_upload(0, disp, label, S)
    ctx.frame_set_graph(0, np.zeros(S + 1, np.uint32), np.zeros(0, np.uint32))
    ctx.frames_moments(); ctx.frames_edge_probs()
    assert ctx.edge_probs(0).size == 0
