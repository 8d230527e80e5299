"""The oracle's sampler against the exact transition law of the reference's proposal / accept code.

tests/literal_sampler.py transcribes Segmentation::createComponent / selectSegment, Problem::propose and
FastMetropolisHastings::iterate as a probability tree (no code shared with oracle/mcmc_oracle.c) and yields the
exact transition matrix P over the partitions of a 3- or 4-superpixel frame.  The oracle runs 40 000
independent chains for t steps from the reference's initial state (one segment, Problem.cpp:447-473); the
empirical distribution of their states must agree with e0 P^t within sampling error for t = 1, 2, 5, 40.
This pins the sampler half of the restatement (draw order aside, which is the build's own convention)
the way literal_ref.py pins the energy half.  SURVEY.md section 8c(4) asked for a detailed-balance check
against the enumerated posterior: the reference's chain does not have the posterior as its stationary law
(its proposal ratio omits the reverse selectSegment term), see test_stationary_law_is_not_the_posterior.
"""
import numpy as np
import pytest

from mcmc_amd.scene import build_adjacency
from tests import literal_sampler as LS
from tests import oracle_lib as O

N_CHAINS = 40000


def _frame(kind):
    rng = np.random.default_rng(3)
    if kind == "path3":      # 0 - 1 - 2
        H, W, S = 12, 36, 3
        label = (np.arange(W) // 12)[None, :].repeat(H, 0).astype(np.uint16)
        offs = [0.0, 0.2, 1.0]
    else:                    # 2 x 2 blocks: complete graph under 8-neighbour contact
        H, W, S = 16, 16, 4
        yy, xx = np.mgrid[0:H, 0:W]
        label = ((yy // 8) * 2 + (xx // 8)).astype(np.uint16)
        offs = [0.0, 0.3, 0.8, 0.1]
    yy, xx = np.mgrid[0:H, 0:W]
    disp = np.clip(np.rint(400 + 0.02 * xx - 0.01 * yy + np.array(offs)[label] + rng.normal(0, 2.0, (H, W))),
                   1, 2047).astype(np.int16)
    return disp, label, S


def _setup(kind):
    disp, label, S = _frame(kind)
    H, W = disp.shape
    fit, valid, _ = O.local_fit(O.moments(disp, label, S), W, H)
    assert valid.all()
    rp, col = build_adjacency(label, S)
    st, q = O.edge_probs(fit, valid, rp, col)
    assert st == 0
    nbrs = [[(int(col[e]), float(q[e])) for e in range(rp[i], rp[i + 1])] for i in range(S)]
    cache = {}

    def probL(m):
        m = frozenset(m)
        if m not in cache:
            st, p, _, _ = O.segment_probability(fit, sorted(m))
            assert st == 0
            cache[m] = p
        return cache[m]

    states, index, P = LS.transition_matrix(S, nbrs, probL)
    assert np.allclose(P.sum(1), 1.0, atol=1e-12)
    return dict(S=S, fit=fit, valid=valid, rp=rp, col=col, q=q, states=states, index=index, P=P, probL=probL)


@pytest.fixture(scope="module", params=["path3", "grid4"])
def law(request):
    return _setup(request.param)


@pytest.mark.parametrize("t", [1, 2, 5, 40])
def test_k_step_distribution_matches_exact_law(law, t):
    ch = O.Chains(law["fit"], law["valid"], law["rp"], law["col"], law["q"], N_CHAINS, seed=11 + t)
    st, _ = ch.run(0, t)
    assert st == 0
    emp = np.zeros(len(law["states"]))
    for row in ch.labels(0, True):
        emp[law["index"][tuple(int(x) for x in row)]] += 1
    emp /= N_CHAINS
    e0 = np.zeros(len(law["states"]))
    e0[law["index"][tuple([0] * law["S"])]] = 1.0
    want = e0 @ np.linalg.matrix_power(law["P"], t)
    tol = 5.0 * np.sqrt(want * (1.0 - want) / N_CHAINS) + 5e-4       # five sigma of a binomial share
    assert (np.abs(emp - want) <= tol).all(), (emp, want)


def test_stationary_law_is_not_the_posterior(law):
    """Documents a property of the reference, not of the build: the chain's stationary distribution differs
    from exp(sum of segment log-probabilities) (Segmentation::computeProbability)."""
    P = law["P"]
    w, v = np.linalg.eig(P.T)
    pi = np.real(v[:, np.argmin(np.abs(w - 1.0))])
    pi /= pi.sum()
    assert (pi > -1e-12).all() and np.allclose(pi @ P, pi, atol=1e-12)
    post = np.array([sum(law["probL"]([i for i in range(law["S"]) if s[i] == l]) for l in set(s)) for s in law["states"]])
    post = np.exp(post - post.max())
    post /= post.sum()
    assert np.abs(pi - post).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["path3", "grid4"])
def test_gpu_k_step_distribution_matches_exact_law(kind, gpu_ctx_factory):
    """The product path (K1, K1b, K2, K3 through the C-ABI) against the transcription directly."""
    law = _setup(kind)
    disp, label, S = _frame(kind)
    H, W = disp.shape
    ctx = gpu_ctx_factory(width=W, height=H)
    ctx.frames_alloc(1, S, law["col"].size)
    ctx.frame_upload(0, disp, label, S)
    ctx.frame_set_graph(0, law["rp"], law["col"])
    ctx.frames_moments(); ctx.frames_edge_probs()
    e0 = np.zeros(len(law["states"]))
    e0[law["index"][tuple([0] * S)]] = 1.0
    for t in (1, 5, 40):
        ctx.chains_init(N_CHAINS, seed=100 + t)
        ctx.chains_run(0, t)
        emp = np.zeros(len(law["states"]))
        for row in ctx.labels(0, True)[:, :S]:
            emp[law["index"][tuple(int(x) for x in row)]] += 1
        emp /= N_CHAINS
        want = e0 @ np.linalg.matrix_power(law["P"], t)
        tol = 5.0 * np.sqrt(want * (1.0 - want) / N_CHAINS) + 5e-4
        assert (np.abs(emp - want) <= tol).all(), (t, emp, want)
