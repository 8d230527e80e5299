"""Segment energies at BASELINE config size against the literal-reference goldens.

tests/golden/config_frames.npz (tests/golden/make_golden_at_size.py) holds what
tests/literal_ref.py -- the numpy emulation of the reference's gtsam call sequence in raw
pixel coordinates (include/MyMath.h:120-154,189-275; src/libraries/Graph.cpp:116-138) --
computes for the C2 scene (640x480, 594 superpixels) and a C3-size scene (1920x1080, 4554
superpixels): the whole frame as ONE segment (the state every chain starts from,
Problem.cpp:447-473), one segment per ground-truth plane, 20 random member sets, every
singleton.  The build computes a segment's error from sums, `(Sbb - m'M^-1 m) - sum ssr_i`,
which cancels ~1e11 against ~1e11 on the big sets; the bar is north_star's 1e-6 relative
on probabilityL and on the error term, theta within 1e-6 of the plane's scale.
"""
import os

import numpy as np
import pytest

from tests import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 1e-6


@pytest.fixture(scope="module")
def big():
    return np.load(os.path.join(ROOT, "tests", "golden", "config_frames.npz"))


_scenes = {}


def scene_of(big, name):
    if name not in _scenes:
        from mcmc_amd.scene import make_scene
        from tests.golden.make_golden_at_size import checksum
        W, H, K, S, seed = (int(x) for x in big[name + "_args"])
        sc = make_scene(width=W, height=H, n_planes=K, n_superpixels=S, seed=seed)
        assert np.array_equal(checksum(sc), big[name + "_checksum"]), "scene generator drifted from the fixture"
        _scenes[name] = sc
    return _scenes[name]


def raw_theta(th3, W, H):
    """(a, b, c) of the centred parameterisation -> raw pixel coordinates."""
    th = np.array(th3, np.float64).copy()
    th[..., 2] -= th[..., 0] * (W // 2) + th[..., 1] * (H // 2)
    return th


def check_sets(big, name, W, H, seg_prob):
    """seg_prob(members) -> (probL, theta3 centred, err).  Returns the worst relative deviations seen."""
    sp, si = big[name + "_set_ptr"], big[name + "_set_idx"]
    worst_p = worst_e = worst_t = 0.0
    for k in range(len(sp) - 1):
        p, th3, err = seg_prob(si[sp[k]:sp[k + 1]])
        gp, ge, gt = big[name + "_set_probL"][k], big[name + "_set_err"][k], big[name + "_set_theta_raw"][k]
        worst_p = max(worst_p, abs(p - gp) / abs(gp))
        worst_e = max(worst_e, abs(err - ge) / max(abs(ge), 1.0))
        # theta: a, b relative to their own size is meaningless near 0; compare the predicted disparity at the
        # frame corners, which is what the plane is for
        c = np.array([[0, 0, 1], [W - 1, 0, 1], [0, H - 1, 1], [W - 1, H - 1, 1]], np.float64)
        worst_t = max(worst_t, float(np.max(np.abs(c @ raw_theta(th3, W, H) - c @ gt) / np.maximum(np.abs(c @ gt), 1.0))))
    assert worst_p <= REL, (name, "probabilityL", worst_p)
    assert worst_e <= REL, (name, "error term", worst_e)
    assert worst_t <= REL, (name, "plane", worst_t)
    return worst_p, worst_e, worst_t


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_oracle_vs_literal_reference_at_config_size(big, name):
    sc = scene_of(big, name)
    S, W, H = sc.n_superpixels, sc.width, sc.height
    fit, valid, lerr = O.local_fit(O.moments(sc.disp, sc.label, S), W, H)
    assert valid.all()
    assert np.allclose(raw_theta(fit[:, 9:12], W, H), big[name + "_theta_raw"], rtol=REL, atol=1e-7)
    assert np.allclose(lerr, big[name + "_local_err"], rtol=REL, atol=1e-9)

    def seg(members):
        st, p, th, e = O.segment_probability(fit, members)
        assert st == 0
        return p, th, e
    w = check_sets(big, name, W, H, seg)
    print(name, "oracle vs literal reference: worst rel probL %.2e err %.2e plane %.2e" % w)
    for i in range(0, S, max(1, S // 97)):          # singletons: error exactly 0, probabilityL the normaliser
        p, _, e = seg(np.array([i], np.uint32))
        assert e == 0.0 and abs(p - big[name + "_single_probL"][i]) <= REL * abs(big[name + "_single_probL"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c3"])
def test_gpu_vs_literal_reference_at_config_size(big, gpu_ctx_factory, name):
    from mcmc_amd import host
    sc = scene_of(big, name)
    S, W, H = sc.n_superpixels, sc.width, sc.height
    ctx = gpu_ctx_factory(width=W, height=H)
    ctx.frames_alloc(1, S, sc.col.size)
    host.load_scene(ctx, 0, sc)
    ctx.frames_moments()
    fit, valid, lerr = ctx.local_fit(0)
    assert valid.all()
    assert np.allclose(raw_theta(fit[:, 9:12], W, H), big[name + "_theta_raw"], rtol=REL, atol=1e-7)
    assert np.allclose(lerr, big[name + "_local_err"], rtol=REL, atol=1e-9)
    w = check_sets(big, name, W, H, lambda m: ctx.segment_probability(0, m))
    print(name, "GPU vs literal reference: worst rel probL %.2e err %.2e plane %.2e" % w)
    # and the GPU equals the oracle bit for bit on the same sets
    ofit, _, _ = O.local_fit(O.moments(sc.disp, sc.label, S), W, H)
    sp, si = big[name + "_set_ptr"], big[name + "_set_idx"]
    for k in range(len(sp) - 1):
        p, th3, err = ctx.segment_probability(0, si[sp[k]:sp[k + 1]])
        _, op, oth, oerr = O.segment_probability(ofit, si[sp[k]:sp[k + 1]])
        assert np.float64(p).view(np.uint64) == np.float64(op).view(np.uint64)
        assert np.float64(err).view(np.uint64) == np.float64(oerr).view(np.uint64)
        assert np.array_equal(np.asarray(th3).view(np.uint64), oth.view(np.uint64))
    # the chains' starting energy (initializeState: everything in one segment) is the whole-frame golden
    ctx.frame_set_graph(0, sc.row_ptr, sc.col)
    ctx.frames_edge_probs()
    ctx.chains_init(4, seed=1)
    e0 = ctx.scalars()["energy"]
    assert np.all(np.abs(e0 - big[name + "_set_probL"][0]) <= REL * abs(big[name + "_set_probL"][0]))


def _exact_residual(m):
    """Sbb - m'M^-1 m of one row of integer moments, in exact rational arithmetic."""
    from fractions import Fraction as Fr
    N, Su, Sv, Suu, Suv, Svv, Sb, Sub, Svb, Sbb = (int(x) for x in m)
    A = [[Fr(Suu), Fr(Suv), Fr(Su), Fr(Sub)], [Fr(Suv), Fr(Svv), Fr(Sv), Fr(Svb)], [Fr(Su), Fr(Sv), Fr(N), Fr(Sb)]]
    for i in range(3):
        for j in range(i + 1, 3):
            f = A[j][i] / A[i][i]
            A[j] = [a - f * b for a, b in zip(A[j], A[i])]
    x = [Fr(0)] * 3
    for i in (2, 1, 0):
        x[i] = (A[i][3] - sum(A[i][j] * x[j] for j in range(i + 1, 3))) / A[i][i]
    return Fr(Sbb) - (Fr(Sub) * x[0] + Fr(Svb) * x[1] + Fr(Sb) * x[2])


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_error_term_against_exact_rational_arithmetic(big, name):
    """A third, rounding-free witness for the segment error: the identity
    sum_i 1/2 |R_i theta - d_i|^2 = ((Sbb - m'M^-1 m) - sum_i (Sbb_i - m_i'M_i^-1 m_i)) / (2 sigma^2)
    (MyMath.h:247-252 at the fused mean) evaluated with Python integers / Fractions from the exact integer
    moments.  Both the literal-reference golden (QR, member by member) and the oracle (sums, cancelling) must sit
    within 1e-6 of it; in practice the whole-frame set agrees to 1e-15 and the per-plane sets, where the true
    error is ~10 against Sbb ~1e10, to 1e-8."""
    sc = scene_of(big, name)
    S = sc.n_superpixels
    mom = O.moments(sc.disp, sc.label, S)
    fit, _, _ = O.local_fit(mom, sc.width, sc.height)
    sp, si = big[name + "_set_ptr"], big[name + "_set_idx"]
    n_sets = len(sp) - 1
    for k in (0, 1, 2, n_sets - 2, n_sets - 1):
        ms = si[sp[k]:sp[k + 1]]
        tot = _exact_residual(mom[ms].sum(axis=0)) - sum(_exact_residual(mom[i]) for i in ms)
        exact = float(tot / 14)
        _, _, _, e = O.segment_probability(fit, ms)
        assert abs(e - exact) <= REL * exact, (name, k, e, exact)
        assert abs(big[name + "_set_err"][k] - exact) <= REL * exact, (name, k)
