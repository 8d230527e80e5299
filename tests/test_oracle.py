"""CPU tests: the oracle (oracle/mcmc_oracle.c) against what pins it.

No reference golden vectors exist for this path; the pins are
Random123's published Philox KAT, libm, numpy integer sums, and
tests/golden/small_frames.npz, which was minted by tests/literal_ref.py (a
numpy emulation of the reference's gtsam call sequence), not by the oracle.
"""
import math

import numpy as np
import pytest

from tests import oracle_lib as O

REL = 1e-6  # north_star: segment energies within 1e-6 relative of the reference's path


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kats:
        out = np.zeros(4, np.uint32)
        O.lib().orc_philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32), out)
        assert out.tolist() == want


def test_det_log_exp_against_libm():
    rng = np.random.default_rng(0)
    xs = np.concatenate([np.exp(rng.uniform(-700, 700, 4000)), rng.uniform(0.5, 2.0, 2000),
                         [1.0, 2.0, 0.5, 7.0, 1e-310, 5e-324, 1.7976931348623157e308]])
    for x in xs:
        got, want = O.lib().orc_det_log(float(x)), math.log(x)
        assert abs(got - want) <= 4e-16 * max(1.0, abs(want)), (x, got, want)
    es = np.concatenate([rng.uniform(-740, 709, 4000), rng.uniform(-1, 1, 2000), [0.0, -0.0, 1.0, -745.0, 709.7]])
    for x in es:
        got, want = O.lib().orc_det_exp(float(x)), math.exp(x)
        assert abs(got - want) <= 1e-15 * want + 5e-324, (x, got, want)
    L = O.lib()
    assert L.orc_det_log(0.0) == -math.inf and math.isnan(L.orc_det_log(-1.0))
    assert L.orc_det_log(math.inf) == math.inf and math.isnan(L.orc_det_log(math.nan))
    assert L.orc_det_exp(-math.inf) == 0.0 and L.orc_det_exp(800.0) == math.inf
    assert L.orc_det_exp(-800.0) == 0.0 and math.isnan(L.orc_det_exp(math.nan))


def _case(golden, ci):
    p = f"c{ci}_"
    return {k[len(p):]: golden[k] for k in golden.files if k.startswith(p)}


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_moments_and_local_fit_golden(golden, ci):
    g = _case(golden, ci)
    S = int(g["S"]); H, W = g["disp"].shape
    mom = O.moments(g["disp"], g["label"], S)
    assert np.array_equal(mom, g["moments"])           # integer work: bit-exact
    fit, valid, lerr = O.local_fit(mom, W, H)
    assert valid.all()
    th = fit[:, 9:12].copy()
    th[:, 2] -= th[:, 0] * (W // 2) + th[:, 1] * (H // 2)   # centred -> raw coordinates
    assert np.allclose(th, g["theta_raw"], rtol=REL, atol=1e-9)
    assert np.allclose(lerr, g["local_err"], rtol=REL, atol=1e-9)


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_segment_energy_golden(golden, ci):
    g = _case(golden, ci)
    S = int(g["S"]); H, W = g["disp"].shape
    fit, valid, _ = O.local_fit(O.moments(g["disp"], g["label"], S), W, H)
    sp, si = g["set_ptr"], g["set_idx"]
    for k in range(len(sp) - 1):
        st, p, th, err = O.segment_probability(fit, si[sp[k]:sp[k + 1]])
        assert st == 0
        assert abs(p - g["set_probL"][k]) <= REL * abs(g["set_probL"][k])
        assert abs(err - g["set_err"][k]) <= REL * abs(g["set_err"][k]) + 1e-9
        raw = th.copy(); raw[2] -= th[0] * (W // 2) + th[1] * (H // 2)
        assert np.allclose(raw, g["set_theta_raw"][k], rtol=REL, atol=1e-9)


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_edge_probs_golden(golden, ci):
    g = _case(golden, ci)
    S = int(g["S"]); H, W = g["disp"].shape
    fit, valid, _ = O.local_fit(O.moments(g["disp"], g["label"], S), W, H)
    st, q = O.edge_probs(fit, valid, g["row_ptr"], g["col"])
    assert st == 0
    assert np.allclose(q, g["q"], rtol=REL, atol=1e-12)
    # symmetric: q_ij == q_ji bit for bit (Problem.cpp:102-103 stores one value on both ends)
    src = np.repeat(np.arange(S), np.diff(g["row_ptr"].astype(np.int64)))
    d = {(int(i), int(j)): v for i, j, v in zip(src, g["col"], q)}
    assert all(d[(j, i)] == v for (i, j), v in d.items())


def test_filter_drops_degenerate_superpixels():
    # OverSegmentation.cpp:198-222: <10 px, a single row/column, or all-zero disparity
    W, H = 32, 16
    label = np.full((H, W), 0xFFFF, np.uint16); disp = np.full((H, W), 100, np.int16)
    label[0:3, 0:3] = 0          # 9 px
    label[5, 0:12] = 1           # one row (linear)
    label[0:12, 20] = 2          # one column (linear)
    label[8:12, 8:12] = 3; disp[8:12, 8:12] = 0   # zero disparity
    label[12:16, 24:32] = 4      # fine
    mom = O.moments(disp, label, 5)
    _, valid, _ = O.local_fit(mom, W, H)
    assert valid.tolist() == [0, 0, 0, 0, 1]
    row_ptr = np.array([0, 1, 1, 1, 1, 2], np.uint32); col = np.array([4, 0], np.uint32)
    fit, valid, _ = O.local_fit(mom, W, H)
    st, q = O.edge_probs(fit, valid, row_ptr, col)
    assert (q == -1.0).all()     # edges to dropped superpixels do not exist in the reference graph


def test_energy_invariant_to_coordinate_shift(scene_c2_small):
    # shifting the frame inside a larger canvas must not change any segment energy
    sc = scene_c2_small
    S = sc.n_superpixels
    fit0, _, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    W2, H2 = sc.width + 200, sc.height + 120
    disp = np.zeros((H2, W2), np.int16); label = np.full((H2, W2), 0xFFFF, np.uint16)
    disp[100:100 + sc.height, 150:150 + sc.width] = sc.disp
    label[100:100 + sc.height, 150:150 + sc.width] = sc.label
    fit1, _, _ = O.local_fit(O.moments(disp, label, S), W2, H2)
    for ms in ([3, 4, 5], list(range(0, S, 3)), list(range(S))):
        p0 = O.segment_probability(fit0, ms)[1]; p1 = O.segment_probability(fit1, ms)[1]
        assert abs(p0 - p1) <= 1e-9 * abs(p0)


def _chains(sc, n, seed, off=0, **kw):
    S = sc.n_superpixels
    fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    _, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    return O.Chains(fit, valid, sc.row_ptr, sc.col, q, n, seed, off, **kw)


def test_sampler_determinism_and_chain_offset(scene_c2_small):
    a = _chains(scene_c2_small, 4, seed=5)
    st, tra = a.run(50, 250, trace=True)
    assert st == 0
    b = _chains(scene_c2_small, 4, seed=5)
    _, trb = b.run(50, 250, trace=True)
    assert np.array_equal(tra, trb) and np.array_equal(a.labels(), b.labels())
    # chains 2,3 run alone (another "GPU") reproduce chains 2,3 of the full run
    c = _chains(scene_c2_small, 2, seed=5, off=2)
    _, trc = c.run(50, 250, trace=True)
    assert np.array_equal(trc, tra[2:]) and np.array_equal(c.labels(), a.labels()[2:])
    assert np.array_equal(c.scalars()["trace_hash"], a.scalars()["trace_hash"][2:])
    # a run split in two calls continues the same chain
    d = _chains(scene_c2_small, 4, seed=5)
    d.run(50, 100); d.run(0, 150)
    assert np.array_equal(d.labels(), a.labels())
    assert np.array_equal(d.scalars()["state_hash"], a.scalars()["state_hash"])


def test_sampler_invariants(scene_c2_small):
    sc = scene_c2_small
    ch = _chains(sc, 3, seed=9)
    st, tr = ch.run(100, 400, trace=True)
    assert st == 0
    s = ch.scalars()
    assert (s["n_case"].sum(axis=1) == 500).all()
    assert (s["n_accept"] == (tr & 1).sum(axis=1)).all()
    # running energy equals the sum of cached segment energies (Segmentation.cpp:185-195)
    assert np.allclose(s["energy"], ch.recompute_energy(), rtol=1e-9)
    assert (s["best_energy"] >= s["energy"] - 1e-9).all()
    # Statistics: counts add up to the post-burn-in steps; most-visited is the arg max
    for k, v in enumerate(ch.visits()):
        assert sum(v.values()) + int(s["overflow"][k]) == 400
        assert max(v.values()) == int(s["top_count"][k])
    m = ch.marginals()
    assert (m.sum(axis=1) == 3 * 400).all()
    # canonical labels: first appearance order, dense
    for lab in ch.labels(0, True):
        seen = -1
        for x in lab:
            assert x <= seen + 1
            seen = max(seen, x)
    # recomputing every segment of the final state from scratch reproduces the cached energy
    fit = ch._keep[0]
    raw = ch.labels(0, False)
    for k in range(3):
        tot = sum(O.segment_probability(fit, np.nonzero(raw[k] == r)[0])[1] for r in np.unique(raw[k]))
        assert abs(tot - s["energy"][k]) <= 1e-8 * abs(tot)


def test_sampler_finds_planes():
    # sanity of the restated algorithm: on a clean 3-plane scene the most visited
    # labelling groups superpixels by plane far better than chance
    from mcmc_amd.scene import make_scene
    sc = make_scene(width=160, height=120, n_planes=3, n_superpixels=40, seed=3, noise=0.5)
    ch = _chains(sc, 2, seed=1)
    ch.run(3000, 3000)
    lab = ch.labels(2, True)[0]
    agree = 0; tot = 0
    for i in range(sc.n_superpixels):
        for j in range(i + 1, sc.n_superpixels):
            tot += 1
            agree += (lab[i] == lab[j]) == (sc.plane_of_sp[i] == sc.plane_of_sp[j])
    assert agree / tot > 0.8


def test_filter_keeps_mixed_sign_disparities():
    # isZeroDisparity (OverSegmentation.cpp:422-432) with the size_t widening of OverSegmentation.cpp:148: a
    # superpixel is dropped only if EVERY disparity is zero; +5 / -5 summing to zero is kept
    H, W = 32, 32
    label = np.full((H, W), 0xFFFF, np.uint16); disp = np.zeros((H, W), np.int16)
    label[0:4, 0:4] = 0; disp[0:4, 0:2] = 5; disp[0:4, 2:4] = -5
    label[8:12, 0:4] = 1
    label[16:20, 0:4] = 2; disp[16:20, 0:4] = -3
    _, valid, _ = O.local_fit(O.moments(disp, label, 3), W, H)
    assert valid.tolist() == [1, 0, 1]
