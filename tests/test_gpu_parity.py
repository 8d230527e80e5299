"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs and against the committed golden fixtures.

Bar: bit-exact for integer / index / accept-reject work and for every f64 the
decisions depend on (same IEEE operations in the same order on both sides);
1e-6 relative against the literal-reference goldens.
"""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
REL = 1e-6


def _case(golden, ci):
    p = f"c{ci}_"
    return {k[len(p):]: golden[k] for k in golden.files if k.startswith(p)}


def _load(ctx, frame, sc):
    from mcmc_amd import host
    host.load_scene(ctx, frame, sc)


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_wave_sums_bit_exact(gpu_ctx_factory):
    """The chain kernel's wave reductions, one value at a time and eleven at once (the folded butterfly of the two-wave
    form), against the oracle's butterfly (oracle/mcmc_oracle.c: wave_sum) on inexact doubles of mixed magnitude."""
    ctx = gpu_ctx_factory()
    rng = np.random.default_rng(11)
    n = 300
    x = np.ldexp(rng.uniform(-0.5, 0.5, (n, 11, 64)), rng.integers(-20, 20, (n, 11, 64)))
    x[rng.random(x.shape) < 0.2] = 0.0
    x[0] = 0.0; x[1, :, 1:] = 0.0           # empty wave; one lane only
    plain, folded = ctx.selftest_wave_sums(x)
    L = O.lib()
    want = np.array([[L.orc_wave_sum(np.ascontiguousarray(x[w, k])) for k in range(11)] for w in range(n)])
    assert np.array_equal(bits(plain), bits(want))
    assert np.array_equal(bits(folded), bits(want))


def test_device_math_bit_exact(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(-745, 709, 20000),
                        rng.uniform(-2, 2, 10000), [0.0, 1.0, 7.0, np.inf, -np.inf, np.nan, 5e-324, 1e-310]])
    y = np.concatenate([rng.uniform(-1e3, 1e3, x.size - 3), [3.0, 7.0, 4294967295.0]])
    gl, ge, gd = ctx.selftest_math(x, y)
    L = O.lib()
    wl = np.array([L.orc_det_log(float(v)) for v in x]); we = np.array([L.orc_det_exp(float(v)) for v in x])
    assert np.array_equal(bits(gl), bits(wl))
    assert np.array_equal(bits(ge), bits(we))
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(gd), bits(x / y))   # IEEE division on the device
    for ctr, key in [([0, 0, 0, 0], [0, 0]), ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])]:
        want = np.zeros(4, np.uint32)
        L.orc_philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32), want)
        assert np.array_equal(ctx.selftest_philox(ctr, key), want)
    assert ctx.selftest_philox([0, 0, 0, 0], [0, 0]).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_frame_kernels_golden(golden, gpu_ctx_factory, ci):
    g = _case(golden, ci)
    S = int(g["S"]); H, W = g["disp"].shape
    ctx = gpu_ctx_factory(width=W, height=H)
    ctx.frames_alloc(1, S, g["col"].size)
    ctx.frame_upload(0, g["disp"], g["label"], S)
    ctx.frame_set_graph(0, g["row_ptr"], g["col"])
    ctx.frames_moments(); ctx.frames_edge_probs()
    assert np.array_equal(ctx.moments(0), g["moments"])
    fit, valid, lerr = ctx.local_fit(0)
    th = fit[:, 9:12].copy(); th[:, 2] -= th[:, 0] * (W // 2) + th[:, 1] * (H // 2)
    assert valid.all() and np.allclose(th, g["theta_raw"], rtol=REL, atol=1e-9)
    assert np.allclose(lerr, g["local_err"], rtol=REL, atol=1e-9)
    assert np.allclose(ctx.edge_probs(0), g["q"], rtol=REL, atol=1e-12)
    sp, si = g["set_ptr"], g["set_idx"]
    for k in range(len(sp) - 1):
        p, th3, err = ctx.segment_probability(0, si[sp[k]:sp[k + 1]])
        assert abs(p - g["set_probL"][k]) <= REL * abs(g["set_probL"][k])
        assert abs(err - g["set_err"][k]) <= REL * abs(g["set_err"][k]) + 1e-9
    # and bit-exact against the oracle
    ofit, ovalid, olerr = O.local_fit(O.moments(g["disp"], g["label"], S), W, H)
    assert np.array_equal(bits(fit), bits(ofit)) and np.array_equal(valid, ovalid)
    assert np.array_equal(bits(lerr), bits(olerr))
    _, oq = O.edge_probs(ofit, ovalid, g["row_ptr"], g["col"])
    assert np.array_equal(bits(ctx.edge_probs(0)), bits(oq))
    for k in range(len(sp) - 1):
        p, th3, err = ctx.segment_probability(0, si[sp[k]:sp[k + 1]])
        _, op, oth, oerr = O.segment_probability(ofit, si[sp[k]:sp[k + 1]])
        assert bits(p) == bits(op) and bits(err) == bits(oerr) and np.array_equal(bits(th3), bits(oth))


@pytest.mark.parametrize("shape", [(640, 480, 500, 8), (1920, 1080, 4000, 24), (100, 37, 12, 2), (648, 480, 300, 4)])
def test_moments_full_size_bit_exact(gpu_ctx_factory, shape):
    # BASELINE configs 2 and 3 at full size; a width that is not a multiple of 8 (generic kernel);
    # a width (648) for which 8*blockDim is not a multiple of W.
    from mcmc_amd.scene import make_scene
    W, H, S, K = shape
    sc = make_scene(width=W, height=H, n_planes=K, n_superpixels=S, seed=2)
    ctx = gpu_ctx_factory(width=W, height=H)
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments()
    want = O.moments(sc.disp, sc.label, sc.n_superpixels)
    assert np.array_equal(ctx.moments(0), want)
    # checksum-of-checksums property: totals over superpixels equal totals over labelled pixels
    m = sc.label < sc.n_superpixels
    assert ctx.moments(0)[:, 0].sum() == m.sum()
    assert ctx.moments(0)[:, 6].sum() == sc.disp[m].astype(np.int64).sum()
    # the same integers from every launch shape: 16-pixel lanes (when the width allows), two bands, table in HBM
    for opt, val in (("k1_px", 16), ("k1_bands", 2), ("k1_use_lds", 0)):
        ctx.set_option(opt, val)
        ctx.frames_moments()
        assert np.array_equal(ctx.moments(0), want), (opt, val)


def test_moments_edge_cases(gpu_ctx_factory):
    W, H = 64, 32
    ctx = gpu_ctx_factory(width=W, height=H)
    ctx.frames_alloc(3, 40, 8)
    rng = np.random.default_rng(3)
    # frame 0: nothing labelled; frame 1: one superpixel = whole frame, extreme disparities
    # (incl. negative, the reference's size_t quirk is NOT reproduced: signed int64 sums);
    # frame 2: salt-and-pepper labels, every pixel a different run
    d0 = rng.integers(0, 2047, (H, W)).astype(np.int16); l0 = np.full((H, W), 0xFFFF, np.uint16)
    d1 = rng.choice(np.array([-32768, 32767, 0, 2047], np.int16), (H, W)); l1 = np.zeros((H, W), np.uint16)
    d2 = rng.integers(0, 2047, (H, W)).astype(np.int16); l2 = rng.integers(0, 40, (H, W)).astype(np.uint16)
    l2[rng.random((H, W)) < 0.1] = 0xFFFF
    for f, (d, l, S) in enumerate([(d0, l0, 5), (d1, l1, 1), (d2, l2, 40)]):
        ctx.frame_upload(f, d, l, S)
    ctx.frames_moments()
    for f, (d, l, S) in enumerate([(d0, l0, 5), (d1, l1, 1), (d2, l2, 40)]):
        assert np.array_equal(ctx.moments(f), O.moments(d, l, S)), f
    # recomputing is idempotent
    ctx.frames_moments()
    assert np.array_equal(ctx.moments(2), O.moments(d2, l2, 40))
    # the A1 filter: degenerate superpixels are flagged and their edges get q = -1
    label = np.full((H, W), 0xFFFF, np.uint16); disp = np.full((H, W), 100, np.int16)
    label[0:3, 0:3] = 0; label[5, 0:12] = 1; label[0:12, 20] = 2
    label[8:12, 8:12] = 3; disp[8:12, 8:12] = 0; label[12:16, 24:32] = 4
    ctx.frame_upload(0, disp, label, 5)
    ctx.frame_set_graph(0, np.array([0, 1, 1, 1, 1, 2], np.uint32), np.array([4, 0], np.uint32))
    ctx.frames_moments(0, 1); ctx.frames_edge_probs(0, 1)
    assert ctx.local_fit(0)[1].tolist() == [0, 0, 0, 0, 1]
    assert (ctx.edge_probs(0) == -1.0).all()


def _oracle_chains(sc, n, seed, off=0, **kw):
    S = sc.n_superpixels
    fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    _, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    return O.Chains(fit, valid, sc.row_ptr, sc.col, q, n, seed, off, **kw)


def _compare_chains(ctx, orc, frame=0, lo=0, hi=None):
    hi = ctx.n_chains if hi is None else hi
    S = orc.S
    gs, os_ = ctx.scalars(), orc.scalars()
    for k in ("state_hash", "trace_hash", "n_accept", "n_case", "status", "top_count", "top_hash", "overflow"):
        assert np.array_equal(gs[k][lo:hi], os_[k]), k
    assert np.array_equal(bits(gs["energy"][lo:hi]), bits(os_["energy"]))
    assert np.array_equal(bits(gs["best_energy"][lo:hi]), bits(os_["best_energy"]))
    for which in (0, 1, 2):
        for canon in (True, False):
            assert np.array_equal(ctx.labels(which, canon)[lo:hi, :S], orc.labels(which, canon)), (which, canon)
    assert np.array_equal(bits(ctx.recompute_energy()[lo:hi]), bits(orc.recompute_energy()))
    assert ctx.visits()[lo:hi] == orc.visits()
    # Problem::propose's out-parameters of the last step (Problem.cpp:213-217)
    gt, gp = ctx.last_ratios(); ot, op = orc.last_ratios()
    assert np.array_equal(bits(gt[lo:hi]), bits(ot)) and np.array_equal(bits(gp[lo:hi]), bits(op))
    # the running energy (sum of targetRatios over accepted moves) against a refit of every segment of the
    # current labeling from its member list (Segment(superPixels), Graph.h:33-40), first chain of the block
    lab = ctx.labels(0, False)[lo, :S]
    tot = 0.0
    for root in np.unique(lab[lab != 0xFFFF]):
        tot += ctx.segment_probability(frame, np.flatnonzero(lab == root).astype(np.uint32))[0]
    assert abs(tot - gs["energy"][lo]) <= 1e-9 * abs(tot), (tot, gs["energy"][lo])


def test_chain_parity_small(scene_c2_small, gpu_ctx_factory):
    sc = scene_c2_small
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height)
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    n = 8
    ctx.chains_init(n, seed=5)
    ctx.chains_set_trace(400)
    ctx.chains_run(100, 300)
    orc = _oracle_chains(sc, n, seed=5)
    st, tr = orc.run(100, 300, trace=True)
    assert st == 0
    assert np.array_equal(ctx.trace(), tr)             # accept/reject + case sequence, bit-exact
    _compare_chains(ctx, orc)
    assert np.array_equal(ctx.marginals(0)[:sc.n_superpixels], orc.marginals())
    # continue the run: steps carry on, statistics accumulate
    ctx.chains_run(0, 200)
    st, tr2 = orc.run(0, 200, trace=True)
    assert np.array_equal(ctx.trace(), tr2)
    _compare_chains(ctx, orc)
    assert np.array_equal(ctx.marginals(0)[:sc.n_superpixels], orc.marginals())


@pytest.mark.parametrize("dense,spec,variant", [(1, -1, 28), (1, 0, 128), (-1, -1, 14), (-1, 0, 244), (-1, 1, 244), (-1, 2, 44), (-1, 3, 144),
                                                (-1, 4, 244)])
def test_chain_parity_dense_variant(scene_c2_small, gpu_ctx_factory, dense, spec, variant):
    # eight chains per workgroup with the per-superpixel records read through L2 (picked by itself when the chains
    # exceed four per CU; forced here); the plain form (four chains per workgroup, everything in LDS); and the
    # speculative form (two, three or four waves per chain in step order; four are picked by themselves when the
    # chains are at most four per CU and the state fits; variant = 44 + 100 * (waves - 2)): identical chains
    sc = scene_c2_small
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height)
    ctx.frames_alloc(2, sc.n_superpixels, sc.col.size)
    _load(ctx, 0, sc); _load(ctx, 1, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.set_option("k3_dense", dense)
    ctx.set_option("k3_spec", spec)
    n = 32
    ctx.chains_init(n, chains_per_frame=16, seed=21)
    ctx.chains_set_trace(260)
    ctx.chains_run(60, 200)
    assert ctx.last_variant() == variant
    tr = ctx.trace()
    for f in range(2):
        orc = _oracle_chains(sc, 16, seed=21, off=16 * f)
        st, otr = orc.run(60, 200, trace=True)
        assert st == 0 and np.array_equal(tr[16 * f:16 * f + 16], otr)
        _compare_chains(ctx, orc, frame=f, lo=16 * f, hi=16 * f + 16)


def test_chain_parity_speculative_form_long_run(gpu_ctx_factory):
    # the speculative form (four waves per chain at this size, in-order commit) at C2 size against the oracle over runs long
    # enough for thousands of hand-overs between the waves, continued twice (the step counter carries on), with
    # traces; and the same chains from the plain form and from two and three waves per chain: bit for bit the same in every read-back
    from mcmc_amd.scene import make_scene
    sc = make_scene(seed=1)
    S = sc.n_superpixels
    ctx = gpu_ctx_factory()
    ctx.frames_alloc(1, S, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    n = 12
    orc = _oracle_chains(sc, n, seed=8)
    ctx.chains_init(n, chains_per_frame=n, seed=8)
    ctx.chains_set_trace(3000)
    for burn, iters in ((500, 2500), (0, 1500), (0, 7)):
        ctx.chains_run(burn, iters)
        assert ctx.last_variant() == 244
        st, tr = orc.run(burn, iters, trace=True)
        assert st == 0 and np.array_equal(ctx.trace(), tr)
        _compare_chains(ctx, orc)
        assert np.array_equal(ctx.marginals(0)[:S], orc.marginals())
    spec = ctx.scalars()
    for opt, variant in ((-1, 14), (2, 44), (3, 144)):
        ctx.set_option("k3_spec", opt)
        ctx.chains_init(n, chains_per_frame=n, seed=8)
        for burn, iters in ((500, 2500), (0, 1500), (0, 7)):
            ctx.chains_run(burn, iters)
        assert ctx.last_variant() == variant
        other = ctx.scalars()
        for k in spec:
            assert np.array_equal(spec[k].view(np.uint64) if spec[k].dtype == np.float64 else spec[k],
                                  other[k].view(np.uint64) if other[k].dtype == np.float64 else other[k]), (k, variant)


def test_best_labeling_on_the_device(scene_c2_small, gpu_ctx_factory):
    # mcmc_chains_get_best: the arg-max over the chains on the device (ties to the lowest chain id) and one labeling
    # back, against the arg-max taken on the host over every chain's read-back, and against the oracle's chain
    sc = scene_c2_small
    S = sc.n_superpixels
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height)
    ctx.frames_alloc(1, S + 3, sc.col.size)                   # s_max above S: the padding reads MCMC_NOLABEL
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    n, off = 70, 1000
    ctx.chains_init(n, seed=6, chain_offset=off)
    for burn, iters in ((0, 3), (40, 200)):                  # after 3 steps many chains tie: the lowest id wins
        ctx.chains_run(burn, iters)
        s = ctx.scalars()
        for which, key, lab_which in ((1, s["best_energy"], 1), (2, s["top_count"], 2)):
            k = int(np.flatnonzero(key == key.max())[0])
            for canon in (True, False):
                e, v, cid, lab = ctx.best(which, canon)
                assert cid == off + k, (which, cid, k)
                assert bits(e) == bits(s["best_energy"][k]) and v == int(s["top_count"][k])
                want = ctx.labels(lab_which, canon)[k]
                assert np.array_equal(lab, want) and (lab[S:] == 0xFFFF).all()
    orc = _oracle_chains(sc, n, seed=6, off=off)
    orc.run(0, 3); orc.run(40, 200)
    e, v, cid, lab = ctx.best(1, True)
    oe = orc.scalars()["best_energy"]
    assert bits(e) == bits(oe.max()) and np.array_equal(lab[:S], orc.labels(1, True)[cid - off])
    # the record a collective reads is the same data, on the device
    import ctypes
    ptr, nbytes = ctx.best_record_device(1, True)
    ctx.synchronize()
    assert nbytes == (2 + (S + 3 + 3) // 4) * 8 and ptr


def test_chain_offset_sharding(scene_c2_small, gpu_ctx_factory):
    # a shard that owns chains [6, 10) of a 10-chain job reproduces them exactly (multi-GPU rule)
    sc = scene_c2_small
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height)
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(4, seed=11, chain_offset=6)
    ctx.chains_run(50, 150)
    orc = _oracle_chains(sc, 10, seed=11)
    orc.run(50, 150)
    assert np.array_equal(ctx.scalars()["trace_hash"], orc.scalars()["trace_hash"][6:10])
    assert np.array_equal(ctx.labels()[:, :sc.n_superpixels], orc.labels()[6:10])


def test_chain_parity_c2_full_size(gpu_ctx_factory):
    # BASELINE config 2 at full size (640x480, ~500 superpixels): a few chains against the oracle,
    # all 1024 chains checked through size-independent properties.
    from mcmc_amd.scene import make_scene
    sc = make_scene(seed=1)
    ctx = gpu_ctx_factory()
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(1024, seed=1)
    ctx.chains_run(100, 200)
    orc = _oracle_chains(sc, 6, seed=1)
    orc.run(100, 200)
    _compare_chains(ctx, orc, lo=0, hi=6)
    s = ctx.scalars()
    assert (s["status"] == 0).all()
    assert (s["n_case"].sum(axis=1) == 300).all()
    assert np.allclose(s["energy"], ctx.recompute_energy(), rtol=1e-9)
    v = ctx.visits()
    assert all(sum(d.values()) + int(o) == 200 for d, o in zip(v, s["overflow"]))
    m = ctx.marginals(0)[:sc.n_superpixels]
    assert (m.sum(axis=1) == 1024 * 200).all()
    # determinism: the same seed twice gives identical chains
    ctx.chains_init(1024, seed=1)
    ctx.chains_run(100, 200)
    assert np.array_equal(ctx.scalars()["trace_hash"], s["trace_hash"])


@pytest.mark.parametrize("placement", [3, 0])
def test_chain_parity_large_s_global_tables(gpu_ctx_factory, placement):
    # tables that do not fit LDS (config 3 style: ~1500 superpixels here): the CSR topology alone in LDS
    # (placement 3, picked by itself), and everything behind L2 (placement 0, which still larger frames take)
    from mcmc_amd.scene import make_scene
    sc = make_scene(width=960, height=540, n_planes=10, n_superpixels=1500, seed=2)
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height)
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    if placement == 0:
        ctx.set_option("k3_dense", -1)
    ctx.chains_init(6, chains_per_frame=6, seed=3)
    ctx.chains_set_trace(120)
    ctx.chains_run(40, 80)
    assert ctx.last_variant() % 100 // 10 == placement
    orc = _oracle_chains(sc, 6, seed=3)
    st, tr = orc.run(40, 80, trace=True)
    assert np.array_equal(ctx.trace(), tr)
    _compare_chains(ctx, orc)


def test_multi_frame_batch(gpu_ctx_factory):
    # config 5 style: several frames, a few chains each; frame f's chains only see frame f
    from mcmc_amd.scene import make_scene
    scenes = [make_scene(width=160, height=120, n_planes=3, n_superpixels=40 + 5 * f, seed=100 + f) for f in range(5)]
    smax = max(s.n_superpixels for s in scenes); emax = max(s.col.size for s in scenes)
    ctx = gpu_ctx_factory(width=160, height=120)
    ctx.frames_alloc(5, smax, emax)
    for f, sc in enumerate(scenes):
        _load(ctx, f, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    cpf = 4
    ctx.chains_init(5 * cpf, chains_per_frame=cpf, seed=21)
    ctx.chains_run(30, 120)
    s = ctx.scalars()
    for f, sc in enumerate(scenes):
        assert np.array_equal(ctx.moments(f), O.moments(sc.disp, sc.label, sc.n_superpixels))
        orc = _oracle_chains(sc, cpf, seed=21, off=f * cpf)
        orc.run(30, 120)
        assert np.array_equal(s["trace_hash"][f * cpf:(f + 1) * cpf], orc.scalars()["trace_hash"]), f
        assert np.array_equal(ctx.labels()[f * cpf:(f + 1) * cpf, :sc.n_superpixels], orc.labels())
        assert np.array_equal(ctx.marginals(f)[:sc.n_superpixels], orc.marginals())


def _properties(ctx, S, frames, cpf, burn, iters):
    """Size-independent checks over every chain of a job (the oracle only re-runs a few of them)."""
    s = ctx.scalars()
    assert (s["status"] == 0).all()
    assert (s["n_case"].sum(axis=1) == burn + iters).all()
    assert np.allclose(s["energy"], ctx.recompute_energy(), rtol=1e-9)
    assert (s["best_energy"] >= s["energy"]).all()
    v = ctx.visits()
    assert all(sum(d.values()) + int(o) == iters for d, o in zip(v, s["overflow"]))
    for f in range(frames):
        m = ctx.marginals(f)[:S[f]]
        assert (m.sum(axis=1) == cpf * iters).all(), f
    return s


def test_chain_parity_c3_full_size(gpu_ctx_factory):
    # BASELINE config 3 at full size: 1920x1080, ~4500 superpixels, one GPU's share of the 4096 chains (512).
    # The tables (~640 KB) do not fit LDS: placement 3 (CSR topology in LDS, the rest behind L2), several waves per chain.
    from mcmc_amd.scene import make_scene
    sc = make_scene(width=1920, height=1080, n_planes=24, n_superpixels=4000, seed=2)
    S = sc.n_superpixels
    assert S > 4000
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height)
    ctx.frames_alloc(1, S, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    # 4 chains x 100 steps, trace-exact against the oracle
    ctx.chains_init(4, seed=2)
    ctx.chains_set_trace(100)
    ctx.chains_run(30, 70)
    assert ctx.last_variant() % 100 // 10 == 3
    orc = _oracle_chains(sc, 4, seed=2)
    st, tr = orc.run(30, 70, trace=True)
    assert st == 0 and np.array_equal(ctx.trace(), tr)
    _compare_chains(ctx, orc)
    assert np.array_equal(ctx.marginals(0)[:S], orc.marginals())
    # the share of one GPU: chains [1024, 1536) of the 4096 (rank 2 of 8), properties over all of them and the
    # first three against the oracle run with the same global ids
    ctx.chains_init(512, seed=2, chain_offset=1024)
    ctx.chains_run(50, 150)
    # two chains per CU: the speculative form with two chains per workgroup and four waves per chain (variant 232)
    assert ctx.last_variant() == 232
    s = _properties(ctx, [S], 1, 512, 50, 150)
    orc = _oracle_chains(sc, 3, seed=2, off=1024)
    orc.run(50, 150)
    assert np.array_equal(s["trace_hash"][:3], orc.scalars()["trace_hash"])
    assert np.array_equal(ctx.labels()[:3, :S], orc.labels())


def test_chain_parity_c5_share_full_size(gpu_ctx_factory):
    # BASELINE config 5, one GPU's share: 32 frames of 640x480 (~500 superpixels), 64 chains each = 2048 chains,
    # more than four per CU, so eight chains share a workgroup with the records behind L2 (variant 28; 128 with two
    # waves per chain, which is what fits and runs here).
    from mcmc_amd.scene import make_scene
    F, cpf, burn, iters = 32, 64, 40, 160
    scenes = [make_scene(seed=100 + f) for f in range(F)]
    smax = max(s.n_superpixels for s in scenes); emax = max(s.col.size for s in scenes)
    ctx = gpu_ctx_factory()
    ctx.frames_alloc(F, smax, emax)
    for f, sc in enumerate(scenes):
        _load(ctx, f, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(F * cpf, chains_per_frame=cpf, seed=5)
    ctx.chains_run(burn, iters)
    assert ctx.last_variant() == 128                          # eight chains per workgroup, two waves per chain
    s = _properties(ctx, [sc.n_superpixels for sc in scenes], F, cpf, burn, iters)
    lab = ctx.labels()
    for f in (0, 13, 31):
        sc = scenes[f]
        assert np.array_equal(ctx.moments(f), O.moments(sc.disp, sc.label, sc.n_superpixels))
        orc = _oracle_chains(sc, 2, seed=5, off=f * cpf)
        orc.run(burn, iters)
        assert np.array_equal(s["trace_hash"][f * cpf:f * cpf + 2], orc.scalars()["trace_hash"]), f
        assert np.array_equal(lab[f * cpf:f * cpf + 2, :sc.n_superpixels], orc.labels()), f


def test_statistics_at_baseline_step_count(gpu_ctx_factory):
    # Statistics keeps every visited state (Statistics.h:30-47).  BASELINE step count (1000 burn-in + 10000
    # measured) on the C2 scene (in fact ~120 distinct states per chain after burn-in, ~150 from the start).  The
    # visit table is sized from the run (visit_capacity = 0, room for one new state per measured step, grown and
    # re-hashed between launches): nothing can overflow, every count equals the oracle's (run with room for every
    # state), and the most-visited labeling (can05_sampler.cpp:36) is the oracle's.
    from mcmc_amd.scene import make_scene
    sc = make_scene(seed=1)
    S = sc.n_superpixels
    ctx = gpu_ctx_factory()
    ctx.frames_alloc(1, S, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(8, seed=1)
    ctx.chains_run(1000, 10000)
    orc = _oracle_chains(sc, 8, seed=1, hcap=16384)
    orc.run(1000, 10000)
    gs = ctx.scalars()
    assert (gs["overflow"] == 0).all() and (orc.scalars()["overflow"] == 0).all()
    gv = ctx.visits()
    assert max(len(v) for v in gv) > 64
    assert 10000 <= ctx.visit_capacity() <= 16384            # slots in use + steps to come, not twice the steps made
    _compare_chains(ctx, orc)
    # Statistics::getOrderedStates: most visited first, and its head is the top state
    od = ctx.ordered_states(3)
    assert od[0] == (int(gs["top_count"][3]), int(gs["top_hash"][3]))
    assert [c for c, _ in od] == sorted((c for c, _ in od), reverse=True) and sum(c for c, _ in od) == 10000
    # continuing the run grows the table again and keeps what it held; short launches do not grow it at all (the
    # occupancy is measured: ~150 states in use, not the 25000 steps made)
    ctx.chains_run(0, 15000)
    cap = ctx.visit_capacity()
    assert cap == 32768
    orc2 = _oracle_chains(sc, 2, seed=1, hcap=65536)
    orc2.run(1000, 25000)
    assert ctx.visits()[:2] == orc2.visits()
    assert np.array_equal(ctx.labels(2, True)[:2, :S], orc2.labels(2, True))
    for _ in range(30):
        ctx.chains_run(0, 1000)
    assert ctx.visit_capacity() == cap and (ctx.scalars()["overflow"] == 0).all()
    orc2.run(0, 30000)
    assert ctx.visits()[:2] == orc2.visits()


@pytest.mark.parametrize("spec,variant", [(0, 244), (2, 44), (-1, 14)])
def test_statistics_fixed_capacity_overflow_is_flagged(scene_c2_small, gpu_ctx_factory, spec, variant):
    # a caller-fixed visit table that fills up: the visits that found no slot are counted exactly as the oracle
    # counts them, and asking for the most-visited labeling says that it may be wrong.  In the several-waves-per-chain
    # kernel the wave that commits walks the full table while the others wait for it: the same counts, status 0, and
    # mcmc_ctx_synchronize has nothing to report
    from mcmc_amd import host
    sc = scene_c2_small
    ctx = gpu_ctx_factory(width=sc.width, height=sc.height, visit_capacity=5)
    ctx.frames_alloc(1, sc.n_superpixels, sc.col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.set_option("k3_spec", spec)
    ctx.chains_init(8, seed=9)        # a multiple of four: the several-waves forms take whole workgroups
    ctx.chains_run(50, 800)
    ctx.synchronize()
    assert ctx.last_variant() == variant and (ctx.scalars()["status"] == 0).all()
    orc = _oracle_chains(sc, 8, seed=9, hcap=5)
    orc.run(50, 800)
    gs, os_ = ctx.scalars(), orc.scalars()
    assert gs["overflow"].sum() > 0 and np.array_equal(gs["overflow"], os_["overflow"])
    assert ctx.visits() == orc.visits() and ctx.visit_capacity() == 5
    assert np.array_equal(gs["top_count"], os_["top_count"]) and np.array_equal(gs["top_hash"], os_["top_hash"])
    with pytest.raises(host.McmcError) as e:
        ctx.labels(2, True)
    assert e.value.code == -9
    assert np.array_equal(ctx.labels(0, True)[:, :sc.n_superpixels], orc.labels(0, True))   # other read-backs unaffected


def _hub_frame(W, H, S, seed, block=4):
    """S superpixels of block x block pixels on one noisy plane, superpixel 0 adjacent to every other one (a hub
    of degree S-1) plus a ring through the others: the graph is the caller's, not derived from the pixels."""
    rng = np.random.default_rng(seed)
    label = np.full((H, W), 0xFFFF, np.uint16)
    per_row = W // block
    assert S <= per_row * (H // block)
    for s in range(S):
        r, c = divmod(s, per_row)
        label[r * block:(r + 1) * block, c * block:(c + 1) * block] = s
    v, u = np.mgrid[0:H, 0:W]
    disp = np.clip(np.rint(300 + 0.03 * u - 0.02 * v + rng.normal(0, 1.0, (H, W))), 1, 2047).astype(np.int16)
    nb = [set() for _ in range(S)]
    for s in range(1, S):
        nb[0].add(s); nb[s].add(0)
        t = s + 1 if s + 1 < S else 1
        if t != s:
            nb[s].add(t); nb[t].add(s)
    rp = np.zeros(S + 1, np.uint32); col = []
    for s in range(S):
        col += sorted(nb[s]); rp[s + 1] = len(col)
    return disp, label, rp, np.array(col, np.uint32)


@pytest.mark.parametrize("S", [300, 2600])
def test_chain_parity_hub_superpixel(gpu_ctx_factory, S):
    # a superpixel with hundreds / thousands of neighbours takes many FIFO slots (one per 8 neighbours): the part
    # index of a slot has 12 bits (degree up to 32760).  S = 300: tables in LDS, 38 parts; S = 2600: degree 2599,
    # 325 parts (beyond the 8 bits round 1 had), tables behind L2.
    from types import SimpleNamespace
    W, H = 640, 480
    disp, label, rp, col = _hub_frame(W, H, S, seed=S)
    sc = SimpleNamespace(width=W, height=H, disp=disp, label=label, n_superpixels=S, row_ptr=rp, col=col)
    ctx = gpu_ctx_factory(width=W, height=H)
    ctx.frames_alloc(1, S, col.size)
    _load(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(4, seed=12)
    ctx.chains_set_trace(150)
    ctx.chains_run(30, 120)
    assert (ctx.last_variant() % 100 // 10 == 4) == (S == 300)
    orc = _oracle_chains(sc, 4, seed=12)
    st, tr = orc.run(30, 120, trace=True)
    assert st == 0 and np.array_equal(ctx.trace(), tr)
    _compare_chains(ctx, orc)


def test_filter_nonzero_disparity_rule_and_extreme_strips(gpu_ctx_factory):
    # A1 filter: isZeroDisparity (OverSegmentation.cpp:422-432) keeps a superpixel as soon as one disparity is
    # non-zero; values of mixed sign that sum to zero must not drop it (sum of squares, not sum).
    W, H = 64, 512
    ctx = gpu_ctx_factory(width=W, height=H)
    ctx.frames_alloc(2, 4, 8)
    label = np.full((H, W), 0xFFFF, np.uint16); disp = np.zeros((H, W), np.int16)
    label[0:4, 0:4] = 0; disp[0:4, 0:2] = 5; disp[0:4, 2:4] = -5          # sums to zero, not all zero: kept
    label[8:12, 0:4] = 1                                                   # all zero: dropped
    label[16:20, 0:4] = 2; disp[16:20, 0:4] = -3                           # all negative: kept (non-zero)
    label[24:28, 0:4] = 3; disp[24, 0] = 1                                 # a single non-zero pixel: kept
    ctx.frame_upload(0, disp, label, 4)
    # K1's strip-local 32-bit sums at the extreme: one superpixel = a whole tall frame of +-32768 / 32767
    rng = np.random.default_rng(5)
    d1 = rng.choice(np.array([-32768, 32767], np.int16), (H, W)); d1[:, ::2] = -32768
    l1 = np.zeros((H, W), np.uint16)
    ctx.frame_upload(1, d1, l1, 1)
    ctx.frames_moments()
    assert ctx.local_fit(0)[1].tolist() == [1, 0, 1, 1]
    _, ovalid, _ = O.local_fit(O.moments(disp, label, 4), W, H)
    assert ovalid.tolist() == [1, 0, 1, 1]
    assert np.array_equal(ctx.moments(0), O.moments(disp, label, 4))
    assert np.array_equal(ctx.moments(1), O.moments(d1, l1, 1))
    assert ctx.moments(1)[0, 9] == (d1.astype(np.int64) ** 2).sum()
    # rows with disparities beyond 11 bits among ordinary rows, several labels per chunk: the squares of the wide
    # rows take the queue path, the others the 32-bit strip sums
    rng = np.random.default_rng(9)
    d2 = rng.integers(0, 2047, (H, W)).astype(np.int16)
    wide_rows = rng.random(H) < 0.2
    d2[wide_rows] = rng.choice(np.array([-32768, 32767, 2048, -1, 5000], np.int16), (int(wide_rows.sum()), W))
    l2 = (np.arange(W)[None, :] // 5 + (np.arange(H)[:, None] // 23) * 13).astype(np.uint16)
    l2[rng.random((H, W)) < 0.02] = 0xFFFF
    S2 = int(l2[l2 != 0xFFFF].max()) + 1
    ctx.frames_alloc(1, S2, 8)
    for px in (8, 16):
        ctx.set_option("k1_px", px)
        ctx.frame_upload(0, d2, l2, S2)
        ctx.frames_moments()
        assert np.array_equal(ctx.moments(0), O.moments(d2, l2, S2)), px


def test_error_paths(gpu_ctx_factory):
    from mcmc_amd import host
    ctx = gpu_ctx_factory(width=64, height=32)
    with pytest.raises(host.McmcError):
        ctx.frames_moments(0, 1)                      # no frames
    ctx.frames_alloc(1, 10, 10)
    with pytest.raises(host.McmcError):
        ctx.frames_moments()                          # not uploaded
    with pytest.raises(host.McmcError):
        ctx.chains_init(4)                            # no edge probabilities
    d = np.full((32, 64), 50, np.int16); l = np.full((32, 64), 0xFFFF, np.uint16)
    ctx.frame_upload(0, d, l, 3)
    with pytest.raises(host.McmcError):
        ctx.frame_set_graph(0, np.array([0, 1, 1, 2], np.uint32), np.array([7, 0], np.uint32))  # id out of range
    with pytest.raises(host.McmcError):
        ctx.frame_set_graph(0, np.array([0, 2, 2, 2], np.uint32), np.array([1, 1], np.uint32))  # duplicate neighbour
    with pytest.raises(host.McmcError):
        ctx.frame_set_graph(0, np.array([0, 2, 2, 2], np.uint32), np.array([2, 1], np.uint32))  # not ascending
    with pytest.raises(host.McmcError):
        ctx.frame_set_graph(0, np.array([0, 9, 2, 2], np.uint32), np.array([1, 2], np.uint32))  # row_ptr beyond e2
    ctx.frame_set_graph(0, np.array([0, 0, 0, 0], np.uint32), np.zeros(0, np.uint32))
    ctx.frames_moments(); ctx.frames_edge_probs()
    with pytest.raises(host.McmcError) as e:
        ctx.chains_init(2)                            # every superpixel fails the filter
    assert e.value.code == -7
    # a superpixel with more neighbours than the growth FIFO can address (MCMC_MAX_DEGREE = 32760)
    big = gpu_ctx_factory(width=64, height=32)
    big.frames_alloc(1, 33000, 70000)
    big.frame_upload(0, d, l, 33000)
    rp = np.zeros(33001, np.uint32); rp[1:] = 32761          # superpixel 0 lists 32761 neighbours, the others none
    col = np.arange(1, 32762, dtype=np.uint32)
    with pytest.raises(host.McmcError) as e:
        big.frame_set_graph(0, rp, col)
    assert e.value.code == -6
