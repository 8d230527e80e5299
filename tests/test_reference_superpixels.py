"""The superpixel layout the reference's own run left behind (output/mapping.txt -> tests/golden/ref_superpixels.npz,
made by tests/golden/make_c1_fixture.py): 177 superpixels of a real 640x480 Kinect frame, from 10 to 64 506 pixels,
as the A matrices ([u v 1] rows) createSuperPixels built (OverSegmentation.cpp:89-194).  The frame itself is not in
the reference, so disparities are synthetic here; what the file pins is the shape of the rows (A1): row-major pixel
order, ids = a member pixel in ascending order, and that everything the reference kept passes the A1 filter."""
import os

import numpy as np
import pytest

from . import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref_sp():
    z = np.load(os.path.join(HERE, "golden", "ref_superpixels.npz"))
    label = np.ascontiguousarray(z["label"])
    S = int(z["ids"].size)
    # a tilted plane per superpixel plus noise, in Kinect 1/8-pixel units; unlabelled pixels get a value too
    rng = np.random.default_rng(7)
    v, u = np.mgrid[0:480, 0:640]
    a = rng.uniform(-0.3, 0.3, S + 1); b = rng.uniform(-0.3, 0.3, S + 1); c = rng.uniform(300, 900, S + 1)
    li = np.where(label == 0xFFFF, S, label).astype(np.int64)
    disp = np.rint(a[li] * (u - 320) + b[li] * (v - 240) + c[li] + rng.normal(0, 1.5, label.shape)).astype(np.int16)
    return dict(z=z, label=label, S=S, disp=disp)


def _moments_np(disp, label, S):
    # N Su Sv Suu Suv Svv Sb Sub Svb Sbb in exact integers
    v, u = np.mgrid[0:disp.shape[0], 0:disp.shape[1]]
    m = label != 0xFFFF
    l = label[m].astype(np.int64); uu = u[m].astype(np.int64); vv = v[m].astype(np.int64); bb = disp[m].astype(np.int64)
    out = np.zeros((S, 10), np.int64)
    for k, c in enumerate([np.ones_like(uu), uu, vv, uu * uu, uu * vv, vv * vv, bb, uu * bb, vv * bb, bb * bb]):
        np.add.at(out[:, k], l, c)
    return out


def test_what_the_file_shows(ref_sp):
    z = ref_sp["z"]
    assert bool(z["row_major_rows"]) and bool(z["id_is_member"]) and bool(z["third_column_ones"])
    ids = z["ids"]
    assert ids.size == 177 and (np.diff(ids) > 0).all()                  # std::map order = ascending id
    n = np.bincount(ref_sp["label"][ref_sp["label"] != 0xFFFF].astype(np.int64), minlength=177)
    assert n.min() == 10 and n.max() == 64506 and n.sum() == 293425      # nothing below the N >= 10 bar (OverSegmentation.cpp:394-420)


def test_oracle_on_the_reference_layout(ref_sp):
    S, label, disp = ref_sp["S"], ref_sp["label"], ref_sp["disp"]
    mom = O.moments(disp, label, S)
    assert np.array_equal(mom, _moments_np(disp, label, S))
    fit, valid, lerr = O.local_fit(mom, 640, 480)
    assert valid.all()                                                   # everything the reference kept passes the A1 filter
    # the plane each superpixel was drawn from comes back (noise sigma 1.5 -> residual per pixel ~2.25 / (2 var))
    assert np.isfinite(fit).all() and (lerr >= 0).all()


@pytest.mark.gpu
def test_gpu_on_the_reference_layout(ref_sp, gpu_ctx_factory):
    S, label, disp = ref_sp["S"], ref_sp["label"], ref_sp["disp"]
    ctx = gpu_ctx_factory(width=640, height=480)
    ctx.frames_alloc(1, S, 8)
    ctx.frame_upload(0, disp, label, S)
    ctx.frame_set_graph(0, np.zeros(S + 1, np.uint32), np.zeros(0, np.uint32))
    ctx.frames_moments()
    mom = O.moments(disp, label, S)
    assert np.array_equal(ctx.moments(0), mom)
    fit, valid, lerr = O.local_fit(mom, 640, 480)
    gfit, gvalid, glerr = ctx.local_fit(0)
    assert np.array_equal(gfit.view(np.uint64), fit.view(np.uint64)) and np.array_equal(gvalid, valid)
    assert np.array_equal(glerr.view(np.uint64), lerr.view(np.uint64))
