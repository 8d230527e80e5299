"""On-disk formats: mcmc_amd/io.py against the formats of IO.cpp."""
import os

import numpy as np
import pytest

from mcmc_amd import io


def test_raw_data_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    disp = rng.integers(0, 2048, (6, 9)).astype(np.int16)
    col = rng.integers(0, 256, (6, 9, 3)).astype(np.uint8)
    p = tmp_path / "raw.txt"
    io.write_raw_data(p, disp, col)
    lines = p.read_text().split("\n")
    # IO.cpp:41-42: "<disp>\t<r << 16 | g << 8 | b>\n" with colours held as (B, G, R)
    assert lines[0] == f"{disp[0, 0]}\t{(int(col[0, 0, 2]) << 16) | (int(col[0, 0, 1]) << 8) | int(col[0, 0, 0])}"
    assert len(lines) == 6 * 9 + 1 and lines[-1] == ""
    d2, c2 = io.read_raw_data(p, width=9, height=6)
    assert np.array_equal(d2, disp) and np.array_equal(c2, col)
    with pytest.raises(ValueError):
        io.read_raw_data(p, width=10, height=6)


def test_segmentation_params_file(tmp_path):
    # the text of the reference's output/overSegmentationParams.txt layout: value, tab or space, "# name"
    p = tmp_path / "params.txt"
    p.write_text("9\t# windowSize \n5.0\t# spatialStdev \n3.0\t# disparityStdev\n300\t# weightThreshold\n"
                 "1.0 # colorFactor\n1.0 # disparityFactor\n1.0\t# distanceFactor\n")
    d = io.read_segmentation_params(p)
    assert d == dict(window_size=9, spatial_stdev=5.0, disparity_stdev=3.0, weight_threshold=300,
                     color_factor=1.0, disparity_factor=1.0, distance_factor=1.0)
    p.write_text("9\t# windowSize \n5.0\t# spatialStdev \n")
    with pytest.raises(ValueError):
        io.read_segmentation_params(p)                    # "Missing line", IO.cpp:117


def test_reference_params_file():
    # the reference's own output/overSegmentationParams.txt (tests/golden/make_c1_fixture.py copies it byte for byte):
    # tabs and spaces before '#', trailing blanks, the values of BASELINE config 1
    d = io.read_segmentation_params(os.path.join(os.path.dirname(__file__), "golden", "overSegmentationParams.txt"))
    assert d == dict(window_size=9, spatial_stdev=5.0, disparity_stdev=3.0, weight_threshold=300,
                     color_factor=1.0, disparity_factor=1.0, distance_factor=1.0)


def test_depth_to_disparity_and_png_loader(tmp_path):
    from PIL import Image
    # 1/5000 m units, d = round(8 * 580 * 0.10 / z): 1 m -> 464, 2 m -> 232, 0 stays 0, halves round to even (np.rint)
    raw = np.array([[5000, 10000, 0], [4640, 2320, 65535]], np.uint16)
    want = np.array([[464, 232, 0], [500, 1000, 35]], np.int16)
    assert np.array_equal(io.depth_to_disparity(raw), want)
    with pytest.raises(ValueError):
        io.depth_to_disparity(np.array([[1]], np.uint16))          # 2 320 000 does not fit int16
    rgb = np.arange(18, dtype=np.uint8).reshape(2, 3, 3)
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.png")
    Image.fromarray(raw).save(tmp_path / "d.png")
    disp, col = io.load_frame_png(tmp_path / "c.png", tmp_path / "d.png")
    assert np.array_equal(disp, want) and np.array_equal(col, rgb[:, :, ::-1])      # cv::Vec3b is B, G, R
    Image.fromarray(raw[:, :2]).save(tmp_path / "d2.png")
    with pytest.raises(ValueError):
        io.load_frame_png(tmp_path / "c.png", tmp_path / "d2.png")


def test_bundled_frames_are_the_loader_output():
    # the three bundled frames as fixtures: what load_frame_png made of color{1,2,3}.png / depth{1,2,3}.png
    for k in (1, 2, 3):
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"c{k}_frame.npz"))
        assert z["disp"].shape == (480, 640) and z["disp"].dtype == np.int16 and z["colors"].shape == (480, 640, 3)
        d = z["disp"]
        assert 0.10 < (d == 0).mean() < 0.16 and d[d > 0].min() >= 120 and d.max() < 600


def test_segments_image():
    lm = np.array([[0, 0, 1], [2, 2, 0xFFFF]], np.uint16)
    img = io.segments_image(lm, np.array([5, 5, 0xFFFF], np.uint16))
    assert img.shape == (2, 3, 3)
    assert (img[0, 0] == img[0, 2]).all() and (img[1, 0] == 0).all() and (img[1, 2] == 0).all()


def test_visualize_matches_the_reference_rendering():
    # Problem::visualize (Problem.cpp:305-370): white where nothing is assigned, one colour per segment, component in
    # black, boundary pixels (LEFT / TOP / TOP-RIGHT differ) in black
    lm = np.array([[0, 0, 1, 1],
                   [0, 0, 1, 1],
                   [2, 2, 2, 3]], np.uint16)
    seg = np.array([7, 7, 0xFFFF, 4], np.uint16)
    b = io.boundary_pixels(lm)
    want_b = np.array([[0, 0, 1, 0],
                       [0, 1, 1, 0],          # (1,1): its TOP-RIGHT neighbour (0,2) is superpixel 1
                       [1, 1, 1, 1]], bool)
    assert np.array_equal(b, want_b)
    img = io.visualize(lm, seg, seed=3)
    assert (img[0, 0] == img[0, 3]).all() and not (img[0, 0] == 255).all()      # superpixels 0 and 1 share segment 7
    assert (img[b] == 0).all()
    img2 = io.visualize(lm, seg, component=[1], seed=3)
    assert (img2[0, 3] == 0).all() and (img2[0, 0] == img[0, 0]).all()
    # superpixel 2 is not assigned: white wherever it is not boundary (none here: the whole row is boundary)
    lm2 = np.zeros((4, 4), np.uint16); lm2[2:] = 1
    assert (io.visualize(lm2, np.array([0xFFFF, 0], np.uint16))[0] == 255).all()
    # error colouring: identical planes -> angle 0 -> (255, 255, 255); orthogonal normals -> e = 0
    th = np.array([[0.1, 0.2, 300.0], [0.1, 0.2, 300.0]])
    e = io.visualize(lm2, np.array([0, 0], np.uint16), theta_superpixel=th, theta_segment=th[:1], draw_error=True)
    assert e[0, 0, 2] == 255 and e[0, 0, 0] == e[0, 0, 1] and e[0, 0, 0] >= 254      # acos of a dot product within an ulp of 1, truncated
    n = io.disparity_to_explicit(np.array([0.0, 0.0, 580.0]))
    assert np.allclose(n, [0.0, 0.0, 1.0, 0.1])                                   # MyMath.h:94-103 with f = 580, B = 0.10
    assert np.allclose(io.raw_theta(np.array([1.0, 2.0, 10.0]), 640, 480), [1.0, 2.0, 10.0 - 320.0 - 480.0])
