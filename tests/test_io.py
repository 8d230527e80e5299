"""On-disk formats: mcmc_amd/io.py against the formats of IO.cpp."""
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


def test_segments_image():
    lm = np.array([[0, 0, 1], [2, 2, 0xFFFF]], np.uint16)
    img = io.segments_image(lm, np.array([5, 5, 0xFFFF], np.uint16))
    assert img.shape == (2, 3, 3)
    assert (img[0, 0] == img[0, 2]).all() and (img[1, 0] == 0).all() and (img[1, 2] == 0).all()
