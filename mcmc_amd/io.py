"""On-disk formats of the reference, host side.

* raw frame text, IO::readRawData / writeRawData (src/libraries/IO.cpp:6-48): one pixel per line in
  row-major order, "<disparity>\\t<colour>\\n", colour packed r << 16 | g << 8 | b; read back into
  int16 disparities and cv::Vec3b (B, G, R) colours.
* over-segmentation parameters, IO::readSegmentationParams (IO.cpp:105-134): seven lines, value
  then "# name": windowSize, spatialStdev, disparityStdev, weightThreshold, colorFactor,
  disparityFactor, distanceFactor.
* a Kinect frame as PNG pair (the reference bundles color{1,2,3}.png / depth{1,2,3}.png; its own PNG route,
  SimpleSeg.cpp:88-101, is compiled out): load_frame_png.
* a segmentation as an image, Problem::visualize (Problem.cpp:305-370): white background, one colour per segment
  over its superpixels' pixels (or the angle between the superpixel's and the segment's plane normals), a component
  in black, superpixel boundaries in black: visualize().  segments_image() is the plain palette image.
"""
from __future__ import annotations

import numpy as np


def write_raw_data(path, disp, colors):
    """disp (H, W) int16, colors (H, W, 3) uint8 in B, G, R order."""
    disp = np.asarray(disp); colors = np.asarray(colors)
    packed = (colors[..., 2].astype(np.uint64) << 16) | (colors[..., 1].astype(np.uint64) << 8) | colors[..., 0].astype(np.uint64)
    # writeRawData prints the short through a size_t (IO.cpp:39): a negative disparity wraps to 2^64 - |d|
    d = disp.astype(np.int64).astype(np.uint64)
    with open(path, "w") as f:
        for a, b in zip(d.ravel().tolist(), packed.ravel().tolist()):
            f.write(f"{a}\t{b}\n")


def read_raw_data(path, width=640, height=480):
    """Returns (disp int16 (H, W), colors uint8 (H, W, 3) B, G, R)."""
    data = np.loadtxt(path, dtype=np.float64, ndmin=2)
    if data.shape[0] < width * height or data.shape[1] < 2:
        raise ValueError(f"{path}: expected {width * height} lines of two numbers, found {data.shape}")
    data = data[: width * height]
    disp = data[:, 0].astype(np.int64).astype(np.int16).reshape(height, width)       # (short) data(i, 0)
    c = data[:, 1].astype(np.uint64)
    colors = np.stack([(c & 0xFF), (c >> 8) & 0xFF, (c >> 16) & 0xFF], axis=1).astype(np.uint8).reshape(height, width, 3)
    return disp, colors


# depth -> disparity of load_frame_png: the 1/8-pixel disparity units of the Kinect, d = round(8 f B / z)
KINECT_FOCAL_PX = 580.0
KINECT_BASELINE_M = 0.10
DEPTH_PNG_UNITS_PER_M = 5000.0


def depth_to_disparity(raw):
    """16-bit depth image values (1/5000 m, 0 = no reading) -> int16 disparity in 1/8 pixel, 0 where there is no
    reading.  The reference receives disparities from the sensor driver (SimpleSeg.cpp:137-192); for the bundled
    PNG frames this conversion is the build's definition (tests/golden/make_c1_fixture.py uses it)."""
    raw = np.asarray(raw).astype(np.int64)
    z = raw / DEPTH_PNG_UNITS_PER_M
    with np.errstate(divide="ignore"):
        d = np.where(raw > 0, np.rint(8.0 * KINECT_FOCAL_PX * KINECT_BASELINE_M / z), 0.0)
    if d.max(initial=0.0) > 32767:
        raise ValueError("disparity does not fit int16")
    return d.astype(np.int16)


def load_frame_png(color_path, depth_path):
    """(disp int16 (H, W), colors uint8 (H, W, 3) in cv::Vec3b order B, G, R) from an RGB8 colour PNG and a
    16-bit depth PNG of the same size (what cv::imread would hand OverSegmentation::overSegment)."""
    from PIL import Image           # data files only; nothing is executed
    rgb = np.asarray(Image.open(color_path).convert("RGB"), np.uint8)
    raw = np.asarray(Image.open(depth_path))
    if raw.ndim != 2 or rgb.shape[:2] != raw.shape:
        raise ValueError(f"colour {rgb.shape} and depth {raw.shape} images do not match")
    return depth_to_disparity(raw), np.ascontiguousarray(rgb[:, :, ::-1])


PARAM_NAMES = ("window_size", "spatial_stdev", "disparity_stdev", "weight_threshold", "color_factor",
               "disparity_factor", "distance_factor")


def read_segmentation_params(path):
    """Dict with the seven OverSegmentationParameters; raises on a missing line, as the reference does."""
    with open(path) as f:
        lines = f.read().split("\n")
    if len(lines) < 7 or any(l.strip() == "" for l in lines[:7]):
        raise ValueError("Missing line")
    out = {}
    for name, line in zip(PARAM_NAMES, lines[:7]):
        h = line.find("#")
        text = line[: h - 1] if h >= 0 else line          # substr(0, find('#') - 1), IO.cpp:121-127
        out[name] = _atoi(text) if name in ("window_size", "weight_threshold") else _atof(text)
    return out


def _atof(t):
    import re
    m = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", t)
    return float(m.group(0)) if m else 0.0


def _atoi(t):
    import re
    m = re.match(r"\s*[-+]?\d+", t)
    return int(m.group(0)) if m else 0


def overseg_params(d):
    """mcmc_overseg_params from the dict of read_segmentation_params."""
    from . import host
    p = host.OversegParams()
    for k in PARAM_NAMES:
        setattr(p, k, d[k])
    return p


def segments_image(label_map, segment_of_superpixel, seed=0):
    """(H, W, 3) uint8 image: one colour per segment, black where the superpixel was dropped
    (label 0xFFFF) -- the role of Problem::printResults' result image."""
    seg = np.asarray(segment_of_superpixel)
    lm = np.asarray(label_map).astype(np.int64)
    rng = np.random.default_rng(seed)
    palette = rng.integers(32, 256, (int(seg[seg != 0xFFFF].max(initial=0)) + 2, 3), dtype=np.uint8)
    s = np.where(lm < seg.size, seg[np.minimum(lm, seg.size - 1)], 0xFFFF)
    img = palette[np.minimum(s, palette.shape[0] - 1)]
    img[s == 0xFFFF] = 0
    return img


KINECT_FOCAL = 580.0        # kFocalLength, MyMath.h:30
KINECT_BASELINE = 0.10      # kBaseline, MyMath.h:29


def disparity_to_explicit(theta):
    """MyMath.h:94-103: (a, b, c) of disparity = a u + b v + c (raw pixel coordinates) -> unit normal and distance
    (nx, ny, nz, d).  theta: (..., 3)."""
    theta = np.asarray(theta, np.float64)
    a, b, c = theta[..., 0], theta[..., 1], theta[..., 2]
    f = KINECT_FOCAL
    n = np.sqrt(a * a + b * b + (c / f) * (c / f))
    return np.stack([a / n, b / n, c / (f * n), KINECT_BASELINE / n], axis=-1)


def raw_theta(theta_centred, width, height):
    """The library fits planes in coordinates centred at (W/2, H/2); the reference's are raw
    pixel coordinates: same a, b, and c_raw = c - a W/2 - b H/2."""
    t = np.array(theta_centred, np.float64, copy=True)
    t[..., 2] -= t[..., 0] * (width // 2) + t[..., 1] * (height // 2)
    return t


def boundary_pixels(label_map):
    """createSuperPixels' boundaryPixels_ (OverSegmentation.cpp:174-190): a pixel whose superpixel differs from that of
    its LEFT, TOP or TOP-RIGHT neighbour (the TOP-LEFT compare never registers: the ring buffer slot it reads has just
    been overwritten with the pixel's own id, :174).  Bool (H, W)."""
    lm = np.asarray(label_map)
    out = np.zeros(lm.shape, bool)
    out[:, 1:] |= lm[:, 1:] != lm[:, :-1]
    out[1:, :] |= lm[1:, :] != lm[:-1, :]
    out[1:, :-1] |= lm[1:, :-1] != lm[:-1, 1:]
    return out


def visualize(label_map, segment_of_superpixel, theta_superpixel=None, theta_segment=None, component=None,
              draw_error=False, seed=0):
    """Problem::visualize (Problem.cpp:305-370) as an (H, W, 3) uint8 image in cv::Vec3b order (B, G, R).

    label_map (H, W) uint16 superpixel ids; segment_of_superpixel [S] segment id per superpixel, 0xFFFF = not
    assigned (dropped by the A1 filter: stays white, as superpixels outside assignments_ do).  Without draw_error
    every segment gets one colour (the reference draws rand() % 256 three times per segment, :343; here a seeded
    generator).  With draw_error the colour is (e, e, 255), e = (size_t)(255 (1 - angle / (pi / 2))), angle between
    the normals of disparityToExplicit(segment plane) and disparityToExplicit(superpixel plane), :323-334 -- thetas in
    raw pixel coordinates (raw_theta), theta_segment indexed by segment id.  Superpixels of `component` are drawn
    black ((125, 0, 125) with draw_error, :347-350); boundary pixels are black (:361-367)."""
    lm = np.asarray(label_map).astype(np.int64)
    seg = np.asarray(segment_of_superpixel).astype(np.int64)
    S = seg.size
    assigned = seg != 0xFFFF
    if draw_error:
        ns = disparity_to_explicit(np.asarray(theta_superpixel, np.float64))[:, :3]
        ng = disparity_to_explicit(np.asarray(theta_segment, np.float64))[np.where(assigned, seg, 0), :3]
        with np.errstate(invalid="ignore"):
            angle = np.arccos(np.sum(ns * ng, axis=1))
        e = 255.0 * (1.0 - angle / (np.pi / 2.0))
        e = np.where(np.isfinite(e), e, 0.0)
        e = np.clip(np.trunc(e), 0, 255).astype(np.uint8)          # (size_t) cast, then cv::saturate_cast<uchar>
        col = np.stack([e, e, np.full(S, 255, np.uint8)], axis=1)
    else:
        rng = np.random.default_rng(seed)
        palette = rng.integers(0, 256, (int(seg[assigned].max(initial=0)) + 1, 3)).astype(np.uint8)
        col = palette[np.where(assigned, seg, 0)]
    if component is not None:
        col = col.copy()
        col[np.asarray(component, np.int64)] = (125, 0, 125) if draw_error else (0, 0, 0)
    img = np.full(lm.shape + (3,), 255, np.uint8)
    inside = lm < S
    sp = np.where(inside, lm, 0)
    paint = inside & assigned[sp]
    img[paint] = col[sp[paint]]
    img[boundary_pixels(label_map)] = 0
    return img
