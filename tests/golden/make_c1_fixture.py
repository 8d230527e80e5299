"""Config 1 fixture: the reference's bundled Kinect frame as raw arrays.

Reads /root/reference/color1.png (640x480 RGB8) and depth1.png (640x480, 16-bit depth in
1/5000 m, 0 = no reading) with PIL -- data files, nothing is executed -- and writes
tests/golden/c1_frame.npz:
  colors  uint8 [480][640][3] in cv::Vec3b order (B, G, R), what cv::imread hands the reference
  disp    int16 [480][640]    disparity

The reference's own PNG route is dead code (SimpleSeg.cpp:88-101 is compiled out and reads
an 8-bit image as short), so the depth -> disparity conversion is the
build's definition: d = round(8 f B / z), z = raw / 5000 m, f = 580 px, B = 0.10 m (the 1/8
pixel disparity units of the Kinect), 0 where raw == 0.
"""
import os
import sys

import numpy as np
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

rgb = np.asarray(Image.open(os.path.join(REF, "color1.png")).convert("RGB"), np.uint8)
raw = np.asarray(Image.open(os.path.join(REF, "depth1.png"))).astype(np.int64)
assert rgb.shape == (480, 640, 3) and raw.shape == (480, 640), (rgb.shape, raw.shape)
z = raw / 5000.0
with np.errstate(divide="ignore"):
    d = np.where(raw > 0, np.rint(8.0 * 580.0 * 0.10 / z), 0.0)
assert d.max() < 2048, d.max()
np.savez_compressed(os.path.join(HERE, "c1_frame.npz"), colors=np.ascontiguousarray(rgb[:, :, ::-1]),
                    disp=d.astype(np.int16))
print("zeros %.1f%%  disparity range %d..%d" % (100.0 * (raw == 0).mean(), d[d > 0].min(), d.max()))
