"""Config 1 fixtures: the reference's bundled Kinect frames as raw arrays, its parameter file, and the superpixel
layout its own run left behind.

Reads /root/reference/color{1,2,3}.png (640x480 RGB8) and depth{1,2,3}.png (640x480, 16-bit depth in 1/5000 m,
0 = no reading) with mcmc_amd.io.load_frame_png (PIL; data files, nothing is executed) and writes
  tests/golden/c1_frame.npz, c2_frame.npz, c3_frame.npz:
    colors  uint8 [480][640][3] in cv::Vec3b order (B, G, R), what cv::imread hands the reference
    disp    int16 [480][640]    disparity
The reference's own PNG route is dead code (SimpleSeg.cpp:88-101 is compiled out and reads an 8-bit image as short,
SURVEY.md section 7), so the depth -> disparity conversion is the build's definition: d = round(8 f B / z),
z = raw / 5000 m, f = 580 px, B = 0.10 m (the 1/8 pixel disparity units of the Kinect), 0 where raw == 0.

  tests/golden/overSegmentationParams.txt: a byte-for-byte copy of the reference's output/overSegmentationParams.txt
  (a data file: seven values with "# name" comments).

  tests/golden/ref_superpixels.npz: output/mapping.txt of the reference, the A matrix ([u v 1] rows) of every
  superpixel that an over-segmentation run of the authors kept, as
    ids     int64 [177]        the superpixels' ids in file order (the union-find root pixel, v * 640 + u)
    label   uint16 [480][640]  index into ids, 0xFFFF where no kept superpixel covers the pixel
    row_major_rows, id_is_member, third_column_ones: what the file shows about createSuperPixels' output
  (the frame it came from is not in the repository: output/colorf.png is none of color{1,2,3}.png).
"""
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mcmc_amd import io  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

for k in (1, 2, 3):
    d, col = io.load_frame_png(os.path.join(REF, f"color{k}.png"), os.path.join(REF, f"depth{k}.png"))
    assert col.shape == (480, 640, 3) and d.shape == (480, 640) and d.max() < 2048
    np.savez_compressed(os.path.join(HERE, f"c{k}_frame.npz"), colors=col, disp=d)
    print("frame %d: zeros %.1f%%  disparity range %d..%d" % (k, 100.0 * (d == 0).mean(), d[d > 0].min(), d.max()))

shutil.copyfile(os.path.join(REF, "output", "overSegmentationParams.txt"), os.path.join(HERE, "overSegmentationParams.txt"))

parts = re.split(r"(\d+)->", open(os.path.join(REF, "output", "mapping.txt")).read())
ids = np.array([int(x) for x in parts[1::2]], np.int64)
label = np.full((480, 640), 0xFFFF, np.uint16)
row_major = id_member = ones = True
for n, body in enumerate(parts[2::2]):
    a = np.array(body.split(), dtype=np.float64).reshape(-1, 3)
    u, v = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    assert (label[v, u] == 0xFFFF).all()
    label[v, u] = n
    row_major &= bool((np.diff(v * 640 + u) > 0).all())
    ones &= bool((a[:, 2] == 1.0).all())
for n, i in enumerate(ids):
    id_member &= bool(label[i // 640, i % 640] == n)
np.savez_compressed(os.path.join(HERE, "ref_superpixels.npz"), ids=ids, label=label, row_major_rows=row_major,
                    id_is_member=id_member, third_column_ones=ones)
print("mapping.txt: %d superpixels, %d pixels covered, rows row-major %s, id is a member pixel %s" %
      (ids.size, int((label != 0xFFFF).sum()), row_major, id_member))
