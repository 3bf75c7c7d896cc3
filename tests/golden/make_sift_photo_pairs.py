"""tests/golden/sift_photo_pairs.npz: two of the picture pairs SiftGPU ships as its own test data
(external/SiftGPU/data/640-k.jpg and 800-k.jpg are the same photograph at two sizes, k = 1, 2), decoded to 8-bit luminance
with PIL here and stored, because neither the reference tree nor a JPEG decoder is assumed at test time.  Used by
tests/test_gpu_sift_e2e.py / tools/sift_e2e.py: extraction -> projectTo3DSiftGPU -> SiftGPU matcher -> RANSAC with
features from the compiled reference pipeline on one side and from rgbdfe_sift_detect on the other.
    python tests/golden/make_sift_photo_pairs.py        (needs /root/reference)"""
import os

import numpy as np
from PIL import Image

DATA = "/root/reference/external/SiftGPU/data"
out = {}
for k in (1, 2):
    for size in (640, 800):
        g = np.asarray(Image.open(os.path.join(DATA, "%d-%d.jpg" % (size, k))).convert("L"), np.uint8)
        out["img_%d_%d" % (size, k)] = np.ascontiguousarray(g)
        print(size, k, g.shape)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sift_photo_pairs.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst))
