"""tests/golden/sift_extract_golden.npz: outputs of the REFERENCE's SIFT extraction pipeline (SiftGPU's CUDA kernels and
host code compiled from /root/reference on the CPU emulation, oracle/_ref/libref_siftgpu.so; see oracle/Makefile) on two
seeded synthetic images, so that the GPU box checks rgbdfe_sift_detect against the reference without the reference tree:
keys (x, y, scale, orientation), descriptors, features per (octave, dog level), every level's keypoint candidates
(x, y, sign, dx, dy, ds) and a CRC of every Gaussian plane.  Run here (needs /root/reference for the pin):
    python tests/golden/make_sift_extract_golden.py"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from rgbdslam_v2_amd import synth  # noqa: E402

CASES = [("a", 200, 152, 400, 3), ("b", 322, 241, 150, 5)]   # name, width, height, max_keypoints, image seed


def image(w, h, seed):
    return synth.make_image_sequence(n_frames=1, seed=seed, width=w, height=h)["gray"][0]


def main():
    out = {}
    for name, w, h, maxf, seed in CASES:
        g = image(w, h, seed)
        keys, desc, cnt = po.ref_sift_detect(g, maxf)
        geo = po.ref_sift_geometry()
        out[name + "_meta"] = np.array([w, h, maxf, seed, geo["octave_min"], geo["octave_num"]], np.int32)
        out[name + "_keys"], out[name + "_desc"], out[name + "_counts"] = keys, desc, cnt
        crcs = []
        for o in range(geo["octave_num"]):
            for l in range(geo["levels"]):
                crcs.append(zlib.crc32(po.ref_sift_level(o, l, 0).tobytes()))
            for j in range(geo["dog_levels"]):
                out["%s_cand_%d_%d" % (name, o, j)] = po.ref_sift_candidates(o, j)
        out[name + "_plane_crc"] = np.array(crcs, np.uint32)
        print(name, w, h, "features", len(keys), "levels", cnt)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sift_extract_golden.npz"), **out)


if __name__ == "__main__":
    main()
