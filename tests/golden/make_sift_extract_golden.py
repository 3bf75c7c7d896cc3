"""tests/golden/sift_extract_golden.npz: outputs of the REFERENCE's SIFT extraction pipeline (SiftGPU's CUDA kernels and
host code compiled from /root/reference on the CPU emulation, oracle/_ref/libref_siftgpu.so; see oracle/Makefile) on two
seeded synthetic images and on two of the photographs SiftGPU ships as its own test data (external/SiftGPU/data/640-1.jpg
whole, a 320 x 240 window of 800-2.jpg; decoded to 8-bit luminance with PIL here and stored in the fixture, because neither
the reference tree nor a JPEG decoder is assumed at test time), so that the GPU box checks rgbdfe_sift_detect against the
reference without the reference tree:
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
DATA = "/root/reference/external/SiftGPU/data"
# name, file, window (x0, y0, w, h) or None, max_keypoints, store the candidate lists (else their CRCs)
PHOTOS = [("c", "640-1.jpg", None, 1000, False), ("d", "800-2.jpg", (240, 180, 320, 240), 400, True)]


def image(w, h, seed):
    return synth.make_image_sequence(n_frames=1, seed=seed, width=w, height=h)["gray"][0]


def photo(fname, win):
    from PIL import Image
    g = np.asarray(Image.open(os.path.join(DATA, fname)).convert("L"), np.uint8)
    if win is not None:
        x0, y0, w, h = win
        g = g[y0:y0 + h, x0:x0 + w]
    return np.ascontiguousarray(g)


def main():
    out = {}
    jobs = [(n, image(w, h, seed), maxf, seed, True, False) for n, w, h, maxf, seed in CASES]
    jobs += [(n, photo(f, win), maxf, -1, lists, True) for n, f, win, maxf, lists in PHOTOS]
    for name, g, maxf, seed, lists, keep_image in jobs:
        h, w = g.shape
        if keep_image:
            out[name + "_img"] = g
        keys, desc, cnt = po.ref_sift_detect(g, maxf)
        geo = po.ref_sift_geometry()
        out[name + "_meta"] = np.array([w, h, maxf, seed, geo["octave_min"], geo["octave_num"]], np.int32)
        out[name + "_keys"], out[name + "_desc"], out[name + "_counts"] = keys, desc, cnt
        crcs, ccrc, ccnt = [], [], []
        for o in range(geo["octave_num"]):
            for l in range(geo["levels"]):
                crcs.append(zlib.crc32(po.ref_sift_level(o, l, 0).tobytes()))
            for j in range(geo["dog_levels"]):
                c = po.ref_sift_candidates(o, j)
                if lists:
                    out["%s_cand_%d_%d" % (name, o, j)] = c
                ccrc.append(zlib.crc32(c.tobytes()))
                ccnt.append(len(c))
        out[name + "_plane_crc"] = np.array(crcs, np.uint32)
        out[name + "_cand_crc"], out[name + "_cand_n"] = np.array(ccrc, np.uint32), np.array(ccnt, np.int32)
        print(name, w, h, "features", len(keys), "levels", cnt)
    # the wrapper's second mode (a caller-provided keypoint list, sift_gpu_wrapper.cpp:132-142): descriptors of the reference
    # pipeline for given (x, y, scale, orientation) -- scales across all bands, below the first and above the last one
    g = image(322, 241, 5)
    rng = np.random.default_rng(12)
    n = 300
    keys = np.stack([rng.uniform(8, 314, n), rng.uniform(8, 233, n), np.exp(rng.uniform(np.log(0.6), np.log(40.0), n)),
                     rng.uniform(0, 2 * np.pi, n)], 1).astype(np.float32)
    out["k_keys"] = keys
    out["k_desc"] = po.ref_sift_describe(g, keys)
    print("keypoint-list mode:", n, "keypoints, descriptor norms", float(np.linalg.norm(out["k_desc"], axis=1).mean()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sift_extract_golden.npz"), **out)


if __name__ == "__main__":
    main()
