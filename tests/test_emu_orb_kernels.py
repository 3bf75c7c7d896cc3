"""CPU: the SOURCE of three kernels of csrc/orb_kernels.hip -- orb_pyramid_kernel, orb_resize_kernel, orb_blur_kernel -- run on
the host: the file compiled with g++ over a small HIP-on-CPU vocabulary (tests/emu/: one OS thread per HIP thread of a workgroup,
__shared__ = static storage, __syncthreads() = a barrier; wave-level operations abort, so only kernels without them run).

orb_pyramid_kernel: rgbdfe_debug_pyramid_plan_check2 builds the geometry and the plan of a workspace, fills a pool with
pseudo-random images and masks, runs one bilinear resize per level on one copy and hands the other copy, with the product's
job / tile / plan tables, to the runner passed in -- here the product's launcher over the product's kernel.  0 differing bytes
= the kernel computes every pyramid pixel the per-level path computes, with the same value.  A second test compares every level
the kernel wrote with oracle/orb_oracle.c's own cv::resize restatement of the level below; two more run the per-level resize
kernel and the 7x7 blur kernel against the oracle.  (The GPU runs of the same kernels: tests/test_gpu_orb.py;
tests/test_pyramid_plan.py checks the plan with a restatement of the kernel.)"""
import ctypes
import os
import shutil
import subprocess

import pytest

from rgbdslam_v2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("emu")
    src = open(os.path.join(ROOT, "rgbdslam_v2_amd", "csrc", "orb_kernels.hip")).read()
    decl = "extern __shared__ __attribute__((aligned(16))) uint8_t pyr_lds[];"
    assert src.count(decl) == 1
    open(os.path.join(d, "orb_kernels_emu.inc"), "w").write(src.replace(decl, "extern uint8_t pyr_lds[];"))
    lib = os.path.join(d, "libemu_orb.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-attributes",
                    "-I", os.path.join(ROOT, "tests", "emu"), "-I", str(d), "-I", os.path.join(ROOT, "rgbdslam_v2_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "emu", "emu_orb.cpp"), "-o", lib],
                   check=True, capture_output=True, timeout=300)
    return ctypes.CDLL(lib)


@pytest.mark.parametrize("cols,rows,grid,frames", [(200, 160, 1, 2), (320, 240, 0, 1), (231, 309, 1, 1), (640, 480, 1, 1)])
def test_the_kernel_source_reproduces_the_per_level_resize(emu, cols, rows, grid, frames):
    L = ctypes.CDLL(_lib.LIB_PATH)
    f = L.rgbdfe_debug_pyramid_plan_check2
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_uint, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    n_tiles, lds = ctypes.c_int(0), ctypes.c_int(0)
    runner = ctypes.cast(emu.emu_orb_pyramid, ctypes.c_void_p)
    rc = f(cols, rows, grid, frames, cols + rows, ctypes.byref(n_tiles), ctypes.byref(lds), runner)
    assert rc == 0, "rc %d (positive: differing pool bytes)" % rc
    assert n_tiles.value > 0


class _ResizeJob(ctypes.Structure):   # csrc/orb_internal.h ResizeJob
    _fields_ = [("src_off", ctypes.c_uint32), ("dst_off", ctypes.c_uint32), ("sw", ctypes.c_int32), ("sh", ctypes.c_int32),
                ("sstride", ctypes.c_int32), ("dw", ctypes.c_int32), ("dh", ctypes.c_int32), ("is_mask", ctypes.c_int32),
                ("scale_x", ctypes.c_double), ("scale_y", ctypes.c_double)]


@pytest.mark.parametrize("cols,rows,grid", [(320, 240, 1), (231, 309, 0)])
def test_the_kernel_source_against_the_oracles_resize(emu, cols, rows, grid):
    """Every level of every image chain the kernel wrote = oracle/orb_oracle.c's cv::resize restatement (its own coefficient
    tables) of the level below, masks thresholded at 254 as orb.cpp does."""
    import numpy as np
    from oracle import pyoracle as po
    po.build()
    O = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    O.orb_resize_linear_u8.restype = None
    O.orb_resize_linear_u8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int]
    L = ctypes.CDLL(_lib.LIB_PATH)
    run = L.rgbdfe_debug_pyramid_run
    run.restype = ctypes.c_long
    run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(cols * 3 + rows)
    gray = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    mask = np.where(rng.random((rows, cols)) < 0.85, 255, rng.integers(0, 256, (rows, cols))).astype(np.uint8)
    level0 = np.concatenate([gray.ravel(), mask.ravel()])
    pool = np.zeros(64 << 20, np.uint8)
    jobs = (_ResizeJob * 4096)()
    n_jobs = ctypes.c_int(0)
    n = run(cols, rows, grid, 1, level0.ctypes.data, ctypes.cast(emu.emu_orb_pyramid, ctypes.c_void_p), pool.ctypes.data,
            pool.size, ctypes.addressof(jobs), 4096, ctypes.byref(n_jobs))
    assert n > 0 and n_jobs.value > 0
    checked = 0
    for k in range(n_jobs.value):
        j = jobs[k]
        src = np.ascontiguousarray(np.lib.stride_tricks.as_strided(pool[j.src_off:], shape=(j.sh, j.sw), strides=(j.sstride, 1)))
        want = np.zeros((j.dh, j.dw), np.uint8)
        O.orb_resize_linear_u8(src.ctypes.data, j.sw, j.sh, j.sw, want.ctypes.data, j.dw, j.dh, j.dw)
        if j.is_mask:
            want[want <= 254] = 0
        got = pool[j.dst_off:j.dst_off + j.dw * j.dh].reshape(j.dh, j.dw)
        assert np.array_equal(got, want), "job %d (%dx%d -> %dx%d, mask %d)" % (k, j.sw, j.sh, j.dw, j.dh, j.is_mask)
        checked += 1
    assert checked == n_jobs.value


def _oracle():
    from oracle import pyoracle as po
    po.build()
    O = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    O.orb_resize_linear_u8.restype = None
    O.orb_resize_linear_u8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int]
    O.orb_gaussian_blur7.restype = None
    O.orb_gaussian_blur7.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return O


@pytest.mark.parametrize("sw,sh,dw,dh", [(320, 240, 267, 200), (275, 222, 229, 185), (77, 62, 64, 52), (130, 97, 108, 81)])
def test_resize_kernel_source_against_the_oracle(emu, sw, sh, dw, dh):
    """orb_resize_kernel (one launch per level: RGBDFE_ORB_PYRAMID=levels) on the host = the oracle's cv::resize, gray and mask."""
    import numpy as np
    O = _oracle()
    rng = np.random.default_rng(sw + dh)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    for is_mask in (0, 1):
        got = np.zeros((dh, dw), np.uint8)
        want = np.zeros((dh, dw), np.uint8)
        emu.emu_orb_resize(src.ctypes.data_as(ctypes.c_void_p), sw, sh, got.ctypes.data_as(ctypes.c_void_p), dw, dh, is_mask)
        O.orb_resize_linear_u8(src.ctypes.data, sw, sh, sw, want.ctypes.data, dw, dh, dw)
        if is_mask:
            want[want <= 254] = 0
        assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h", [(320, 240), (131, 97), (64, 16), (70, 33)])
def test_blur_kernel_source_against_the_oracle(emu, w, h):
    """orb_blur_kernel on the host = the oracle's GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in 8-bit fixed point."""
    import numpy as np
    O = _oracle()
    rng = np.random.default_rng(w * 5 + h)
    src = rng.integers(0, 256, (h, w), dtype=np.uint8)
    got = np.zeros((h, w), np.uint8)
    want = np.zeros((h, w), np.uint8)
    emu.emu_orb_blur(src.ctypes.data_as(ctypes.c_void_p), w, h, got.ctypes.data_as(ctypes.c_void_p))
    O.orb_gaussian_blur7(src.ctypes.data, w, h, w, want.ctypes.data)
    assert np.array_equal(got, want)
