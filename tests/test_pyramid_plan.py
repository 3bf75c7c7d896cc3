"""CPU: the host-side plan of the fused ORB pyramid kernel (csrc/orb_host.hip plan_pyramid).

rgbdfe_debug_pyramid_plan_check builds the geometry and the plan of a workspace without touching a device, fills a pool with
pseudo-random images and masks, and compares (A) one resize per level with (B) an emulation of orb_pyramid_kernel that uses the
plan's regions, its 16-bit table entries and its LDS sizes: 0 differing bytes = every pixel of every level is written by exactly
the workgroup that owns it, from taps that lie inside what the workgroup holds in LDS.  The GPU tests (tests/test_gpu_orb.py)
run the kernel itself at 640x480 and 1280x960; this test runs the planner over the shapes nobody benchmarks."""
import ctypes

import pytest

from rgbdslam_v2_amd import _lib


def _check(cols, rows, grid, frames, seed=1):
    L = ctypes.CDLL(_lib.LIB_PATH)
    f = L.rgbdfe_debug_pyramid_plan_check
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_uint, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    n_tiles, lds = ctypes.c_int(0), ctypes.c_int(0)
    rc = f(cols, rows, grid, frames, seed, ctypes.byref(n_tiles), ctypes.byref(lds))
    return rc, n_tiles.value, lds.value


@pytest.mark.parametrize("cols,rows,grid,frames", [
    (640, 480, 1, 1), (640, 480, 1, 7), (640, 480, 0, 1), (1280, 960, 1, 2), (1280, 960, 0, 1),
    (320, 240, 1, 1), (752, 480, 1, 3), (653, 491, 1, 1), (97, 83, 0, 1), (200, 160, 1, 2), (1920, 1080, 1, 1),
    (129, 97, 0, 1), (1023, 769, 1, 1)])
def test_fused_pyramid_plan_reproduces_the_per_level_resize(cols, rows, grid, frames):
    rc, n_tiles, lds = _check(cols, rows, grid, frames, seed=cols * 7 + rows)
    assert rc == 0, "rc %d (negative: -1 geometry, -2 plan, -3 an LDS index left its buffer; positive: differing bytes)" % rc
    assert n_tiles > 0 and 0 < lds <= 64 * 1024


def test_plan_of_the_bench_shapes():
    # the super-frame workspace of bench.py's detect sub-record: 7 frames of 640x480 with the 3 x 3 grid
    rc, n_tiles, lds = _check(640, 480, 1, 7)
    assert rc == 0 and n_tiles == 679
