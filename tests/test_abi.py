"""The C-ABI library must build, load and export every symbol include/rgbdfe.h declares.
(No compute calls here: this runs without a GPU.)"""
import ctypes
import os
import re

import pytest

from rgbdslam_v2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rgbdfe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbdfe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(frontend_lib):
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"librgbdfe.so does not export {n}"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_result_pod_layout(frontend_lib):
    assert frontend_lib.rgbdfe_sizeof_match_result() == ctypes.sizeof(_lib.RgbdfeMatchResult) == 1744
    assert _lib.RESULT_DTYPE.itemsize == 1744
    assert _lib.RESULT_DTYPE.fields["inlier_mask"][1] == _lib.RgbdfeMatchResult.inlier_mask.offset
    assert _lib.RESULT_DTYPE.fields["info_scale"][1] == _lib.RgbdfeMatchResult.info_scale.offset
    assert frontend_lib.rgbdfe_abi_version() == 6


def test_default_config_matches_reference_defaults(frontend_lib):
    cfg = _lib.RgbdfeConfig()
    frontend_lib.rgbdfe_default_config(ctypes.byref(cfg))
    p = cfg.params
    # parameter_server.cpp:85,86,100,101
    assert (p.max_matches, p.min_matches, p.ransac_iterations) == (300, 20, 200)
    assert p.max_dist_for_inliers == 3.0 and abs(p.depth_cov - 1e-4) < 1e-18


def test_no_cpu_fallback_without_device(frontend_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = _lib.RgbdfeConfig()
    frontend_lib.rgbdfe_default_config(ctypes.byref(cfg))
    ctx = ctypes.c_void_p()
    st = frontend_lib.rgbdfe_create(ctypes.byref(cfg), ctypes.byref(ctx))
    assert st == -2 and not ctx.value  # RGBDFE_ERR_NO_DEVICE: the product path fails loudly
    from rgbdslam_v2_amd.frontend import FrontEnd
    with pytest.raises(_lib.RgbdfeError):
        FrontEnd()


def test_default_hamming_mode_is_the_headers():
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rgbdfe.h")).read()
    m = re.search(r"#define\s+RGBDFE_HAMMING_MODE_DEFAULT\s+(\d+)", hdr)
    assert m and int(m.group(1)) == _lib.DEFAULT_HAMMING_MODE
