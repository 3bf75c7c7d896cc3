"""ctypes binding of the ORB restatement in oracle/orb_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C

import numpy as np

from . import pyoracle as _po

ORB_MAX_CELLS = 64
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])


class GridState(C.Structure):
    _fields_ = [("grid", C.c_int32), ("max_iters", C.c_int32), ("cell_min", C.c_int32),
                ("cell_max", C.c_int32), ("max_total", C.c_int32), ("edge", C.c_int32),
                ("thresh", C.c_double * ORB_MAX_CELLS)]


_ready = False


def lib():
    global _ready
    L = _po.lib()
    if not _ready:
        vp, i = C.c_void_p, C.c_int
        L.orb_resize_linear_u8.restype = None
        L.orb_resize_linear_u8.argtypes = [vp, i, i, i, vp, i, i, i]
        L.orb_level_geometry.restype = None
        L.orb_level_geometry.argtypes = [i, i, i, vp, vp, vp]
        L.orb_fast_score_map.restype = None
        L.orb_fast_score_map.argtypes = [vp, i, i, i, i, vp]
        L.orb_fast_keypoints.restype = i
        L.orb_fast_keypoints.argtypes = [vp, vp, i, i, i, vp, i]
        L.orb_harris_at.restype = C.c_float
        L.orb_harris_at.argtypes = [vp, i, i, i]
        L.orb_fast_atan2.restype = C.c_float
        L.orb_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orb_umax.restype = None
        L.orb_umax.argtypes = [vp]
        L.orb_ic_angle_at.restype = C.c_float
        L.orb_ic_angle_at.argtypes = [vp, i, i, i, vp]
        L.orb_detect.restype = i
        L.orb_detect.argtypes = [vp, vp, i, i, i, i, i, vp, i]
        L.orb_grid_state_init.restype = None
        L.orb_grid_state_init.argtypes = [C.POINTER(GridState), i, i, i]
        L.orb_grid_detect.restype = i
        L.orb_grid_detect.argtypes = [C.POINTER(GridState), vp, vp, i, i, vp, i]
        L.orb_gaussian_blur7.restype = None
        L.orb_gaussian_blur7.argtypes = [vp, i, i, i, vp]
        L.orb_gauss7_kernel_fixed.restype = None
        L.orb_gauss7_kernel_fixed.argtypes = [vp]
        L.orb_compute.restype = i
        L.orb_compute.argtypes = [vp, i, i, vp, i, vp]
        L.orb_node_features.restype = i
        L.orb_node_features.argtypes = [C.POINTER(GridState), vp, vp, vp, i, i, i, vp, i, vp]
        L.orb_pattern.restype = C.POINTER(C.c_int8)
        _ready = True
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def level_geometry(cols, rows, nlevels=8):
    sc = np.zeros(nlevels, np.float32)
    lw = np.zeros(nlevels, np.int32)
    lh = np.zeros(nlevels, np.int32)
    lib().orb_level_geometry(cols, rows, nlevels, _p(sc), _p(lw), _p(lh))
    return sc, lw, lh


def resize(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().orb_resize_linear_u8(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(out), dw, dh, dw)
    return out


def pyramid(img, mask=None, nlevels=8):
    """Levels 0..nlevels-1 of the ORB image (and mask) pyramid."""
    img = np.ascontiguousarray(img, np.uint8)
    sc, lw, lh = level_geometry(img.shape[1], img.shape[0], nlevels)
    imgs, masks = [img], [None if mask is None else np.ascontiguousarray(mask, np.uint8)]
    for l in range(1, nlevels):
        imgs.append(resize(imgs[-1], int(lw[l]), int(lh[l])))
        if mask is not None:
            m = resize(masks[-1], int(lw[l]), int(lh[l]))
            m[m <= 254] = 0
            masks.append(m)
        else:
            masks.append(None)
    return imgs, masks, sc


def fast_score_map(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().orb_fast_score_map(_p(img), img.shape[1], img.shape[0], img.shape[1], threshold, _p(out))
    return out


def fast_keypoints(score, mask, edge):
    score = np.ascontiguousarray(score, np.uint8)
    cap = score.size // 4 + 16
    kp = np.zeros(cap, KP_DTYPE)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    n = lib().orb_fast_keypoints(_p(score), None if m is None else _p(m), score.shape[1], score.shape[0],
                                 edge, _p(kp), cap)
    return kp[:min(n, cap)].copy()


def harris_at(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orb_harris_at(_p(img), img.shape[1], x, y)


def umax():
    u = np.zeros(17, np.int32)
    lib().orb_umax(_p(u))
    return u


def ic_angle_at(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    u = umax()
    return lib().orb_ic_angle_at(_p(img), img.shape[1], x, y, _p(u))


def fast_atan2(y, x):
    return lib().orb_fast_atan2(float(y), float(x))


def detect(img, mask, fast_threshold, cap=60000):
    img = np.ascontiguousarray(img, np.uint8)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    kp = np.zeros(cap, KP_DTYPE)
    n = lib().orb_detect(_p(img), None if m is None else _p(m), img.shape[1], img.shape[0], img.shape[1],
                         img.shape[1], fast_threshold, _p(kp), cap)
    return kp[:n].copy()


def grid_state(max_keypoints=1000, grid=3, max_iters=5):
    st = GridState()
    lib().orb_grid_state_init(C.byref(st), max_keypoints, grid, max_iters)
    return st


def grid_detect(st, img, mask, cap=60000):
    img = np.ascontiguousarray(img, np.uint8)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    kp = np.zeros(cap, KP_DTYPE)
    n = lib().orb_grid_detect(C.byref(st), _p(img), None if m is None else _p(m), img.shape[1],
                              img.shape[0], _p(kp), cap)
    return kp[:n].copy()


def gaussian_blur7(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().orb_gaussian_blur7(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(out))
    return out


def gauss_kernel():
    k = np.zeros(7, np.int32)
    lib().orb_gauss7_kernel_fixed(_p(k))
    return k


def compute(img, kp):
    """Returns (kept keypoints regrouped by level, descriptors [n,32])."""
    img = np.ascontiguousarray(img, np.uint8)
    kp = np.ascontiguousarray(kp.copy())
    desc = np.zeros((max(len(kp), 1), 32), np.uint8)
    n = lib().orb_compute(_p(img), img.shape[1], img.shape[0], _p(kp), len(kp), _p(desc))
    return kp[:n].copy(), desc[:n].copy()


def node_features(st, gray, mask, depth, max_keypoints=1000, cap=60000):
    gray = np.ascontiguousarray(gray, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    depth = np.ascontiguousarray(depth, np.float32)
    kp = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().orb_node_features(C.byref(st), _p(gray), _p(mask), _p(depth), gray.shape[1], gray.shape[0],
                                max_keypoints, _p(kp), cap, _p(desc))
    return kp[:n].copy(), desc[:n].copy()


def set_use_feature_min_depth(on):
    """Parameter "use_feature_min_depth" for node_features (node.cpp:82)."""
    lib().orb_set_use_feature_min_depth(1 if on else 0)


def pattern():
    return np.ctypeslib.as_array(lib().orb_pattern(), shape=(1024,)).copy()


_ref_adj = None


def ref_adjuster_lib():
    """The reference's own detector grid + threshold adaptation (src/feature_adjuster.cpp, src/features.cpp:35-60)
    compiled from /root/reference around the oracle's cv::ORB::detect restatement (or None)."""
    global _ref_adj
    if _ref_adj is None:
        import os
        from . import pyoracle
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_adjuster.so")
        if not os.path.exists(p):
            pyoracle.build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_grid_detector_create.restype = C.c_void_p
        R.ref_grid_detector_create.argtypes = [C.c_int, C.c_int, C.c_int]
        R.ref_grid_detector_destroy.restype = None
        R.ref_grid_detector_destroy.argtypes = [C.c_void_p]
        R.ref_grid_detector_detect.restype = C.c_int
        R.ref_grid_detector_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _ref_adj = R
    return _ref_adj


def ref_grid_detect(handle, img, mask, cap=60000):
    img = np.ascontiguousarray(img, np.uint8)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    kp = np.zeros(cap, KP_DTYPE)
    n = ref_adjuster_lib().ref_grid_detector_detect(handle, _p(img), None if m is None else _p(m), img.shape[1],
                                                    img.shape[0], _p(kp), cap)
    return kp[:n].copy()
