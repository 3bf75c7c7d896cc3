"""ctypes binding of the CPU oracle (oracle/liboracle.so) and of the executable
reference pin (oracle/_ref/libref_bforb.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under rgbdslam_v2_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORC_MAX_MATCHES = 320


class OrcParams(C.Structure):
    _fields_ = [
        ("max_matches", C.c_int32),
        ("min_matches", C.c_int32),
        ("ransac_iterations", C.c_int32),
        ("max_dist_for_inliers", C.c_float),
        ("depth_cov", C.c_double),
        ("seed", C.c_uint32),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("id1", C.c_int32),
        ("id2", C.c_int32),
        ("n_all", C.c_int32),
        ("n_inl", C.c_int32),
        ("rmse", C.c_float),
        ("T", C.c_float * 16),
        ("info_scale", C.c_double),
        ("valid_iterations", C.c_int32),
        ("real_iterations", C.c_int32),
        ("all_q", C.c_int32 * ORC_MAX_MATCHES),
        ("all_t", C.c_int32 * ORC_MAX_MATCHES),
        ("all_hd", C.c_int32 * ORC_MAX_MATCHES),
        ("inl_idx", C.c_int32 * ORC_MAX_MATCHES),
    ]


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "rgbd_oracle.c")
    hdr = os.path.join(_HERE, "rgbd_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(f) > os.path.getmtime(so) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if force or not all(os.path.exists(os.path.join(_HERE, "_ref", f))
                        for f in ("libref_bforb.so", "libref_node.so", "libref_adjuster.so", "libref_ransac.so",
                                  "libref_frame.so", "libref_siftmatch.so", "libref_graph.so", "libref_siftgpu.so")):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        vp = C.c_void_p
        L.orc_hamming_nn.restype = C.c_int
        L.orc_hamming_nn.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_int)]
        L.orc_hamming_nn_batch.restype = None
        L.orc_hamming_nn_batch.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, vp]
        L.orc_feature_matching_orb.restype = C.c_int
        L.orc_feature_matching_orb.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_int, vp, vp, vp]
        L.orc_rand31.restype = C.c_uint32
        L.orc_rand31.argtypes = [C.c_uint32] * 4
        L.orc_pair_uid.restype = C.c_uint32
        L.orc_pair_uid.argtypes = [C.c_int32, C.c_int32]
        L.orc_sample4.restype = C.c_int
        L.orc_sample4.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.orc_fit_transform.restype = None
        L.orc_fit_transform.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
        L.orc_svd3.restype = None
        L.orc_svd3.argtypes = [vp, vp, vp, vp]
        L.orc_error_function2.restype = C.c_double
        L.orc_error_function2.argtypes = [vp, vp, vp, C.c_double]
        L.orc_raster_cov.restype = None
        L.orc_raster_cov.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_compute_inliers_and_error.restype = C.c_int
        L.orc_compute_inliers_and_error.argtypes = [vp, vp, vp, vp, C.c_int, vp, C.c_double,
                                                    C.c_double, vp, C.POINTER(C.c_double)]
        L.orc_ransac.restype = C.c_int
        L.orc_ransac.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(OrcParams), C.c_uint32, vp,
                                 C.POINTER(C.c_float), vp, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_match_node_pair.restype = None
        L.orc_match_node_pair.argtypes = [vp, vp, C.c_uint32, C.c_int32, vp, vp, C.c_uint32,
                                          C.c_int32, C.POINTER(OrcParams), C.POINTER(OrcResult)]
        L.orc_match_pairs_mt.restype = None
        L.orc_match_pairs_mt.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(OrcParams),
                                         vp, C.c_int]
        L.orc_project_to_3d_cloud.restype = C.c_int
        L.orc_project_to_3d_cloud.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp]
        L.orc_project_to_3d_sift.restype = C.c_int
        L.orc_project_to_3d_sift.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_double,
                                             C.c_double, C.c_double, C.c_double, C.c_int, vp, vp]
        L.orc_gather_rows_f32.restype = None
        L.orc_gather_rows_f32.argtypes = [vp, vp, C.c_int, C.c_int, vp]
        L.orc_root_sift.restype = None
        L.orc_root_sift.argtypes = [vp, C.c_int, C.c_int]
        L.orc_depth_to_mono8_f32.restype = None
        L.orc_depth_to_mono8_f32.argtypes = [vp, C.c_size_t, vp]
        L.orc_depth_u16_to_mono8_f32.restype = None
        L.orc_depth_u16_to_mono8_f32.argtypes = [vp, C.c_size_t, vp, vp]
        L.orc_create_point_cloud.restype = None
        L.orc_create_point_cloud.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_double,
                                             C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, vp]
        L.orc_observation_likelihood.restype = None
        L.orc_observation_likelihood.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_double, C.c_double, C.c_double,
                                                 C.c_double, C.c_int, C.c_int, C.c_double, vp]
        L.orc_emm_erf_boundaries.restype = None
        L.orc_emm_erf_boundaries.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_observation_criterion_met.restype = C.c_int
        L.orc_observation_criterion_met.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_double, C.POINTER(C.c_double)]
        L.orc_project_to_3d.restype = C.c_int
        L.orc_project_to_3d.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_double, C.c_double, C.c_double, C.c_int, vp, vp]
        L.orc_num_cores.restype = C.c_int
        L.orc_sift_match.restype = C.c_int
        L.orc_sift_match.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
        L.orc_match_sift_node_pair.restype = None
        L.orc_match_sift_node_pair.argtypes = [vp, vp, C.c_int, C.c_int32, vp, vp, C.c_int, C.c_int32,
                                               C.POINTER(OrcParams), C.POINTER(OrcResult), vp]
        _lib = L
    return _lib


def ref_lib():
    """The reference's own bruteForceSearchORB, compiled from /root/reference (or None)."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libref_bforb.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_bruteForceSearchORB.restype = C.c_int
        R.ref_bruteForceSearchORB.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_int)]
        _ref = R
    return _ref


_ref_node = None


def ref_node_lib():
    """The reference's own sample_matches_prefer_by_distance and keepStrongestMatches
    (node.cpp:1023-1047, 516-531), compiled from /root/reference (or None)."""
    global _ref_node
    if _ref_node is None:
        p = os.path.join(_HERE, "_ref", "libref_node.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_sample_ids.restype = C.c_int
        R.ref_sample_ids.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        R.ref_keep_strongest.restype = C.c_int
        R.ref_keep_strongest.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        R.ref_set_sigma_depth.restype = None
        R.ref_set_sigma_depth.argtypes = [C.c_double]
        R.ref_depth_covariance.restype = C.c_double
        R.ref_depth_covariance.argtypes = [C.c_double]
        R.ref_back_project.restype = None
        R.ref_back_project.argtypes = [C.c_float] * 7 + [C.c_void_p]
        _ref_node = R
    return _ref_node


def ref_sample_ids(n_matches, stream, sample_size=4):
    """Reference sampling with rand() replaced by `stream` (ints): returns (ids, draws consumed)."""
    stream = np.ascontiguousarray(stream, np.int32)
    ids = np.zeros(max(sample_size, 1), np.int32)
    used = C.c_int(0)
    n = ref_node_lib().ref_sample_ids(int(n_matches), int(sample_size), _p(stream), _p(ids), C.byref(used))
    return ids[:n].copy(), used.value


def ref_keep_strongest(n, dist):
    dist = np.ascontiguousarray(dist, np.float32)
    kept = np.zeros(max(len(dist), 1), np.int32)
    k = ref_node_lib().ref_keep_strongest(int(n), _p(dist), len(dist), _p(kept))
    return kept[:k].copy()


_ref_ransac = None


def ref_ransac_lib():
    """The reference's own getRelativeTransformationTo / computeInliersAndError / errorFunction2 /
    getTransformFromMatches, compiled from /root/reference with Eigen / PCL stand-ins (or None)."""
    global _ref_ransac
    if _ref_ransac is None:
        p = os.path.join(_HERE, "_ref", "libref_ransac.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_get_relative_transformation.restype = C.c_int
        R.ref_get_relative_transformation.argtypes = (
            [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
             C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_void_p,
             C.POINTER(C.c_int), C.POINTER(C.c_int)])
        R.ref_match_node_pair.restype = C.c_int
        R.ref_match_node_pair.argtypes = (
            [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
             C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
             C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int),
             C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)])
        _ref_ransac = R
    return _ref_ransac


def ref_match_node_pair(qdesc, qxyz1, qid, tdesc, txyz1, tid, params):
    """The reference's own Node::matchNodePair (featureMatching -> keepStrongestMatches -> RANSAC -> edge),
    compiled from /root/reference; the distance jitter grows with the query index (D2), draws are D1's."""
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    tdesc = np.ascontiguousarray(tdesc, np.uint8)
    qxyz1 = np.ascontiguousarray(qxyz1, np.float32)
    txyz1 = np.ascontiguousarray(txyz1, np.float32)
    nq, nt = len(qdesc), len(tdesc)
    cap = max(nq, 1)
    aq, at, iq, it = (np.zeros(cap, np.int32) for _ in range(4))
    ad = np.zeros(cap, np.float32)
    T = np.zeros(16, np.float32)
    n_all, n_inl, id1, id2, iters = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    rmse, info = C.c_float(0), C.c_double(0)
    accepted = ref_ransac_lib().ref_match_node_pair(
        _p(qdesc), _p(qxyz1), nq, int(qid), _p(tdesc), _p(txyz1), nt, int(tid), params.max_matches, params.min_matches,
        params.ransac_iterations, float(params.max_dist_for_inliers), params.depth_cov, params.seed,
        lib().orc_pair_uid(int(qid), int(tid)), _p(aq), _p(at), _p(ad), C.byref(n_all), _p(iq), _p(it), C.byref(n_inl),
        _p(T), C.byref(rmse), C.byref(id1), C.byref(id2), C.byref(info), C.byref(iters))
    na, ni = n_all.value, n_inl.value
    return dict(id1=id1.value, id2=id2.value, all_q=aq[:na].copy(), all_t=at[:na].copy(), all_dist=ad[:na].copy(),
                inl_q=iq[:ni].copy(), inl_t=it[:ni].copy(), T=T.reshape(4, 4).T.copy(), rmse=np.float32(rmse.value),
                info_scale=info.value, real_iterations=iters.value, accepted=accepted)


def ref_get_relative_transformation(qxyz1, txyz1, mq, mt, mdist, params, uid):
    """Runs the reference's RANSAC on a match list (any order; it sorts by distance itself).  Returns a dict
    shaped like result_to_dict's RANSAC part."""
    qxyz1 = np.ascontiguousarray(qxyz1, np.float32)
    txyz1 = np.ascontiguousarray(txyz1, np.float32)
    mq = np.ascontiguousarray(mq, np.int32)
    mt = np.ascontiguousarray(mt, np.int32)
    mdist = np.ascontiguousarray(mdist, np.float32)
    n = len(mq)
    T = np.zeros(16, np.float32)
    rmse, n_inl, iters = C.c_float(0), C.c_int(0), C.c_int(0)
    iq = np.zeros(max(n, 1), np.int32)
    it = np.zeros(max(n, 1), np.int32)
    found = ref_ransac_lib().ref_get_relative_transformation(
        _p(qxyz1), len(qxyz1), _p(txyz1), len(txyz1), _p(mq), _p(mt), _p(mdist), n, params.min_matches,
        params.ransac_iterations, float(params.max_dist_for_inliers), params.depth_cov, params.seed, int(uid),
        _p(T), C.byref(rmse), _p(iq), _p(it), C.byref(n_inl), C.byref(iters))
    return dict(found=bool(found), T=T.reshape(4, 4).T.copy(), rmse=np.float32(rmse.value),
                inl_q=iq[:n_inl.value].copy(), inl_t=it[:n_inl.value].copy(), real_iterations=iters.value)


_ref_frame = None


def ref_frame_lib():
    """The reference's own removeDepthless / projectTo3D / projectTo3DSiftGPU / squareroot_descriptor_space /
    createXYZRGBPointCloud / observationLikelihood, compiled from /root/reference with stand-ins (or None)."""
    global _ref_frame
    if _ref_frame is None:
        p = os.path.join(_HERE, "_ref", "libref_frame.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        d, i, vp = C.c_double, C.c_int, C.c_void_p
        R.ref_remove_depthless.restype = i
        R.ref_remove_depthless.argtypes = [vp, i, vp, i, i, vp]
        R.ref_project_to_3d.restype = i
        R.ref_project_to_3d.argtypes = [vp, i, vp, i, i, d, d, d, d, d, i, vp, vp]
        R.ref_project_to_3d_cloud.restype = i
        R.ref_project_to_3d_cloud.argtypes = [vp, i, vp, i, i, d, i, vp, vp]
        R.ref_project_to_3d_sift.restype = i
        R.ref_project_to_3d_sift.argtypes = [vp, i, vp, vp, i, i, d, d, d, d, d, i, vp, vp, vp, vp]
        if hasattr(R, "ref_project_to_3d_sift_min_depth"):
            R.ref_project_to_3d_sift_min_depth.restype = i
            R.ref_project_to_3d_sift_min_depth.argtypes = [vp, vp, i, vp, vp, i, i, d, d, d, d, d, i, vp, vp]
        R.ref_root_sift.restype = None
        R.ref_root_sift.argtypes = [vp, i, i]
        R.ref_create_point_cloud.restype = None
        R.ref_create_point_cloud.argtypes = [vp, i, i, vp, i, i, d, d, d, d, d, d, i, vp]
        R.ref_observation_likelihood.restype = None
        R.ref_observation_likelihood.argtypes = [vp, vp, i, i, vp, d, d, d, d, i, i, d, vp]
        R.ref_observation_criterion_met.restype = i
        R.ref_observation_criterion_met.argtypes = [C.c_uint, C.c_uint, C.c_uint, d, C.POINTER(d)]
        _ref_frame = R
    return _ref_frame


_ref_sift = None


def ref_sift_lib():
    """The SiftGPU matcher shipped in the reference tree (CUDA kernels + SiftMatchCU host code) and
    SiftGPUWrapper::match, compiled from /root/reference on a CUDA-on-CPU emulation (or None)."""
    global _ref_sift
    if _ref_sift is None:
        p = os.path.join(_HERE, "_ref", "libref_siftmatch.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_sift_match.restype = C.c_int
        R.ref_sift_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _ref_sift = R
    return _ref_sift


def ref_sift_match(d1, d2):
    d1 = np.ascontiguousarray(d1, np.float32)
    d2 = np.ascontiguousarray(d2, np.float32)
    n1 = d1.shape[0]
    mq = np.empty(max(n1, 1), np.int32)
    mt = np.empty(max(n1, 1), np.int32)
    md = np.empty(max(n1, 1), np.float32)
    n = ref_sift_lib().ref_sift_match(_p(d1), n1, _p(d2), d2.shape[0], _p(mq), _p(mt), _p(md))
    return mq[:n].copy(), mt[:n].copy(), md[:n].copy()


_ref_siftgpu = None


def ref_siftgpu_lib():
    """SiftGPU's CUDA extraction pipeline as the reference vendors it (ProgramCU.cu kernels + launchers, PyramidCU.cpp,
    SiftPyramid::RunSIFT), compiled from /root/reference on the fiber-based CUDA-on-CPU emulation of
    oracle/ref_stubs/siftgpu_emu_prelude.h (or None when the pin is not built)."""
    global _ref_siftgpu
    if _ref_siftgpu is None:
        p = os.path.join(_HERE, "_ref", "libref_siftgpu.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        vp, i = C.c_void_p, C.c_int
        R.ref_siftgpu_run.restype = i
        R.ref_siftgpu_run.argtypes = [vp, i, i, i, vp, vp, i]
        R.ref_siftgpu_geometry.restype = i
        R.ref_siftgpu_geometry.argtypes = [C.POINTER(i)] * 4
        R.ref_siftgpu_level.restype = i
        R.ref_siftgpu_level.argtypes = [i, i, i, vp, C.POINTER(i), C.POINTER(i)]
        R.ref_siftgpu_level_counts.restype = i
        R.ref_siftgpu_level_counts.argtypes = [vp, i]
        R.ref_siftgpu_filter_kernel.restype = i
        R.ref_siftgpu_filter_kernel.argtypes = [C.c_float, vp]
        _ref_siftgpu = R
    return _ref_siftgpu


def ref_sift_detect(gray, max_features=1000):
    """SiftGPUWrapper::detect's SiftGPU call on a mono8 image through the compiled reference: (keys [n, 4] = x, y, scale,
    orientation; descriptors [n, 128]; features per (octave, dog level))."""
    R = ref_siftgpu_lib()
    gray = np.ascontiguousarray(gray, np.uint8)
    cap = 1 << 16
    while True:
        keys = np.zeros((cap, 4), np.float32)
        desc = np.zeros((cap, 128), np.float32)
        n = R.ref_siftgpu_run(_p(gray), gray.shape[1], gray.shape[0], int(max_features), _p(keys), _p(desc), cap)
        if n >= 0:
            break
        cap = -n - 1 + 16
    cnt = np.zeros(128, np.int32)
    m = R.ref_siftgpu_level_counts(_p(cnt), 128)
    return keys[:n].copy(), desc[:n].copy(), cnt[:m].copy()


def ref_sift_describe(gray, keys):
    """The reference's SiftGPU pipeline with a caller-provided keypoint list (SiftGPUWrapper::detect's second mode):
    keys [n, 4] = (x, y, scale, orientation in radians) -> descriptors [n, 128]."""
    L = ref_siftgpu_lib()
    gray = np.ascontiguousarray(gray, np.uint8)
    keys = np.ascontiguousarray(keys, np.float32)
    n = keys.shape[0]
    desc = np.zeros((max(n, 1), 128), np.float32)
    L.ref_siftgpu_describe.restype = C.c_int
    L.ref_siftgpu_describe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rc = L.ref_siftgpu_describe(_p(gray), gray.shape[1], gray.shape[0], _p(keys), n, _p(desc))
    if rc != n:
        raise RuntimeError("ref_siftgpu_describe failed")
    return desc[:n]


def ref_sift_geometry():
    R = ref_siftgpu_lib()
    a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    R.ref_siftgpu_geometry(C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    return dict(octave_min=a.value, octave_num=b.value, levels=c.value, dog_levels=d.value)


def ref_sift_level(octave, level, data=0):
    """One plane of the latest ref_sift_detect pyramid.  data: 0 Gaussian, 1 DoG, 2 keypoint map (sign, dx, dy, ds),
    3 gradient (magnitude, angle).  Returns [h, w] or [h, w, channels]."""
    R = ref_siftgpu_lib()
    ch = {0: 1, 1: 1, 2: 4, 3: 2}[data]
    buf = np.zeros(4 * 4096 * 4096 // 4, np.float32) if False else np.zeros(1 << 25, np.float32)
    w, h = C.c_int(), C.c_int()
    n = R.ref_siftgpu_level(octave, level, data, _p(buf), C.byref(w), C.byref(h))
    if n <= 0:
        return None
    out = buf[:n].reshape(h.value, w.value, ch) if ch > 1 else buf[:n].reshape(h.value, w.value)
    return out.copy()


def ref_sift_candidates(octave, dog_level):
    """The keypoint candidates of one (octave, dog level) of the latest ref_sift_detect run, as InitHist / the list
    generation enumerate them (ProgramCU.cu:665-688, PyramidCU.cpp:738-795): the non-zero entries of the keypoint map in
    raster order, rows 1 .. h-2, columns 1 .. w-2.  Rows of (x, y, sign, dx, dy, ds)."""
    key = ref_sift_level(octave, dog_level + 2, 2)
    h, w = key.shape[:2]
    inner = np.zeros((h, w), bool)
    inner[1:h - 1, 1:w - 1] = True
    ys, xs = np.nonzero((key[:, :, 0] != 0) & inner)
    return np.concatenate([xs[:, None].astype(np.float32), ys[:, None].astype(np.float32), key[ys, xs]], axis=1)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw):
    p = OrcParams(max_matches=300, min_matches=20, ransac_iterations=200,
                  max_dist_for_inliers=3.0, depth_cov=1e-4, seed=20260923)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def hamming_nn_batch(qdesc, tdesc):
    qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
    tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
    nq, nt = qdesc.shape[0], tdesc.shape[0]
    hd = np.empty(nq, np.int32)
    idx = np.empty(nq, np.int32)
    lib().orc_hamming_nn_batch(_p(qdesc), nq, _p(tdesc), nt, _p(hd), _p(idx))
    return hd, idx


def ref_hamming_nn(q_row, tdesc):
    """One query through the reference's own compiled function."""
    R = ref_lib()
    q = np.ascontiguousarray(q_row, dtype=np.uint8)
    t = np.ascontiguousarray(tdesc, dtype=np.uint8)
    idx = C.c_int(-2)
    d = R.ref_bruteForceSearchORB(_p(q), _p(t), t.shape[0], C.byref(idx))
    return d, idx.value


def feature_matching_orb(qdesc, tdesc, max_matches=300):
    qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
    tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
    cap = max(max_matches, 1)
    mq = np.empty(cap, np.int32)
    mt = np.empty(cap, np.int32)
    mhd = np.empty(cap, np.int32)
    n = lib().orc_feature_matching_orb(_p(qdesc), qdesc.shape[0], _p(tdesc), tdesc.shape[0],
                                       max_matches, _p(mq), _p(mt), _p(mhd))
    return mq[:n].copy(), mt[:n].copy(), mhd[:n].copy()


def svd3(Cm):
    Cm = np.ascontiguousarray(Cm, dtype=np.float32)
    U = np.empty((3, 3), np.float32)
    S = np.empty(3, np.float32)
    V = np.empty((3, 3), np.float32)
    lib().orc_svd3(_p(Cm), _p(U), _p(S), _p(V))
    return U, S, V


def fit_transform(qxyz1, txyz1, mq, mt, sel):
    qxyz1 = np.ascontiguousarray(qxyz1, np.float32)
    txyz1 = np.ascontiguousarray(txyz1, np.float32)
    mq = np.ascontiguousarray(mq, np.int32)
    mt = np.ascontiguousarray(mt, np.int32)
    sel = np.ascontiguousarray(sel, np.int32)
    T = np.empty(16, np.float32)
    lib().orc_fit_transform(_p(qxyz1), _p(txyz1), _p(mq), _p(mt), _p(sel), len(sel), _p(T))
    return T.reshape(4, 4).T.copy()  # column-major -> numpy [row, col]


def error_function2(x1, x2, T, depth_cov):
    x1 = np.ascontiguousarray(x1, np.float32)
    x2 = np.ascontiguousarray(x2, np.float32)
    Tc = np.ascontiguousarray(np.asarray(T, np.float64).T)  # to column-major
    return lib().orc_error_function2(_p(x1), _p(x2), _p(Tc), float(depth_cov))


def raster_cov():
    a, b = C.c_double(), C.c_double()
    lib().orc_raster_cov(C.byref(a), C.byref(b))
    return a.value, b.value


def match_node_pair(qdesc, qxyz1, qid, tdesc, txyz1, tid, params=None):
    params = params or default_params()
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    tdesc = np.ascontiguousarray(tdesc, np.uint8)
    qxyz1 = np.ascontiguousarray(qxyz1, np.float32)
    txyz1 = np.ascontiguousarray(txyz1, np.float32)
    out = OrcResult()
    lib().orc_match_node_pair(_p(qdesc), _p(qxyz1), qdesc.shape[0], qid, _p(tdesc), _p(txyz1),
                              tdesc.shape[0], tid, C.byref(params), C.byref(out))
    return result_to_dict(out)


def result_to_dict(r):
    n_all, n_inl = r.n_all, r.n_inl
    return dict(
        id1=r.id1, id2=r.id2, n_all=n_all, n_inl=n_inl, rmse=np.float32(r.rmse),
        T=np.array(r.T, np.float32).reshape(4, 4).T.copy(),
        info_scale=r.info_scale, valid_iterations=r.valid_iterations,
        real_iterations=r.real_iterations,
        all_q=np.array(r.all_q[:n_all], np.int32), all_t=np.array(r.all_t[:n_all], np.int32),
        all_hd=np.array(r.all_hd[:n_all], np.int32),
        inl_idx=np.array(r.inl_idx[:n_inl], np.int32))


def usable_cpus():
    """Hardware threads this process may use, capped by the container's cgroup CPU quota (oversubscribing the quota
    makes the OpenMP run slower, not faster)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def match_pairs_mt(descs, xyzs, node_ids, pair_q, pair_t, params=None, n_threads=0):
    """Pair-parallel oracle run (the CPU baseline).  descs/xyzs: lists of per-node arrays;
    pair_q/pair_t index into those lists.  n_threads = 0: the CPUs this process may use."""
    params = params or default_params()
    n_threads = n_threads or usable_cpus()
    n_nodes = len(descs)
    descs = [np.ascontiguousarray(d, np.uint8) for d in descs]
    xyzs = [np.ascontiguousarray(x, np.float32) for x in xyzs]
    dptr = (C.c_void_p * n_nodes)(*[d.ctypes.data for d in descs])
    xptr = (C.c_void_p * n_nodes)(*[x.ctypes.data for x in xyzs])
    counts = np.array([d.shape[0] for d in descs], np.uint32)
    ids = np.ascontiguousarray(node_ids, np.int32)
    pq = np.ascontiguousarray(pair_q, np.int32)
    pt = np.ascontiguousarray(pair_t, np.int32)
    out = (OrcResult * len(pq))()
    lib().orc_match_pairs_mt(dptr, xptr, _p(counts), _p(ids), _p(pq), _p(pt), len(pq),
                             C.byref(params), out, n_threads)
    return out


def project_to_3d(kp_xy, depth, fx, fy, cx, cy, depth_scaling=1.0, max_keypoints=1000):
    kp_xy = np.ascontiguousarray(kp_xy, np.float32)
    depth = np.ascontiguousarray(depth, np.float32)
    n = kp_xy.shape[0]
    kept = np.empty(max(n, 1), np.int32)
    xyz1 = np.empty((max(n, 1), 4), np.float32)
    k = lib().orc_project_to_3d(_p(kp_xy), n, _p(depth), depth.shape[0], depth.shape[1],
                                fx, fy, cx, cy, depth_scaling, max_keypoints, _p(kept), _p(xyz1))
    return kept[:k].copy(), xyz1[:k].copy()


def min_depth_in_neighborhood(depth, x, y, diameter):
    """getMinDepthInNeighborhood (misc.cpp:774-793)."""
    depth = np.ascontiguousarray(depth, np.float32)
    L = lib()
    L.orc_min_depth_in_neighborhood.restype = C.c_float
    L.orc_min_depth_in_neighborhood.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    return float(L.orc_min_depth_in_neighborhood(_p(depth), depth.shape[0], depth.shape[1], x, y, diameter))


def remove_depthless_min_depth(kp_xy, kp_size, depth):
    """removeDepthless with use_feature_min_depth (node.cpp:66-97): kept input positions."""
    kp_xy = np.ascontiguousarray(kp_xy, np.float32)
    kp_size = np.ascontiguousarray(kp_size, np.float32)
    depth = np.ascontiguousarray(depth, np.float32)
    n = kp_xy.shape[0]
    kept = np.empty(max(n, 1), np.int32)
    k = lib().orc_remove_depthless_min_depth(_p(kp_xy), _p(kp_size), n, _p(depth), depth.shape[0], depth.shape[1], _p(kept))
    return kept[:k].copy()


def project_to_3d_min_depth(kp_xy, kp_size, depth, fx, fy, cx, cy, depth_scaling=1.0, max_keypoints=1000):
    """projectTo3D with use_feature_min_depth (node.cpp:900-965, :940)."""
    kp_xy = np.ascontiguousarray(kp_xy, np.float32)
    kp_size = np.ascontiguousarray(kp_size, np.float32)
    depth = np.ascontiguousarray(depth, np.float32)
    n = kp_xy.shape[0]
    kept = np.empty(max(n, 1), np.int32)
    xyz1 = np.empty((max(n, 1), 4), np.float32)
    L = lib()
    L.orc_project_to_3d_min_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                              C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    k = L.orc_project_to_3d_min_depth(_p(kp_xy), _p(kp_size), n, _p(depth), depth.shape[0], depth.shape[1], fx, fy, cx, cy,
                                      depth_scaling, max_keypoints, _p(kept), _p(xyz1))
    return kept[:k].copy(), xyz1[:k].copy()


def project_to_3d_cloud(kp_xy, cloud, maximum_depth, max_keypoints=1000):
    """Node::projectTo3D, point-cloud overload (node.cpp:855-898).  cloud: [rows, cols, 4] float32."""
    kp_xy = np.ascontiguousarray(kp_xy, np.float32)
    cloud = np.ascontiguousarray(cloud, np.float32)
    n = kp_xy.shape[0]
    kept = np.empty(max(n, 1), np.int32)
    xyz1 = np.empty((max(n, 1), 4), np.float32)
    k = lib().orc_project_to_3d_cloud(_p(kp_xy), n, _p(cloud), cloud.shape[0], cloud.shape[1], maximum_depth,
                                      max_keypoints, _p(kept), _p(xyz1))
    return kept[:k].copy(), xyz1[:k].copy()


def sift_node_features(kp_xy, desc, depth, fx, fy, cx, cy, depth_scaling=1.0, max_keypoints=1000,
                       use_root_sift=True, kp_size=None):
    """projectTo3DSiftGPU + squareroot_descriptor_space: (kept_idx, xyz1, siftgpu_descriptors,
    feature_descriptors).  kp_size (cv::KeyPoint::size per keypoint): the use_feature_min_depth variant (node.cpp:730)."""
    kp_xy = np.ascontiguousarray(kp_xy, np.float32)
    desc = np.ascontiguousarray(desc, np.float32)
    depth = np.ascontiguousarray(depth, np.float32)
    n = kp_xy.shape[0]
    kept = np.empty(max(n, 1), np.int32)
    xyz1 = np.empty((max(n, 1), 4), np.float32)
    if kp_size is not None:
        kp_size = np.ascontiguousarray(kp_size, np.float32)
        f = lib().orc_project_to_3d_sift_min_depth
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
        k = f(_p(kp_xy), _p(kp_size), n, _p(depth), depth.shape[0], depth.shape[1], fx, fy, cx, cy, depth_scaling,
              max_keypoints, _p(kept), _p(xyz1))
    else:
        k = lib().orc_project_to_3d_sift(_p(kp_xy), n, _p(depth), depth.shape[0], depth.shape[1],
                                         fx, fy, cx, cy, depth_scaling, max_keypoints, _p(kept), _p(xyz1))
    raw = np.empty((max(k, 1), desc.shape[1]), np.float32)
    lib().orc_gather_rows_f32(_p(desc), _p(kept), k, desc.shape[1], _p(raw))
    raw = raw[:k].copy()
    feat = raw.copy()
    if use_root_sift and k > 0:
        lib().orc_root_sift(_p(feat), k, desc.shape[1])
    return kept[:k].copy(), xyz1[:k].copy(), raw, feat


def depth_to_mono8(depth):
    depth = np.ascontiguousarray(depth)
    mono8 = np.empty(depth.shape, np.uint8)
    if depth.dtype == np.uint16:
        dm = np.empty(depth.shape, np.float32)
        lib().orc_depth_u16_to_mono8_f32(_p(depth), depth.size, _p(mono8), _p(dm))
        return mono8, dm
    depth = np.ascontiguousarray(depth, np.float32)
    lib().orc_depth_to_mono8_f32(_p(depth), depth.size, _p(mono8))
    return mono8


def create_point_cloud(depth, fx, fy, cx, cy, rgb=None, encoding_bgr=False, depth_scaling=1.0, min_depth=0.1,
                       cloud_skip=2):
    depth = np.ascontiguousarray(depth, np.float32)
    rows, cols = depth.shape
    ch = 1
    if rgb is not None:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        ch = 1 if rgb.ndim == 2 else rgb.shape[2]
    out = np.empty((-(-rows // cloud_skip), -(-cols // cloud_skip), 4), np.float32)
    lib().orc_create_point_cloud(_p(depth), rows, cols, _p(rgb) if rgb is not None else None, ch,
                                 int(bool(encoding_bgr)), fx, fy, cx, cy, depth_scaling, min_depth,
                                 int(cloud_skip), _p(out))
    return out


def observation_likelihood(new_cloud, old_cloud, T, fx, fy, cx, cy, cloud_skip=2, skip_step=8, depth_cov=1e-4):
    """T: 4x4 row-major new -> old.  Returns uint32[4] = (inliers, outliers, occluded, all)."""
    new_cloud = np.ascontiguousarray(new_cloud, np.float32)
    old_cloud = np.ascontiguousarray(old_cloud, np.float32)
    T = np.ascontiguousarray(T, np.float32)
    out = np.zeros(4, np.uint32)
    lib().orc_observation_likelihood(_p(new_cloud), _p(old_cloud), old_cloud.shape[0], old_cloud.shape[1], _p(T),
                                     fx, fy, cx, cy, int(cloud_skip), int(skip_step), depth_cov, _p(out))
    return out


def emm_erf_boundaries():
    lo, hi = C.c_double(0), C.c_double(0)
    lib().orc_emm_erf_boundaries(C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def observation_criterion_met(inliers, outliers, all_points, obs_thresh):
    q = C.c_double(0)
    return bool(lib().orc_observation_criterion_met(int(inliers), int(outliers), int(all_points), obs_thresh,
                                                    C.byref(q))), q.value


def num_cores():
    return lib().orc_num_cores()


def sift_match(d1, d2):
    """SiftGPUWrapper::match restatement: (queryIdx, trainIdx, L2 distance) in ascending query order."""
    d1 = np.ascontiguousarray(d1, np.float32)
    d2 = np.ascontiguousarray(d2, np.float32)
    n1 = d1.shape[0]
    mq = np.empty(max(n1, 1), np.int32)
    mt = np.empty(max(n1, 1), np.int32)
    md = np.empty(max(n1, 1), np.float32)
    n = lib().orc_sift_match(_p(d1), n1, _p(d2), d2.shape[0], _p(mq), _p(mt), _p(md))
    return mq[:n].copy(), mt[:n].copy(), md[:n].copy()


def match_sift_node_pair(qdesc, qxyz1, qid, tdesc, txyz1, tid, params=None):
    params = params or default_params()
    qdesc = np.ascontiguousarray(qdesc, np.float32)
    tdesc = np.ascontiguousarray(tdesc, np.float32)
    qxyz1 = np.ascontiguousarray(qxyz1, np.float32)
    txyz1 = np.ascontiguousarray(txyz1, np.float32)
    out = OrcResult()
    dist = np.zeros(ORC_MAX_MATCHES, np.float32)
    lib().orc_match_sift_node_pair(_p(qdesc), _p(qxyz1), qdesc.shape[0], qid, _p(tdesc), _p(txyz1),
                                   tdesc.shape[0], tid, C.byref(params), C.byref(out), _p(dist))
    r = result_to_dict(out)
    r["all_dist"] = dist[: r["n_all"]].copy()
    return r


def match_node_pair_g2o(qdesc, qxyz1, qkp, qid, tdesc, txyz1, tkp, tid, g2o_iterations, params=None):
    """matchNodePair with g2o_transformation_refinement = g2o_iterations (node.cpp:1222-1268)."""
    params = params or default_params()
    arrs = [np.ascontiguousarray(a, dt) for a, dt in ((qdesc, np.uint8), (qxyz1, np.float32), (qkp, np.float32),
                                                      (tdesc, np.uint8), (txyz1, np.float32), (tkp, np.float32))]
    out = OrcResult()
    L = lib()
    vp = C.c_void_p
    L.orc_match_node_pair_g2o.restype = None
    L.orc_match_node_pair_g2o.argtypes = [vp, vp, vp, C.c_uint32, C.c_int32, vp, vp, vp, C.c_uint32, C.c_int32,
                                          C.POINTER(OrcParams), C.c_int, C.POINTER(OrcResult)]
    L.orc_match_node_pair_g2o(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), arrs[0].shape[0], qid, _p(arrs[3]), _p(arrs[4]),
                              _p(arrs[5]), arrs[3].shape[0], tid, C.byref(params), g2o_iterations, C.byref(out))
    return result_to_dict(out)


def flann_match(qdesc, tdesc, nn_distance_ratio=0.95):
    """Node::featureMatching's FLANN branch (node.cpp:610-667) with exact neighbours: (queryIdx, trainIdx, ratio)."""
    qdesc = np.ascontiguousarray(qdesc, np.float32)
    tdesc = np.ascontiguousarray(tdesc, np.float32)
    L = lib()
    L.orc_flann_match.restype = C.c_int
    L.orc_flann_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
    n = max(qdesc.shape[0], 1)
    mq, mt, md = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
    k = L.orc_flann_match(_p(qdesc), qdesc.shape[0], _p(tdesc), tdesc.shape[0], qdesc.shape[1], nn_distance_ratio,
                          _p(mq), _p(mt), _p(md))
    return mq[:k].copy(), mt[:k].copy(), md[:k].copy()


def match_float_node_pair(qdesc, qxyz1, qid, tdesc, txyz1, tid, nn_distance_ratio=0.95, params=None):
    params = params or default_params()
    qdesc = np.ascontiguousarray(qdesc, np.float32)
    tdesc = np.ascontiguousarray(tdesc, np.float32)
    qxyz1 = np.ascontiguousarray(qxyz1, np.float32)
    txyz1 = np.ascontiguousarray(txyz1, np.float32)
    out = OrcResult()
    dist = np.zeros(ORC_MAX_MATCHES, np.float32)
    L = lib()
    L.orc_match_float_node_pair.restype = None
    L.orc_match_float_node_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int32, C.c_int, C.c_double, C.POINTER(OrcParams), C.POINTER(OrcResult),
                                            C.c_void_p]
    L.orc_match_float_node_pair(_p(qdesc), _p(qxyz1), qdesc.shape[0], qid, _p(tdesc), _p(txyz1), tdesc.shape[0], tid,
                                qdesc.shape[1], nn_distance_ratio, C.byref(params), C.byref(out), _p(dist))
    r = result_to_dict(out)
    r["all_dist"] = dist[: r["n_all"]].copy()
    return r


_ref_graph = None


def ref_graph_lib():
    """The reference's own GraphManager::getPotentialEdgeTargetsWithDijkstra (graph_manager.cpp:204-324), compiled
    from /root/reference with Qt / g2o stand-ins (oracle/ref_stubs/graph_prelude.h), or None."""
    global _ref_graph
    if _ref_graph is None:
        p = os.path.join(_HERE, "_ref", "libref_graph.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        i, vp = C.c_int, C.c_void_p
        R.ref_potential_edge_targets.restype = i
        R.ref_potential_edge_targets.argtypes = [i, vp, vp, vp, i, vp, i, vp, vp, i, i, i, i, i, i, C.c_uint, vp, i]
        _ref_graph = R
    return _ref_graph


def ref_potential_edge_targets(node_ids, vertex_ids, matchable, keyframes, edges, sequential_targets, geodesic_targets,
                               sampled_targets, geodesic_depth, predecessor_id, include_predecessor, srand_seed):
    R = ref_graph_lib()
    nid = np.ascontiguousarray(node_ids, np.int32)
    vid = np.ascontiguousarray(vertex_ids, np.int32)
    mt = np.ascontiguousarray(matchable, np.int32)
    kf = np.ascontiguousarray(keyframes, np.int32)
    ea = np.ascontiguousarray([e[0] for e in edges], np.int32)
    eb = np.ascontiguousarray([e[1] for e in edges], np.int32)
    cap = int(sequential_targets + geodesic_targets + sampled_targets + 2)
    out = np.zeros(cap, np.int32)
    n = R.ref_potential_edge_targets(len(nid), nid.ctypes.data, vid.ctypes.data, mt.ctypes.data, len(kf), kf.ctypes.data,
                                     len(ea), ea.ctypes.data, eb.ctypes.data, int(sequential_targets),
                                     int(geodesic_targets), int(sampled_targets), int(geodesic_depth), int(predecessor_id),
                                     int(bool(include_predecessor)), int(srand_seed), out.ctypes.data, cap)
    return out[:n].copy()


def place_recognition(qdesc, cand_descs, k_neighbours=2, max_hd=128):
    """Descriptor-vote ranking of loop-closure candidates: GraphManager::getNeighbours (loop_closing.cpp:190-277) with
    exact binary neighbours.  Every query descriptor votes `k - rank` (:241) for the k candidate nodes whose best match
    (bruteForceSearchORB's answer, features.cpp:163-182, incl. its last-row quirk) has the smallest Hamming distance
    (ties: the candidate listed first; matches with hd >= max_hd do not vote); a node's votes are divided by its
    descriptor count (:263); nodes are ranked by score, descending (:269; ties: listed first); nodes without votes are
    absent (:243-248).  Returns (positions into cand_descs, float32 scores).  numpy: test infrastructure."""
    nq = len(qdesc)
    n = len(cand_descs)
    hd = np.full((n, nq), 257, np.int64)
    for c, t in enumerate(cand_descs):
        if nq and len(t):
            hd[c] = hamming_nn_batch(qdesc, t)[0]
    votes = np.zeros(n, np.int64)
    key = hd * 65536 + np.arange(n)[:, None]          # (hd, candidate position)
    order = np.argsort(key, axis=0, kind="stable")    # per query descriptor: candidates by (hd, position)
    for rank in range(min(k_neighbours, n)):
        c = order[rank]
        ok = hd[c, np.arange(nq)] < max_hd
        np.add.at(votes, c[ok], k_neighbours - rank)
    rows = np.array([len(t) for t in cand_descs], np.float32)
    pos = np.flatnonzero(votes > 0)
    score = votes[pos].astype(np.float32) / rows[pos]
    o = np.argsort(-score, kind="stable")
    return pos[o].astype(np.int32), score[o].astype(np.float32)
