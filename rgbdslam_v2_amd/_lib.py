"""ctypes binding of librgbdfe.so (the C ABI in include/rgbdfe.h).

The library is built in-tree by ``rgbdslam_v2_amd.build.build()`` (hipcc, gfx950).
There is no CPU fallback anywhere in this package: if the shared library is
missing, or no HIP device is present, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGBDFE_LIB", os.path.join(_HERE, "librgbdfe.so"))  # RGBDFE_LIB: diagnostics builds

RGBDFE_MAX_MATCHES = 320
RGBDFE_MASK_WORDS = 5

DEFAULT_HAMMING_MODE = 3   # RGBDFE_HAMMING_MODE_DEFAULT of include/rgbdfe.h (tests/test_abi.py keeps the two equal)
KERNEL_HAMMING = 0
KERNEL_RANSAC = 1
KERNEL_SIFT_DOT = 2
KERNEL_SIFT_FINISH = 3
KERNEL_EMM = 4


class RgbdfeParams(C.Structure):
    _fields_ = [
        ("max_matches", C.c_int32),
        ("min_matches", C.c_int32),
        ("ransac_iterations", C.c_int32),
        ("max_dist_for_inliers", C.c_float),
        ("depth_cov", C.c_double),
        ("seed", C.c_uint32),
        ("g2o_iterations", C.c_uint32),
    ]


class RgbdfeConfig(C.Structure):
    _fields_ = [
        ("device_id", C.c_int32),
        ("max_nodes", C.c_int32),
        ("max_keypoints", C.c_int32),
        ("max_pairs_per_batch", C.c_int32),
        ("params", RgbdfeParams),
    ]


class RgbdfeMatchResult(C.Structure):
    _fields_ = [
        ("id1", C.c_int32),
        ("id2", C.c_int32),
        ("n_all", C.c_int32),
        ("n_inl", C.c_int32),
        ("rmse", C.c_float),
        ("trafo", C.c_float * 16),
        ("pad0", C.c_uint32),
        ("info_scale", C.c_double),
        ("valid_iterations", C.c_int32),
        ("real_iterations", C.c_int32),
        ("all_q", C.c_uint16 * RGBDFE_MAX_MATCHES),
        ("all_t", C.c_uint16 * RGBDFE_MAX_MATCHES),
        ("all_hd", C.c_uint8 * RGBDFE_MAX_MATCHES),
        ("inlier_mask", C.c_uint64 * RGBDFE_MASK_WORDS),
    ]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4")])

# numpy view of the same POD (used for bulk result handling and the all-gather payload)
RESULT_DTYPE = np.dtype([
    ("id1", "<i4"), ("id2", "<i4"), ("n_all", "<i4"), ("n_inl", "<i4"), ("rmse", "<f4"),
    ("trafo", "<f4", (16,)), ("pad0", "<u4"), ("info_scale", "<f8"),
    ("valid_iterations", "<i4"), ("real_iterations", "<i4"),
    ("all_q", "<u2", (RGBDFE_MAX_MATCHES,)), ("all_t", "<u2", (RGBDFE_MAX_MATCHES,)),
    ("all_hd", "u1", (RGBDFE_MAX_MATCHES,)), ("inlier_mask", "<u8", (RGBDFE_MASK_WORDS,)),
], align=True)

# rgbdfe_compact_result: the record without its all_matches lists (the default multi-GPU gather payload, 144 B)
COMPACT_DTYPE = np.dtype([
    ("id1", "<i4"), ("id2", "<i4"), ("n_all", "<i4"), ("n_inl", "<i4"), ("rmse", "<f4"),
    ("trafo", "<f4", (16,)), ("pad0", "<u4"), ("info_scale", "<f8"),
    ("valid_iterations", "<i4"), ("real_iterations", "<i4"), ("inlier_mask", "<u8", (RGBDFE_MASK_WORDS,)),
], align=True)
COMPACT_FIELDS = [n for n in COMPACT_DTYPE.names]
# rgbdfe_inlier_header: the leading 104 bytes of a record, pad0 = first_inlier
INLIER_HEADER_DTYPE = np.dtype([
    ("id1", "<i4"), ("id2", "<i4"), ("n_all", "<i4"), ("n_inl", "<i4"), ("rmse", "<f4"),
    ("trafo", "<f4", (16,)), ("first_inlier", "<u4"), ("info_scale", "<f8"),
    ("valid_iterations", "<i4"), ("real_iterations", "<i4"),
], align=True)


def compact_of(records: np.ndarray) -> np.ndarray:
    """Host twin of compact_pack_kernel: RESULT_DTYPE records -> COMPACT_DTYPE records (tests)."""
    out = np.zeros(len(records), COMPACT_DTYPE)
    for f in COMPACT_FIELDS:
        out[f] = records[f]
    return out


def inlier_stream_of(records: np.ndarray, n_headers: int):
    """Host twin of rgbdfe_pack_inliers: RESULT_DTYPE records -> (headers INLIER_HEADER_DTYPE [n_headers], list uint32 [total])."""
    n = len(records)
    hdr = np.zeros(n_headers, INLIER_HEADER_DTYPE)
    hdr["id1"][n:] = -1
    hdr["id2"][n:] = -1
    for f in INLIER_HEADER_DTYPE.names:
        if f != "first_inlier":
            hdr[f][:n] = records[f]
    bits = np.unpackbits(np.ascontiguousarray(records["inlier_mask"]).view(np.uint8).reshape(n, -1), axis=1, bitorder="little") \
        if n else np.zeros((0, 64 * RGBDFE_MASK_WORDS), np.uint8)
    counts = bits.sum(1).astype(np.int64)
    first = np.concatenate([[0], np.cumsum(counts)])
    hdr["first_inlier"][:n] = first[:n]
    hdr["first_inlier"][n:] = first[n]
    lst = np.zeros(int(first[n]), np.uint32)
    for k in range(n):
        m = np.nonzero(bits[k])[0]
        lst[first[k]:first[k + 1]] = records["all_q"][k][m].astype(np.uint32) | (records["all_t"][k][m].astype(np.uint32) << 16)
    return hdr, lst


def parse_inlier_stream(buf: np.ndarray, n_headers: int, total: int):
    """A shard's inlier stream (uint8) -> (headers, list uint32 [total])."""
    b = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    hb = n_headers * INLIER_HEADER_DTYPE.itemsize
    return b[:hb].view(INLIER_HEADER_DTYPE), b[hb:hb + 4 * total].view(np.uint32)


def inlier_pairs(hdr, lst, k):
    """(query rows, train rows) of pair k's inlier matches from a parsed inlier stream."""
    a = int(hdr["first_inlier"][k])
    e = lst[a:a + int(hdr["n_inl"][k])]
    return (e & 0xFFFF).astype(np.int32), (e >> 16).astype(np.int32)


_lib = None


class RgbdfeError(RuntimeError):
    pass


def load():
    """Load librgbdfe.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RgbdfeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64; two HIP runtimes in one process cannot both see the
    # GPU.  Loading torch first makes its runtime the process-wide one (librgbdfe.so then binds to it
    # through the soname), whatever order the caller imports things in.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    i32 = C.c_int32
    ctx = vp
    L.rgbdfe_default_config.restype = None
    L.rgbdfe_default_config.argtypes = [C.POINTER(RgbdfeConfig)]
    L.rgbdfe_create.restype = C.c_int
    L.rgbdfe_create.argtypes = [C.POINTER(RgbdfeConfig), C.POINTER(vp)]
    L.rgbdfe_destroy.restype = None
    L.rgbdfe_destroy.argtypes = [ctx]
    L.rgbdfe_set_params.restype = C.c_int
    L.rgbdfe_set_params.argtypes = [ctx, C.POINTER(RgbdfeParams)]
    L.rgbdfe_status_string.restype = C.c_char_p
    L.rgbdfe_status_string.argtypes = [C.c_int]
    L.rgbdfe_last_error.restype = C.c_char_p
    L.rgbdfe_last_error.argtypes = [ctx]
    L.rgbdfe_upload_node.restype = C.c_int
    L.rgbdfe_upload_node.argtypes = [ctx, i32, vp, vp, i32]
    L.rgbdfe_upload_nodes.restype = C.c_int
    L.rgbdfe_upload_nodes.argtypes = [ctx, i32, vp, vp, vp, vp]
    L.rgbdfe_upload_node_device.restype = C.c_int
    L.rgbdfe_upload_node_device.argtypes = [ctx, i32, vp, vp, i32, vp]
    L.rgbdfe_release_node.restype = C.c_int
    L.rgbdfe_release_node.argtypes = [ctx, i32]
    L.rgbdfe_node_count.restype = C.c_int
    L.rgbdfe_node_count.argtypes = [ctx, i32]
    L.rgbdfe_match_node_pairs.restype = C.c_int
    L.rgbdfe_match_node_pairs.argtypes = [ctx, i32, vp, i32, vp]
    L.rgbdfe_match_pair_list.restype = C.c_int
    L.rgbdfe_match_pair_list.argtypes = [ctx, vp, vp, i32, vp]
    L.rgbdfe_match_pair_list_device.restype = C.c_int
    L.rgbdfe_match_pair_list_device.argtypes = [ctx, vp, vp, i32, vp, vp]
    L.rgbdfe_submit_pair_list.restype = C.c_int
    L.rgbdfe_submit_pair_list.argtypes = [ctx, vp, vp, i32, vp, C.POINTER(C.c_int64)]
    L.rgbdfe_wait_ticket.restype = C.c_int
    L.rgbdfe_wait_ticket.argtypes = [ctx, C.c_int64, vp]
    L.rgbdfe_submit_pair_list_host.restype = C.c_int
    L.rgbdfe_submit_pair_list_host.argtypes = [ctx, vp, vp, i32, vp, C.c_size_t, C.c_int, C.POINTER(C.c_int64)]
    L.rgbdfe_wait_host.restype = C.c_int
    L.rgbdfe_wait_host.argtypes = [ctx, C.c_int64, C.POINTER(C.c_int64)]
    L.rgbdfe_wait_host_into.restype = C.c_int
    L.rgbdfe_wait_host_into.argtypes = [ctx, C.c_int64, vp, C.c_size_t, C.POINTER(C.c_int64)]
    L.rgbdfe_upload_sift_node.restype = C.c_int
    L.rgbdfe_upload_sift_node.argtypes = [ctx, i32, vp, vp, i32]
    L.rgbdfe_match_sift_pair_list.restype = C.c_int
    L.rgbdfe_match_sift_pair_list.argtypes = [ctx, vp, vp, i32, vp, vp]
    L.rgbdfe_submit_sift_pair_list.restype = C.c_int
    L.rgbdfe_submit_sift_pair_list.argtypes = [ctx, vp, vp, i32, vp, vp, C.POINTER(C.c_int64)]
    L.rgbdfe_sift_match_nodes.restype = C.c_int
    L.rgbdfe_sift_match_nodes.argtypes = [ctx, i32, i32, vp, vp, vp, C.POINTER(i32)]
    L.rgbdfe_detector_configure.restype = C.c_int
    L.rgbdfe_detector_configure.argtypes = [ctx, i32, i32, i32]
    L.rgbdfe_detector_thresholds.restype = C.c_int
    L.rgbdfe_detector_thresholds.argtypes = [ctx, vp, C.POINTER(i32)]
    L.rgbdfe_detect_describe.restype = C.c_int
    L.rgbdfe_detect_describe.argtypes = [ctx, vp, vp, vp, i32, i32, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_double, vp, vp, vp, C.POINTER(i32)]
    L.rgbdfe_orb_detect.restype = C.c_int
    L.rgbdfe_orb_detect.argtypes = [ctx, vp, vp, i32, i32, i32, vp, i32, C.POINTER(i32)]
    L.rgbdfe_orb_compute.restype = C.c_int
    L.rgbdfe_orb_compute.argtypes = [ctx, vp, i32, i32, vp, i32, vp, C.POINTER(i32)]
    L.rgbdfe_synchronize.restype = C.c_int
    L.rgbdfe_synchronize.argtypes = [ctx]
    L.rgbdfe_hamming_nn_nodes.restype = C.c_int
    L.rgbdfe_hamming_nn_nodes.argtypes = [ctx, i32, i32, vp, vp]
    L.rgbdfe_hamming_nn_host.restype = C.c_int
    L.rgbdfe_hamming_nn_host.argtypes = [ctx, vp, i32, vp, i32, vp, vp]
    L.rgbdfe_project_to_3d.restype = C.c_int
    L.rgbdfe_project_to_3d.argtypes = [ctx, vp, i32, vp, i32, i32, C.c_double, C.c_double,
                                       C.c_double, C.c_double, C.c_double, i32, vp, vp,
                                       C.POINTER(i32)]
    L.rgbdfe_project_to_3d_cloud.restype = C.c_int
    L.rgbdfe_project_to_3d_cloud.argtypes = [ctx, vp, i32, vp, i32, i32, C.c_double, i32, vp, vp, C.POINTER(i32)]
    L.rgbdfe_set_feature_min_depth.restype = C.c_int
    L.rgbdfe_set_feature_min_depth.argtypes = [ctx, i32]
    L.rgbdfe_project_to_3d_min_depth.restype = C.c_int
    L.rgbdfe_project_to_3d_min_depth.argtypes = [ctx, vp, vp, i32, vp, i32, i32, C.c_double, C.c_double, C.c_double,
                                                 C.c_double, C.c_double, i32, vp, vp, C.POINTER(i32)]
    L.rgbdfe_detect_describe_batch.restype = C.c_int
    L.rgbdfe_detect_describe_batch.argtypes = [ctx, i32, vp, vp, vp, i32, i32, C.c_double, C.c_double, C.c_double,
                                               C.c_double, C.c_double, i32, vp, vp, vp, vp]
    L.rgbdfe_detect_describe_batch_nodes.restype = C.c_int
    L.rgbdfe_detect_describe_batch_nodes.argtypes = [ctx, i32, vp, vp, vp, i32, i32, C.c_double, C.c_double, C.c_double,
                                                     C.c_double, C.c_double, i32, vp, vp, vp, vp, vp]
    L.rgbdfe_detect_describe_cloud.restype = C.c_int
    L.rgbdfe_detect_describe_cloud.argtypes = [ctx, vp, vp, vp, i32, i32, C.c_double, vp, vp, vp, C.POINTER(i32)]
    L.rgbdfe_place_recognition.restype = C.c_int
    L.rgbdfe_place_recognition.argtypes = [ctx, i32, vp, i32, i32, i32, i32, vp, vp, C.POINTER(i32)]
    L.rgbdfe_place_recognition_batch.restype = C.c_int
    L.rgbdfe_place_recognition_batch.argtypes = [ctx, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp]
    L.rgbdfe_upload_float_node.restype = C.c_int
    L.rgbdfe_upload_float_node.argtypes = [ctx, i32, vp, i32, vp, i32]
    L.rgbdfe_match_flann_pair_list.restype = C.c_int
    L.rgbdfe_match_flann_pair_list.argtypes = [ctx, vp, vp, i32, C.c_double, vp, vp]
    L.rgbdfe_upload_node_keypoints.restype = C.c_int
    L.rgbdfe_upload_node_keypoints.argtypes = [ctx, i32, vp, i32]
    L.rgbdfe_sift_node_features.restype = C.c_int
    L.rgbdfe_sift_node_features.argtypes = [ctx, vp, i32, vp, vp, i32, i32, C.c_double, C.c_double,
                                            C.c_double, C.c_double, C.c_double, i32, i32, vp, vp, vp, vp,
                                            C.POINTER(i32)]
    L.rgbdfe_sift_node_features_min_depth.restype = C.c_int
    L.rgbdfe_sift_node_features_min_depth.argtypes = [ctx, vp, vp, i32, vp, vp, i32, i32, C.c_double, C.c_double,
                                                      C.c_double, C.c_double, C.c_double, i32, i32, vp, vp, vp, vp,
                                                      C.POINTER(i32)]
    L.rgbdfe_host_register.restype = C.c_int
    L.rgbdfe_host_register.argtypes = [ctx, vp, C.c_size_t]
    L.rgbdfe_host_unregister.restype = C.c_int
    L.rgbdfe_host_unregister.argtypes = [ctx, vp]
    L.rgbdfe_depth_to_mono8.restype = C.c_int
    L.rgbdfe_depth_to_mono8.argtypes = [ctx, vp, i32, i32, i32, vp, vp]
    L.rgbdfe_upload_node_cloud.restype = C.c_int
    L.rgbdfe_upload_node_cloud.argtypes = [ctx, i32, vp, i32, i32, vp, i32, i32, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_double, C.c_double, i32, vp]
    L.rgbdfe_release_node_cloud.restype = C.c_int
    L.rgbdfe_release_node_cloud.argtypes = [ctx, i32]
    L.rgbdfe_observation_likelihood.restype = C.c_int
    L.rgbdfe_observation_likelihood.argtypes = [ctx, i32, vp, vp, vp, i32, vp]
    L.rgbdfe_observation_criterion_met.restype = C.c_int
    L.rgbdfe_observation_criterion_met.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                                   C.POINTER(C.c_double)]
    L.rgbdfe_set_latency_mode.restype = C.c_int
    L.rgbdfe_set_latency_mode.argtypes = [ctx, i32, i32]
    L.rgbdfe_set_profiling.restype = C.c_int
    L.rgbdfe_set_profiling.argtypes = [ctx, C.c_int]
    L.rgbdfe_get_kernel_time.restype = C.c_int
    L.rgbdfe_get_kernel_time.argtypes = [ctx, C.c_int, C.POINTER(C.c_double),
                                         C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.rgbdfe_reset_kernel_time.restype = C.c_int
    L.rgbdfe_reset_kernel_time.argtypes = [ctx]
    L.rgbdfe_set_graph_capture.restype = C.c_int
    L.rgbdfe_set_graph_capture.argtypes = [ctx, C.c_int]
    L.rgbdfe_graph_stats.restype = C.c_int
    L.rgbdfe_graph_stats.argtypes = [ctx, C.POINTER(C.c_int64), C.c_int32]
    L.rgbdfe_create_multi.restype = C.c_int
    L.rgbdfe_create_multi.argtypes = [C.POINTER(RgbdfeConfig), vp, i32, C.POINTER(vp)]
    L.rgbdfe_device_count.restype = C.c_int
    L.rgbdfe_device_count.argtypes = [ctx]
    L.rgbdfe_device_context.restype = vp
    L.rgbdfe_device_context.argtypes = [ctx, i32]
    L.rgbdfe_match_pair_list_allgather.restype = C.c_int
    L.rgbdfe_match_pair_list_allgather.argtypes = [ctx, vp, vp, i32, vp, C.POINTER(i32)]
    L.rgbdfe_match_pair_list_allgather_edges.restype = C.c_int
    L.rgbdfe_match_pair_list_allgather_edges.argtypes = [ctx, vp, vp, i32, vp, vp, vp, C.POINTER(i32)]
    L.rgbdfe_group_submit_us.restype = C.c_int
    L.rgbdfe_group_submit_us.argtypes = [ctx, C.POINTER(C.c_double)]
    L.rgbdfe_gather_transport.restype = C.c_char_p
    L.rgbdfe_gather_transport.argtypes = [ctx]
    L.rgbdfe_gather_exchanges.restype = C.c_int
    L.rgbdfe_gather_exchanges.argtypes = [ctx]
    L.rgbdfe_set_hamming_mode.restype = C.c_int
    L.rgbdfe_set_hamming_mode.argtypes = [ctx, i32]
    L.rgbdfe_match_pair_list_allgather_compact.restype = C.c_int
    L.rgbdfe_match_pair_list_allgather_compact.argtypes = [ctx, vp, vp, i32, vp, C.POINTER(i32)]
    L.rgbdfe_pack_compact.restype = C.c_int
    L.rgbdfe_pack_compact.argtypes = [ctx, vp, i32, vp, vp]
    L.rgbdfe_sizeof_compact_result.restype = C.c_int
    L.rgbdfe_match_pair_list_allgather_inliers.restype = C.c_int
    L.rgbdfe_match_pair_list_allgather_inliers.argtypes = [ctx, vp, vp, i32, vp, C.POINTER(i32), vp, C.POINTER(C.c_int64)]
    L.rgbdfe_pack_inliers.restype = C.c_int
    L.rgbdfe_pack_inliers.argtypes = [ctx, vp, i32, i32, vp, vp, vp]
    L.rgbdfe_sizeof_inlier_header.restype = C.c_int
    if L.rgbdfe_sizeof_inlier_header() != INLIER_HEADER_DTYPE.itemsize:
        raise RgbdfeError("rgbdfe_inlier_header layout mismatch between librgbdfe.so and the binding")
    if L.rgbdfe_sizeof_compact_result() != COMPACT_DTYPE.itemsize:
        raise RgbdfeError("rgbdfe_compact_result layout mismatch between librgbdfe.so and the binding")
    L.rgbdfe_sift_detect.restype = C.c_int
    L.rgbdfe_sift_detect.argtypes = [ctx, vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
    L.rgbdfe_sift_describe.restype = C.c_int
    L.rgbdfe_sift_describe.argtypes = [ctx, vp, i32, i32, vp, i32, vp]
    L.rgbdfe_sift_detect_batch.restype = C.c_int
    L.rgbdfe_sift_detect_batch.argtypes = [ctx, i32, vp, i32, i32, i32, i32, vp, vp, vp]
    L.rgbdfe_sift_geometry.restype = C.c_int
    L.rgbdfe_sift_geometry.argtypes = [ctx] + [C.POINTER(i32)] * 4
    L.rgbdfe_sift_debug_plane.restype = C.c_int
    L.rgbdfe_sift_debug_plane.argtypes = [ctx, i32, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.rgbdfe_sift_debug_candidates.restype = C.c_int
    L.rgbdfe_sift_debug_candidates.argtypes = [ctx, i32, i32, vp, i32, C.POINTER(i32)]
    L.rgbdfe_sizeof_match_result.restype = C.c_int
    L.rgbdfe_abi_version.restype = C.c_int
    if L.rgbdfe_sizeof_match_result() != C.sizeof(RgbdfeMatchResult) or \
            RESULT_DTYPE.itemsize != C.sizeof(RgbdfeMatchResult):
        raise RgbdfeError("rgbdfe_match_result layout mismatch between librgbdfe.so and the binding")
    RAND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
    L.rgbdfe_rand_fn = RAND_FN
    L.rgbdfe_pose_graph_create.restype = vp
    L.rgbdfe_pose_graph_create.argtypes = []
    L.rgbdfe_pose_graph_destroy.restype = None
    L.rgbdfe_pose_graph_destroy.argtypes = [vp]
    L.rgbdfe_pose_graph_add_node.restype = C.c_int
    L.rgbdfe_pose_graph_add_node.argtypes = [vp, i32, i32, i32, i32]
    L.rgbdfe_pose_graph_add_edge.restype = C.c_int
    L.rgbdfe_pose_graph_add_edge.argtypes = [vp, i32, i32]
    L.rgbdfe_pose_graph_set_matchable.restype = C.c_int
    L.rgbdfe_pose_graph_set_matchable.argtypes = [vp, i32, i32]
    L.rgbdfe_potential_edge_targets.restype = C.c_int
    L.rgbdfe_potential_edge_targets.argtypes = [vp, i32, i32, i32, i32, i32, i32, RAND_FN, vp, C.c_uint32, vp, i32,
                                                C.POINTER(i32)]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "rgbdfe_default_config", "rgbdfe_create", "rgbdfe_destroy", "rgbdfe_set_params",
    "rgbdfe_status_string", "rgbdfe_last_error", "rgbdfe_upload_node", "rgbdfe_upload_nodes",
    "rgbdfe_upload_node_device", "rgbdfe_release_node", "rgbdfe_node_count",
    "rgbdfe_match_node_pairs", "rgbdfe_match_pair_list", "rgbdfe_match_pair_list_device",
    "rgbdfe_detector_configure", "rgbdfe_detector_thresholds", "rgbdfe_detect_describe",
    "rgbdfe_orb_detect", "rgbdfe_orb_compute",
    "rgbdfe_submit_pair_list", "rgbdfe_wait_ticket", "rgbdfe_submit_pair_list_host", "rgbdfe_wait_host", "rgbdfe_wait_host_into", "rgbdfe_upload_sift_node",
    "rgbdfe_match_sift_pair_list", "rgbdfe_submit_sift_pair_list", "rgbdfe_sift_match_nodes",
    "rgbdfe_synchronize", "rgbdfe_hamming_nn_nodes", "rgbdfe_hamming_nn_host",
    "rgbdfe_project_to_3d", "rgbdfe_sift_node_features", "rgbdfe_sift_node_features_min_depth", "rgbdfe_depth_to_mono8",
    "rgbdfe_host_register", "rgbdfe_host_unregister",
    "rgbdfe_upload_node_cloud", "rgbdfe_release_node_cloud", "rgbdfe_observation_likelihood",
    "rgbdfe_observation_criterion_met", "rgbdfe_set_latency_mode", "rgbdfe_set_profiling", "rgbdfe_get_kernel_time",
    "rgbdfe_reset_kernel_time", "rgbdfe_graph_stats", "rgbdfe_set_graph_capture", "rgbdfe_match_pair_list_allgather_inliers", "rgbdfe_pack_inliers", "rgbdfe_sizeof_inlier_header", "rgbdfe_sizeof_match_result", "rgbdfe_abi_version",
    "rgbdfe_pose_graph_create", "rgbdfe_pose_graph_destroy", "rgbdfe_pose_graph_add_node",
    "rgbdfe_pose_graph_add_edge", "rgbdfe_pose_graph_set_matchable", "rgbdfe_potential_edge_targets",
    "rgbdfe_create_multi", "rgbdfe_device_count", "rgbdfe_device_context", "rgbdfe_match_pair_list_allgather",
    "rgbdfe_gather_transport", "rgbdfe_gather_exchanges", "rgbdfe_set_hamming_mode", "rgbdfe_project_to_3d_cloud", "rgbdfe_detect_describe_cloud",
    "rgbdfe_detect_describe_batch", "rgbdfe_detect_describe_batch_nodes", "rgbdfe_match_pair_list_allgather_edges",
    "rgbdfe_set_feature_min_depth", "rgbdfe_project_to_3d_min_depth",
    "rgbdfe_place_recognition", "rgbdfe_place_recognition_batch", "rgbdfe_upload_float_node",
    "rgbdfe_match_flann_pair_list", "rgbdfe_upload_node_keypoints",
    "rgbdfe_match_pair_list_allgather_compact", "rgbdfe_pack_compact", "rgbdfe_sizeof_compact_result",
    "rgbdfe_group_submit_us", "rgbdfe_sift_detect", "rgbdfe_sift_detect_batch", "rgbdfe_sift_describe", "rgbdfe_sift_geometry", "rgbdfe_sift_debug_plane", "rgbdfe_sift_debug_candidates",
]
