"""Host-side mirror of the reference's front-end interface over the C ABI.

The reference's seam is a set of C++ methods (there is no FFI):
``Node::matchNodePair`` (src/node.h:85), ``Node::featureMatching`` (:117),
``Node::getRelativeTransformationTo`` (:102), ``bruteForceSearchORB``
(src/features.h:14) and ``GraphManager::nodeComparisons``'s
``QtConcurrent::blockingMapped`` fan-out (src/graph_manager.cpp:541-548).
This module keeps those names and argument meanings so the parity tests read
like tests of the reference; all arithmetic happens in librgbdfe.so on the GPU.
"""
import ctypes as C
import threading
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import DEFAULT_HAMMING_MODE, RESULT_DTYPE, RgbdfeConfig, RgbdfeError, RgbdfeParams


@dataclass
class DMatch:
    """cv::DMatch as the reference fills it (node.cpp:573): distance = hd/256 (jitter dropped)."""
    queryIdx: int
    trainIdx: int
    distance: float


@dataclass
class LoadedEdge3D:
    """src/edge.h:24-32"""
    id1: int = -1
    id2: int = -1
    transform: np.ndarray = field(default_factory=lambda: np.eye(4))
    informationMatrix: np.ndarray = field(default_factory=lambda: np.zeros((6, 6)))


@dataclass
class MatchingResult:
    """src/matching_result.h:24-46"""
    inlier_matches: List[DMatch] = field(default_factory=list)
    all_matches: List[DMatch] = field(default_factory=list)
    edge: LoadedEdge3D = field(default_factory=LoadedEdge3D)
    rmse: float = 0.0
    ransac_trafo: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    final_trafo: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    valid_iterations: int = 0
    real_iterations: int = 0


def inlier_indices(rec) -> np.ndarray:
    """Positions (into all_*) of the inlier matches of one RESULT_DTYPE record."""
    bits = np.unpackbits(np.asarray(rec["inlier_mask"]).view(np.uint8), bitorder="little")
    return np.flatnonzero(bits[: int(rec["n_all"])]).astype(np.int32)


def record_to_matching_result(rec) -> MatchingResult:
    n_all = int(rec["n_all"])
    q = rec["all_q"][:n_all].astype(np.int32)
    t = rec["all_t"][:n_all].astype(np.int32)
    d = rec["all_hd"][:n_all].astype(np.float32) / np.float32(256.0)
    all_matches = [DMatch(int(a), int(b), float(c)) for a, b, c in zip(q, t, d)]
    inl = inlier_indices(rec)
    T = np.array(rec["trafo"], np.float32).reshape(4, 4).T.copy()  # column-major storage
    mr = MatchingResult(all_matches=all_matches, inlier_matches=[all_matches[i] for i in inl],
                        rmse=float(rec["rmse"]), ransac_trafo=T, final_trafo=T.copy(),
                        valid_iterations=int(rec["valid_iterations"]),
                        real_iterations=int(rec["real_iterations"]))
    mr.edge.id1, mr.edge.id2 = int(rec["id1"]), int(rec["id2"])
    if mr.edge.id1 >= 0:
        mr.edge.transform = T.astype(np.float64)                        # node.cpp:1339
        mr.edge.informationMatrix = np.eye(6) * float(rec["info_scale"])  # node.cpp:1335
    return mr


class FrontEnd:
    """One rgbdfe context = one GPU: resident node features + the batched pair op."""

    def __init__(self, device_id: int = 0, max_nodes: int = 256, max_keypoints: int = 1024,
                 max_pairs_per_batch: int = 4096, device_ids: Optional[Sequence[int]] = None, **params):
        """device_ids: several GPUs behind this one handle (rgbdfe_create_multi): node features replicated,
        pairs sharded pair k -> device k mod G, results gathered; a device may be listed twice."""
        self._L = _lib.load()
        self._batch_lock = threading.Lock()
        cfg = RgbdfeConfig()
        self._L.rgbdfe_default_config(C.byref(cfg))
        cfg.device_id = device_id
        cfg.max_nodes = max_nodes
        cfg.max_keypoints = max_keypoints
        cfg.max_pairs_per_batch = max_pairs_per_batch
        for k, v in params.items():
            if not hasattr(cfg.params, k):
                raise TypeError(f"unknown front-end parameter {k!r}")
            setattr(cfg.params, k, v)
        self.cfg = cfg
        self._ctx = C.c_void_p()
        if device_ids is not None:
            ids = np.ascontiguousarray(device_ids, np.int32)
            st = self._L.rgbdfe_create_multi(C.byref(cfg), ids.ctypes.data, len(ids), C.byref(self._ctx))
        else:
            st = self._L.rgbdfe_create(C.byref(cfg), C.byref(self._ctx))
        if st != 0:
            self._ctx = C.c_void_p()
            raise RgbdfeError(f"rgbdfe_create failed: {self._L.rgbdfe_status_string(st).decode()}")

    @property
    def device_count(self) -> int:
        return int(self._L.rgbdfe_device_count(self._ctx))

    def set_hamming_mode(self, mode: int):
        """0 = popcount kernel, 1 = fp4 MFMA kernel, 2 = MFMA kernel with the VALU row term, 3 = the MFMA kernel as a
        software pipeline inside every wave (include/rgbdfe.h; _lib.DEFAULT_HAMMING_MODE is what a new context uses)."""
        self._check(self._L.rgbdfe_set_hamming_mode(self._ctx, mode))
        self._hamming_mode = int(mode)

    @property
    def hamming_mode(self) -> int:
        """The Hamming kernel in use (the library falls back to 0 above 32768 keypoints per node)."""
        import os
        m = getattr(self, "_hamming_mode", None)
        if m is None:
            try:
                m = int(os.environ.get("RGBDFE_HAMMING_MODE", str(DEFAULT_HAMMING_MODE)))
            except ValueError:
                m = DEFAULT_HAMMING_MODE
            m = m if 0 <= m <= 3 else DEFAULT_HAMMING_MODE
        return 0 if self.cfg.max_keypoints > 32768 else m

    def group_submit_us(self) -> float:
        """Host microseconds the calling thread spent enqueueing the latest sharded batch on all devices."""
        v = C.c_double(0.0)
        self._check(self._L.rgbdfe_group_submit_us(self._ctx, C.byref(v)))
        return float(v.value)

    def gather_transport(self) -> str:
        return self._L.rgbdfe_gather_transport(self._ctx).decode()

    def gather_exchanges(self) -> int:
        """Exchanges the latest match_pair_list_allgather_inliers issued (1: one collective, no host read in front of it)."""
        return int(self._L.rgbdfe_gather_exchanges(self._ctx))

    def match_pair_list_allgather(self, query_ids, train_ids, d_out_ptrs: Sequence[int]) -> int:
        """All results on every device (ncclAllGather / peer copies).  d_out_ptrs: one device buffer per device,
        each device_count * ceil(n / device_count) records.  Returns records per device."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        ptrs = (C.c_void_p * len(d_out_ptrs))(*[C.c_void_p(int(p)) for p in d_out_ptrs])
        per = C.c_int32(0)
        self._check(self._L.rgbdfe_match_pair_list_allgather(self._ctx, q.ctypes.data, t.ctypes.data, len(q),
                                                             C.cast(ptrs, C.c_void_p), C.byref(per)))
        return int(per.value)

    def match_pair_list_allgather_compact(self, query_ids, train_ids, d_out_ptrs: Sequence[int]) -> int:
        """rgbdfe_match_pair_list_allgather with COMPACT_DTYPE records (144 B: no all_matches lists) as the payload."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        ptrs = (C.c_void_p * len(d_out_ptrs))(*[C.c_void_p(int(p)) for p in d_out_ptrs])
        per = C.c_int32(0)
        self._check(self._L.rgbdfe_match_pair_list_allgather_compact(self._ctx, q.ctypes.data, t.ctypes.data, len(q),
                                                                     C.cast(ptrs, C.c_void_p), C.byref(per)))
        return int(per.value)

    def match_pair_list_allgather_inliers(self, query_ids, train_ids, d_out_ptrs: Sequence[int]):
        """rgbdfe_match_pair_list_allgather_inliers: every device buffer ends up with every device's inlier stream.
        Returns (records per device, list entries per device, stride in bytes)."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        g = len(d_out_ptrs)
        ptrs = (C.c_void_p * g)(*[C.c_void_p(int(p)) for p in d_out_ptrs])
        per, stride = C.c_int32(0), C.c_int64(0)
        totals = np.zeros(g, np.int32)
        self._check(self._L.rgbdfe_match_pair_list_allgather_inliers(self._ctx, q.ctypes.data, t.ctypes.data, len(q),
                                                                     C.cast(ptrs, C.c_void_p), C.byref(per), totals.ctypes.data,
                                                                     C.byref(stride)))
        return int(per.value), totals, int(stride.value)

    def pack_compact(self, d_records_ptr: int, n: int, d_compact_ptr: int, stream: Optional[int] = None):
        """n records in HBM -> n compact records in HBM on `stream` (rgbdfe_pack_compact)."""
        self._check(self._L.rgbdfe_pack_compact(self._ctx, C.c_void_p(int(d_records_ptr)), int(n),
                                                C.c_void_p(int(d_compact_ptr)), C.c_void_p(stream or 0)))

    def pack_inliers(self, d_records_ptr: int, n: int, n_headers: int, d_stream_ptr: int, d_total_ptr: int,
                     stream: Optional[int] = None):
        """n records in HBM -> the shard's inlier stream in HBM on `stream` (rgbdfe_pack_inliers: n_headers headers of 104
        bytes, then query | train << 16 of every inlier; the entry count goes to the device int32 at d_total_ptr)."""
        self._check(self._L.rgbdfe_pack_inliers(self._ctx, C.c_void_p(int(d_records_ptr)), int(n), int(n_headers),
                                                C.c_void_p(int(d_stream_ptr)), C.c_void_p(int(d_total_ptr)),
                                                C.c_void_p(stream or 0)))

    def match_pair_list_allgather_edges(self, query_ids, train_ids, d_out_ptrs: Sequence[int], d_index_ptrs=None):
        """Only the accepted edges travel (rgbdfe_match_pair_list_allgather_edges).  Returns (counts per device, stride):
        device i's edges sit at records [i * stride, i * stride + counts[i]) of every buffer, their positions in the pair
        list at the same offsets of the index buffers."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        g = len(d_out_ptrs)
        ptrs = (C.c_void_p * g)(*[C.c_void_p(int(p)) for p in d_out_ptrs])
        iptrs = None if d_index_ptrs is None else (C.c_void_p * g)(*[C.c_void_p(int(p)) for p in d_index_ptrs])
        counts = np.zeros(g, np.int32)
        stride = C.c_int32(0)
        self._check(self._L.rgbdfe_match_pair_list_allgather_edges(
            self._ctx, q.ctypes.data, t.ctypes.data, len(q), C.cast(ptrs, C.c_void_p),
            None if iptrs is None else C.cast(iptrs, C.c_void_p), counts.ctypes.data, C.byref(stride)))
        return counts, int(stride.value)

    # -- plumbing --------------------------------------------------------------
    def _check(self, st):
        if st != 0:
            msg = self._L.rgbdfe_last_error(self._ctx).decode()
            raise RgbdfeError(f"{self._L.rgbdfe_status_string(st).decode()}: {msg}")

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.rgbdfe_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def params(self) -> RgbdfeParams:
        return self.cfg.params

    def set_params(self, **params):
        for k, v in params.items():
            setattr(self.cfg.params, k, v)
        self._check(self._L.rgbdfe_set_params(self._ctx, C.byref(self.cfg.params)))

    # -- node residency ----------------------------------------------------------
    def upload_node(self, node_id: int, desc: np.ndarray, xyz1: np.ndarray):
        desc = np.ascontiguousarray(desc, np.uint8)
        xyz1 = np.ascontiguousarray(xyz1, np.float32)
        n = desc.shape[0]
        if desc.ndim != 2 or desc.shape[1] != 32 or xyz1.shape != (n, 4):
            raise ValueError("desc must be [n,32] uint8 and xyz1 [n,4] float32")
        self._check(self._L.rgbdfe_upload_node(self._ctx, node_id, desc.ctypes.data,
                                               xyz1.ctypes.data, n))

    def upload_node_device(self, node_id: int, d_desc_ptr: int, d_xyz1_ptr: int, n: int,
                           stream: Optional[int] = None):
        self._check(self._L.rgbdfe_upload_node_device(self._ctx, node_id, d_desc_ptr, d_xyz1_ptr,
                                                      n, stream))

    def upload_node_keypoints(self, node_id: int, kp_xy: np.ndarray):
        """Node::feature_locations_2d_ (KeyPoint.pt) of a resident node, for the g2o refinement."""
        kp_xy = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
        self._check(self._L.rgbdfe_upload_node_keypoints(self._ctx, node_id, kp_xy.ctypes.data, kp_xy.shape[0]))

    def release_node(self, node_id: int):
        self._check(self._L.rgbdfe_release_node(self._ctx, node_id))

    def node_count(self, node_id: int) -> int:
        return self._L.rgbdfe_node_count(self._ctx, node_id)

    # -- the pair op ---------------------------------------------------------------
    def match_pair_list(self, query_ids: Sequence[int], train_ids: Sequence[int]) -> np.ndarray:
        """Batched Node::matchNodePair; returns a RESULT_DTYPE array (one record per pair)."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        if q.shape != t.shape or q.ndim != 1:
            raise ValueError("query_ids / train_ids must be 1-D and of equal length")
        out = np.zeros(q.shape[0], RESULT_DTYPE)
        self._check(self._L.rgbdfe_match_pair_list(self._ctx, q.ctypes.data, t.ctypes.data,
                                                   q.shape[0], out.ctypes.data))
        return out

    def match_node_pairs(self, new_node_id: int, candidate_ids: Sequence[int]) -> np.ndarray:
        """1:1 replacement of blockingMapped(nodes_to_comp, matchNodePair) (graph_manager.cpp:548)."""
        c = np.ascontiguousarray(candidate_ids, np.int32)
        out = np.zeros(c.shape[0], RESULT_DTYPE)
        self._check(self._L.rgbdfe_match_node_pairs(self._ctx, new_node_id, c.ctypes.data,
                                                    c.shape[0], out.ctypes.data))
        return out

    def match_pair_list_device(self, query_ids: np.ndarray, train_ids: np.ndarray,
                               d_out_ptr: int, stream: Optional[int] = None):
        """Asynchronous variant: results stay in HBM at d_out_ptr (n x sizeof(result))."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        self._check(self._L.rgbdfe_match_pair_list_device(self._ctx, q.ctypes.data, t.ctypes.data,
                                                          q.shape[0], d_out_ptr, stream))

    def submit_pair_list(self, query_ids: np.ndarray, train_ids: np.ndarray, d_out_ptr: int) -> int:
        """Pipelined variant: enqueue on an internal stream, return a ticket (see wait_ticket)."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        ticket = C.c_int64(0)
        self._check(self._L.rgbdfe_submit_pair_list(self._ctx, q.ctypes.data, t.ctypes.data,
                                                    q.shape[0], d_out_ptr, C.byref(ticket)))
        return ticket.value

    def wait_ticket(self, ticket: int, stream: Optional[int] = None):
        """Make `stream` (hipStream_t as int) wait for the batch, or block the host if None."""
        self._check(self._L.rgbdfe_wait_ticket(self._ctx, ticket, stream))

    def submit_pair_list_host(self, query_ids, train_ids, out: np.ndarray, inliers: bool = False) -> int:
        """rgbdfe_submit_pair_list_host: results into the HOST array `out` (RESULT_DTYPE records, or -- inliers=True -- a
        uint8 buffer for the batch's inlier stream); returns the ticket for wait_host.  `out` must stay alive and untouched
        until then; pinned arrays (host_register) are written by the download itself."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        if not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be contiguous")
        ticket = C.c_int64(0)
        self._check(self._L.rgbdfe_submit_pair_list_host(self._ctx, q.ctypes.data, t.ctypes.data, q.shape[0], out.ctypes.data,
                                                         out.nbytes, 1 if inliers else 0, C.byref(ticket)))
        return ticket.value

    def wait_host(self, ticket: int) -> int:
        """rgbdfe_wait_host: block until the job's results are in its `out`; returns the payload's size in bytes."""
        nb = C.c_int64(0)
        self._check(self._L.rgbdfe_wait_host(self._ctx, ticket, C.byref(nb)))
        return nb.value

    def wait_host_into(self, ticket: int, out) -> int:
        """rgbdfe_wait_host_into: collect a job whose payload did not fit its buffer into `out` (None: drop the job)."""
        nb = C.c_int64(0)
        if out is None:
            self._check(self._L.rgbdfe_wait_host_into(self._ctx, ticket, None, 0, C.byref(nb)))
            return 0
        if not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be contiguous")
        self._check(self._L.rgbdfe_wait_host_into(self._ctx, ticket, out.ctypes.data, out.nbytes, C.byref(nb)))
        return nb.value

    def synchronize(self):
        self._check(self._L.rgbdfe_synchronize(self._ctx))

    # -- per-frame feature path (Node::Node, node.cpp:139-210) ------------------------------
    def detector_configure(self, max_keypoints=600, grid_resolution=3, adjuster_max_iterations=5):
        """createDetector("ORB") (features.cpp:63-113): resets the per-cell FAST thresholds."""
        self._max_keypoints = max_keypoints
        self._check(self._L.rgbdfe_detector_configure(self._ctx, max_keypoints, grid_resolution,
                                                      adjuster_max_iterations))

    def detector_thresholds(self):
        t = np.zeros(64, np.float64)
        n = C.c_int32(0)
        self._check(self._L.rgbdfe_detector_thresholds(self._ctx, t.ctypes.data, C.byref(n)))
        return t[: n.value].copy()

    def host_register(self, array):
        """Page-locks a numpy array's memory (rgbdfe_host_register): detect_describe then copies it to the device directly
        instead of through the library's staging buffer.  Keep the array alive and call host_unregister before freeing it."""
        self._check(self._L.rgbdfe_host_register(self._ctx, array.ctypes.data, array.nbytes))

    def host_unregister(self, array):
        self._check(self._L.rgbdfe_host_unregister(self._ctx, array.ctypes.data))

    def detect_describe(self, gray, mask, depth, fx, fy, cx, cy, depth_scaling=1.0):
        """detect -> removeDepthless -> retainBest -> compute -> projectTo3D for one frame.
        Returns (keypoints KEYPOINT_DTYPE, descriptors [n,32] uint8, xyz1 [n,4] float32)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        cap = getattr(self, "_max_keypoints", 600)
        kp = np.zeros(cap, _lib.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        xyz = np.zeros((cap, 4), np.float32)
        n = C.c_int32(0)
        self._check(self._L.rgbdfe_detect_describe(
            self._ctx, gray.ctypes.data, None if m is None else m.ctypes.data, depth.ctypes.data,
            gray.shape[0], gray.shape[1], fx, fy, cx, cy, depth_scaling, kp.ctypes.data, desc.ctypes.data,
            xyz.ctypes.data, C.byref(n)))
        return kp[: n.value].copy(), desc[: n.value].copy(), xyz[: n.value].copy()

    def set_feature_min_depth(self, on: bool):
        """Parameter "use_feature_min_depth" (parameter_server.cpp:90) for detect_describe(_batch)."""
        self._check(self._L.rgbdfe_set_feature_min_depth(self._ctx, 1 if on else 0))

    def project_to_3d_min_depth(self, kp_xy, kp_size, depth, fx, fy, cx, cy, depth_scaling=1.0, max_keypoints=1000):
        """projectTo3D with use_feature_min_depth (node.cpp:940): returns (kept_idx, xyz1)."""
        kp_xy = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
        kp_size = np.ascontiguousarray(kp_size, np.float32)
        depth = np.ascontiguousarray(depth, np.float32)
        n = kp_xy.shape[0]
        kept = np.zeros(max(n, 1), np.int32)
        xyz = np.zeros((max(n, 1), 4), np.float32)
        k = C.c_int32(0)
        self._check(self._L.rgbdfe_project_to_3d_min_depth(
            self._ctx, kp_xy.ctypes.data, kp_size.ctypes.data, n, depth.ctypes.data, depth.shape[0], depth.shape[1],
            fx, fy, cx, cy, depth_scaling, max_keypoints, kept.ctypes.data, xyz.ctypes.data, C.byref(k)))
        return kept[: k.value].copy(), xyz[: k.value].copy()

    def detect_describe_batch(self, grays, masks, depths, fx, fy, cx, cy, depth_scaling=1.0, node_ids=None, copy=True):
        """A run of frames through the same detector state, in order (rgbdfe_detect_describe_batch): the results of
        calling detect_describe frame by frame, with frame k+1's upload overlapped with frame k's detection.
        Returns a list of (keypoints, descriptors, xyz1) per frame.  node_ids: frame f's features also become the resident
        node node_ids[f] (rgbdfe_detect_describe_batch_nodes; a negative id: no node).  copy=False returns views of the
        output arrays this object keeps and reuses for its next call (what an integration with its own buffers does;
        sift_detect_batch has the same switch): valid until then."""
        n = len(grays)
        if n == 0:
            return []
        g = [np.ascontiguousarray(x, np.uint8) for x in grays]
        d = [np.ascontiguousarray(x, np.float32) for x in depths]
        m = [None if (masks is None or masks[i] is None) else np.ascontiguousarray(masks[i], np.uint8) for i in range(n)]
        rows, cols = g[0].shape
        for a in g + d + [x for x in m if x is not None]:
            if a.shape != (rows, cols):
                raise ValueError("all frames of a batch share one size")
        cap = getattr(self, "_max_keypoints", 600)
        # the ABI's output arrays (n * cap rows each: 8.7 MB for 112 frames of 1024) are scratch of this object, kept between
        # calls of the same shape: fresh np.zeros arrays of that size are fresh mmap'ed pages every call, i.e. ~2000 page
        # faults inside the library's result copies (10 us per frame of a 640 x 480 run); what is returned are copies of the
        # rows in use
        with self._batch_lock:   # (the scratch is one per object)
            key = (n, cap)
            if getattr(self, "_batch_out_key", None) != key:
                self._batch_out = (np.zeros((n, cap), _lib.KEYPOINT_DTYPE), np.zeros((n, cap, 32), np.uint8),
                                   np.zeros((n, cap, 4), np.float32), np.zeros(n, np.int32))
                self._batch_out_key = key
            kp, desc, xyz, cnt = self._batch_out
            cnt[:] = 0
            vp = C.c_void_p * n
            pg = vp(*[x.ctypes.data for x in g])
            pd = vp(*[x.ctypes.data for x in d])
            pm = vp(*[None if x is None else x.ctypes.data for x in m])
            if node_ids is not None:
                ids = np.ascontiguousarray(node_ids, np.int32)
                if ids.shape != (n,):
                    raise ValueError("node_ids must hold one id per frame")
                self._check(self._L.rgbdfe_detect_describe_batch_nodes(
                    self._ctx, n, C.cast(pg, C.c_void_p), C.cast(pm, C.c_void_p), C.cast(pd, C.c_void_p), rows, cols, fx, fy, cx, cy,
                    depth_scaling, cap, kp.ctypes.data, desc.ctypes.data, xyz.ctypes.data, cnt.ctypes.data, ids.ctypes.data))
            else:
                self._check(self._L.rgbdfe_detect_describe_batch(
                    self._ctx, n, C.cast(pg, C.c_void_p), C.cast(pm, C.c_void_p), C.cast(pd, C.c_void_p), rows, cols, fx, fy, cx, cy,
                    depth_scaling, cap, kp.ctypes.data, desc.ctypes.data, xyz.ctypes.data, cnt.ctypes.data))
            if not copy:
                return [(kp[f, : cnt[f]], desc[f, : cnt[f]], xyz[f, : cnt[f]]) for f in range(n)]
            return [(kp[f, : cnt[f]].copy(), desc[f, : cnt[f]].copy(), xyz[f, : cnt[f]].copy()) for f in range(n)]

    def orb_detect(self, gray, mask, fast_threshold, capacity=60000):
        """cv::ORB::create(10000,1.2,8,15,0,2,HARRIS,31,thr)->detect(gray, kps, mask) (feature_adjuster.cpp:94)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        kp = np.zeros(capacity, _lib.KEYPOINT_DTYPE)
        n = C.c_int32(0)
        self._check(self._L.rgbdfe_orb_detect(self._ctx, gray.ctypes.data, None if m is None else m.ctypes.data,
                                              gray.shape[0], gray.shape[1], fast_threshold, kp.ctypes.data,
                                              capacity, C.byref(n)))
        return kp[: n.value].copy()

    def orb_compute(self, gray, keypoints):
        """cv::ORB::create()->compute(gray, kps, desc) (features.cpp:117-119)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        kp = np.ascontiguousarray(keypoints.copy())
        desc = np.zeros((max(len(kp), 1), 32), np.uint8)
        n = C.c_int32(0)
        self._check(self._L.rgbdfe_orb_compute(self._ctx, gray.ctypes.data, gray.shape[0], gray.shape[1],
                                               kp.ctypes.data, len(kp), desc.ctypes.data, C.byref(n)))
        return kp[: n.value].copy(), desc[: n.value].copy()

    # -- SIFT (128-d float descriptors, matcher_type == SIFTGPU) ---------------------------
    def sift_detect(self, gray, mask=None, max_keypoints: int = 1000):
        """SiftGPUWrapper::detect (sift_gpu_wrapper.cpp:113-167): (keypoints KEYPOINT_DTYPE [n], descriptors [n, 128] f32,
        unnormalised) of a mono8 image.  `mask` is ignored, as in the reference."""
        gray = np.ascontiguousarray(gray, np.uint8)
        rows, cols = gray.shape
        cap = min(max(2 * int(max_keypoints) + 4096, 8192), 1 << 16)
        max_keypoints = min(int(max_keypoints), (1 << 31) - 1)
        while True:
            kp = np.zeros(cap, _lib.KEYPOINT_DTYPE)
            desc = np.zeros((cap, 128), np.float32)
            n = C.c_int32(0)
            st = self._L.rgbdfe_sift_detect(self._ctx, gray.ctypes.data, None, rows, cols, int(max_keypoints), kp.ctypes.data,
                                            desc.ctypes.data, cap, C.byref(n))
            if st == -5 and n.value > cap:        # RGBDFE_ERR_CAPACITY: *n_out = the number of features
                cap = n.value + 64
                continue
            self._check(st)
            return kp[: n.value].copy(), desc[: n.value].copy()

    def sift_describe(self, gray, keypoints):
        """SiftGPUWrapper::detect with a given keypoint list (sift_gpu_wrapper.cpp:132-142): descriptors at the keypoints'
        positions, sizes and angles.  Returns (keypoints as the wrapper rebuilds them, descriptors [n, 128])."""
        gray = np.ascontiguousarray(gray, np.uint8)
        kp = np.ascontiguousarray(keypoints.copy())
        n = len(kp)
        desc = np.zeros((max(n, 1), 128), np.float32)
        self._check(self._L.rgbdfe_sift_describe(self._ctx, gray.ctypes.data, gray.shape[0], gray.shape[1], kp.ctypes.data, n,
                                                 desc.ctypes.data))
        return kp, desc[:n]

    def sift_detect_batch(self, grays, max_keypoints: int = 1000, out_stride=None, copy=True):
        """rgbdfe_sift_detect_batch: sift_detect over a run of frames of one size, up to 8 frames per launch chain.
        Returns a list of (keypoints, descriptors) per frame.  copy=False returns views of output arrays this object keeps
        and reuses for the next call of the same shape (what an integration with its own buffers does)."""
        n = len(grays)
        if n == 0:
            return []
        g = [np.ascontiguousarray(x, np.uint8) for x in grays]
        rows, cols = g[0].shape
        for a in g:
            if a.shape != (rows, cols):
                raise ValueError("all frames of a batch share one size")
        max_keypoints = min(int(max_keypoints), (1 << 31) - 1)
        stride = int(out_stride) if out_stride else min(2 * max_keypoints + 1024, 1 << 16)
        while True:
            cache = getattr(self, "_sift_out", None)
            if copy or cache is None or cache[0] != (n, stride):
                kp = np.zeros((n, stride), _lib.KEYPOINT_DTYPE)
                desc = np.zeros((n, stride, 128), np.float32)
                if not copy:
                    self._sift_out = ((n, stride), kp, desc)
            else:
                kp, desc = cache[1], cache[2]
            cnt = np.zeros(n, np.int32)
            pg = (C.c_void_p * n)(*[x.ctypes.data for x in g])
            rc = self._L.rgbdfe_sift_detect_batch(self._ctx, n, C.cast(pg, C.c_void_p), rows, cols, max_keypoints, stride,
                                                  kp.ctypes.data, desc.ctypes.data, cnt.ctypes.data)
            if rc == -5 and int(cnt.max()) > stride and not out_stride:   # RGBDFE_ERR_CAPACITY: cnt holds the counts
                stride = int(cnt.max())
                continue
            self._check(rc)
            if not copy:
                return [(kp[f, : cnt[f]], desc[f, : cnt[f]]) for f in range(n)]
            return [(kp[f, : cnt[f]].copy(), desc[f, : cnt[f]].copy()) for f in range(n)]

    def sift_geometry(self):
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.rgbdfe_sift_geometry(self._ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(octave_min=a.value, octave_num=b.value, levels=c.value, dog_levels=d.value)

    def sift_debug_plane(self, octave: int, level: int) -> np.ndarray:
        """A Gaussian plane [h, w] of the latest sift_detect frame (tests)."""
        buf = np.zeros(1 << 25, np.float32)
        w, h = C.c_int32(), C.c_int32()
        self._check(self._L.rgbdfe_sift_debug_plane(self._ctx, octave, level, buf.ctypes.data, buf.size, C.byref(w),
                                                    C.byref(h)))
        return buf[: w.value * h.value].reshape(h.value, w.value).copy()

    def sift_debug_candidates(self, octave: int, dog_level: int) -> np.ndarray:
        """The keypoint candidates [n, 6] = (x, y, sign, dx, dy, ds) of one (octave, dog level), list order (tests)."""
        buf = np.zeros((1 << 20, 6), np.float32)
        n = C.c_int32()
        self._check(self._L.rgbdfe_sift_debug_candidates(self._ctx, octave, dog_level, buf.ctypes.data, buf.shape[0],
                                                         C.byref(n)))
        return buf[: n.value].copy()

    def upload_nodes(self, node_ids, descs, xyz1s):
        """rgbdfe_upload_nodes: upload_node for a list of nodes in one call (one wait, pinned staging)."""
        n = len(node_ids)
        d = [np.ascontiguousarray(x, np.uint8) for x in descs]
        p = [np.ascontiguousarray(x, np.float32) for x in xyz1s]
        for a, b in zip(d, p):
            if a.ndim != 2 or a.shape[1] != 32 or b.shape != (a.shape[0], 4):
                raise ValueError("desc must be [n,32] uint8 and xyz1 [n,4] float32")
        ids = np.ascontiguousarray(node_ids, np.int32)
        cnt = np.array([a.shape[0] for a in d], np.int32)
        vp = C.c_void_p * max(n, 1)
        pd = vp(*[a.ctypes.data for a in d])
        pp = vp(*[b.ctypes.data for b in p])
        self._check(self._L.rgbdfe_upload_nodes(self._ctx, n, ids.ctypes.data, C.cast(pd, C.c_void_p), C.cast(pp, C.c_void_p),
                                                cnt.ctypes.data))

    def upload_sift_node(self, node_id: int, desc128: np.ndarray, xyz1: np.ndarray):
        desc128 = np.ascontiguousarray(desc128, np.float32)
        xyz1 = np.ascontiguousarray(xyz1, np.float32)
        n = desc128.shape[0]
        if desc128.ndim != 2 or desc128.shape[1] != 128 or xyz1.shape != (n, 4):
            raise ValueError("desc128 must be [n,128] float32 and xyz1 [n,4] float32")
        self._check(self._L.rgbdfe_upload_sift_node(self._ctx, node_id, desc128.ctypes.data,
                                                    xyz1.ctypes.data, n))

    def match_sift_pair_list(self, query_ids, train_ids):
        """Batched matchNodePair for SIFT nodes; returns (records, all_dist [n, 320] float32)."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        out = np.zeros(q.shape[0], RESULT_DTYPE)
        dist = np.zeros((q.shape[0], _lib.RGBDFE_MAX_MATCHES), np.float32)
        self._check(self._L.rgbdfe_match_sift_pair_list(self._ctx, q.ctypes.data, t.ctypes.data,
                                                        q.shape[0], out.ctypes.data, dist.ctypes.data))
        return out, dist

    def upload_float_node(self, node_id: int, desc: np.ndarray, xyz1: np.ndarray):
        """Node::feature_descriptors_ (N x dim CV_32F) for the FLANN branch (node.cpp:610-667)."""
        desc = np.ascontiguousarray(desc, np.float32)
        xyz1 = np.ascontiguousarray(xyz1, np.float32)
        n = desc.shape[0]
        if desc.ndim != 2 or xyz1.shape != (n, 4):
            raise ValueError("desc must be [n, dim] float32 and xyz1 [n, 4] float32")
        self._check(self._L.rgbdfe_upload_float_node(self._ctx, node_id, desc.ctypes.data, desc.shape[1],
                                                     xyz1.ctypes.data, n))

    def match_flann_pair_list(self, query_ids, train_ids, nn_distance_ratio: float = 0.95):
        """Batched matchNodePair on the FLANN branch with exact neighbours; returns (records, ratios [n, 320])."""
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        out = np.zeros(q.shape[0], RESULT_DTYPE)
        dist = np.zeros((q.shape[0], _lib.RGBDFE_MAX_MATCHES), np.float32)
        self._check(self._L.rgbdfe_match_flann_pair_list(self._ctx, q.ctypes.data, t.ctypes.data, q.shape[0],
                                                         nn_distance_ratio, out.ctypes.data, dist.ctypes.data))
        return out, dist

    def submit_sift_pair_list(self, query_ids, train_ids, d_out_ptr: int, d_dist_ptr: Optional[int] = None) -> int:
        q = np.ascontiguousarray(query_ids, np.int32)
        t = np.ascontiguousarray(train_ids, np.int32)
        ticket = C.c_int64(0)
        self._check(self._L.rgbdfe_submit_sift_pair_list(self._ctx, q.ctypes.data, t.ctypes.data,
                                                         q.shape[0], d_out_ptr, d_dist_ptr,
                                                         C.byref(ticket)))
        return ticket.value

    def sift_match_nodes(self, query_id: int, train_id: int):
        """SiftGPUWrapper::match twin (sift_gpu_wrapper.cpp:169-227): (queryIdx, trainIdx, L2)."""
        n = max(self.node_count(query_id), 1)
        mq = np.empty(n, np.int32)
        mt = np.empty(n, np.int32)
        md = np.empty(n, np.float32)
        k = C.c_int32(0)
        self._check(self._L.rgbdfe_sift_match_nodes(self._ctx, query_id, train_id, mq.ctypes.data,
                                                    mt.ctypes.data, md.ctypes.data, C.byref(k)))
        return mq[: k.value].copy(), mt[: k.value].copy(), md[: k.value].copy()

    def place_recognition(self, query_id: int, candidate_ids, k_neighbours: int = 2, max_hd: int = 128,
                          max_out: Optional[int] = None):
        """Descriptor-vote ranking of loop-closure candidates (loop_closing.cpp:190-277 with exact neighbours).
        Returns (ids, scores) in descending score order."""
        c = np.ascontiguousarray(candidate_ids, np.int32)
        max_out = len(c) if max_out is None else max_out
        ids = np.zeros(max(max_out, 1), np.int32)
        sc = np.zeros(max(max_out, 1), np.float32)
        n = C.c_int32(0)
        self._check(self._L.rgbdfe_place_recognition(self._ctx, query_id, c.ctypes.data, len(c), k_neighbours, max_hd,
                                                     max_out, ids.ctypes.data, sc.ctypes.data, C.byref(n)))
        return ids[: n.value].copy(), sc[: n.value].copy()

    def place_recognition_batch(self, query_ids, candidate_lists, k_neighbours: int = 2, max_hd: int = 128,
                                max_out: int = 16):
        """Many queries in one launch.  candidate_lists[s] = the candidate ids of query_ids[s].
        Returns a list of (ids, scores) per query."""
        qs = np.ascontiguousarray(query_ids, np.int32)
        offs = np.zeros(len(qs) + 1, np.int32)
        offs[1:] = np.cumsum([len(c) for c in candidate_lists])
        cands = np.ascontiguousarray(np.concatenate([np.asarray(c, np.int32) for c in candidate_lists])
                                     if len(candidate_lists) else np.zeros(0, np.int32), np.int32)
        ids = np.zeros((max(len(qs), 1), max(max_out, 1)), np.int32)
        sc = np.zeros((max(len(qs), 1), max(max_out, 1)), np.float32)
        cnt = np.zeros(max(len(qs), 1), np.int32)
        self._check(self._L.rgbdfe_place_recognition_batch(self._ctx, qs.ctypes.data, len(qs), offs.ctypes.data,
                                                           cands.ctypes.data, k_neighbours, max_hd, max_out,
                                                           ids.ctypes.data, sc.ctypes.data, cnt.ctypes.data))
        return [(ids[s, : cnt[s]].copy(), sc[s, : cnt[s]].copy()) for s in range(len(qs))]

    # -- pieces -----------------------------------------------------------------------
    def hamming_nn_nodes(self, query_id: int, train_id: int):
        n = self.node_count(query_id)
        if n < 0:
            raise RgbdfeError("unknown node id")
        hd = np.empty(n, np.int32)
        idx = np.empty(n, np.int32)
        self._check(self._L.rgbdfe_hamming_nn_nodes(self._ctx, query_id, train_id,
                                                    hd.ctypes.data, idx.ctypes.data))
        return hd, idx

    def bruteForceSearchORB_batch(self, qdesc: np.ndarray, tdesc: np.ndarray):
        """Every row of qdesc through bruteForceSearchORB(row, tdesc, len(tdesc)) (features.h:14)."""
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        tdesc = np.ascontiguousarray(tdesc, np.uint8)
        hd = np.empty(qdesc.shape[0], np.int32)
        idx = np.empty(qdesc.shape[0], np.int32)
        self._check(self._L.rgbdfe_hamming_nn_host(self._ctx, qdesc.ctypes.data, qdesc.shape[0],
                                                   tdesc.ctypes.data, tdesc.shape[0],
                                                   hd.ctypes.data, idx.ctypes.data))
        return hd, idx

    def project_to_3d(self, kp_xy, depth, fx, fy, cx, cy, depth_scaling=1.0, max_keypoints=1000):
        """removeDepthless + projectTo3D (node.cpp:67-97, 900-965)."""
        kp_xy = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
        depth = np.ascontiguousarray(depth, np.float32)
        n = kp_xy.shape[0]
        kept = np.empty(max(n, 1), np.int32)
        xyz1 = np.empty((max(n, 1), 4), np.float32)
        n_out = C.c_int32(0)
        self._check(self._L.rgbdfe_project_to_3d(
            self._ctx, kp_xy.ctypes.data, n, depth.ctypes.data, depth.shape[0], depth.shape[1],
            fx, fy, cx, cy, depth_scaling, max_keypoints, kept.ctypes.data, xyz1.ctypes.data,
            C.byref(n_out)))
        return kept[: n_out.value].copy(), xyz1[: n_out.value].copy()

    def project_to_3d_cloud(self, kp_xy, cloud, maximum_depth, max_keypoints=1000):
        """Node::projectTo3D, point-cloud overload (node.cpp:855-898).  cloud: [rows, cols, 4] float32."""
        kp_xy = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
        cloud = np.ascontiguousarray(cloud, np.float32)
        n = kp_xy.shape[0]
        kept = np.empty(max(n, 1), np.int32)
        xyz1 = np.empty((max(n, 1), 4), np.float32)
        n_out = C.c_int32(0)
        self._check(self._L.rgbdfe_project_to_3d_cloud(self._ctx, kp_xy.ctypes.data, n, cloud.ctypes.data, cloud.shape[0],
                                                       cloud.shape[1], maximum_depth, max_keypoints, kept.ctypes.data,
                                                       xyz1.ctypes.data, C.byref(n_out)))
        return kept[: n_out.value].copy(), xyz1[: n_out.value].copy()

    def detect_describe_cloud(self, gray, mask, cloud, maximum_depth):
        """The feature path of the Node constructor that is given the organised cloud (node.cpp:252-369)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        cloud = np.ascontiguousarray(cloud, np.float32)
        cap = int(1.5 * getattr(self, "_max_keypoints", 600)) + 64
        kps = np.zeros(cap, _lib.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        xyz = np.zeros((cap, 4), np.float32)
        n = C.c_int32(0)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._check(self._L.rgbdfe_detect_describe_cloud(self._ctx, gray.ctypes.data, None if m is None else m.ctypes.data,
                                                         cloud.ctypes.data, gray.shape[0], gray.shape[1], maximum_depth,
                                                         kps.ctypes.data, desc.ctypes.data, xyz.ctypes.data, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy(), xyz[: n.value].copy()

    def sift_node_features(self, kp_xy, desc, depth, fx, fy, cx, cy, depth_scaling=1.0, max_keypoints=1000,
                           use_root_sift=True, kp_size=None):
        """projectTo3DSiftGPU (node.cpp:695-769) + squareroot_descriptor_space (node.cpp:1557-1571):
        returns (kept_idx, xyz1, siftgpu_descriptors, feature_descriptors).  kp_size (cv::KeyPoint::size per keypoint)
        selects the use_feature_min_depth variant (node.cpp:727-731)."""
        kp_xy = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
        desc = np.ascontiguousarray(desc, np.float32)
        depth = np.ascontiguousarray(depth, np.float32)
        n = kp_xy.shape[0]
        if desc.shape != (n, 128):
            raise ValueError("desc must be n_kp x 128 float32")
        cap = max(min(n, max_keypoints), 1)
        kept = np.empty(cap, np.int32)
        xyz1 = np.empty((cap, 4), np.float32)
        raw = np.empty((cap, 128), np.float32)
        feat = np.empty((cap, 128), np.float32)
        n_out = C.c_int32(0)
        if kp_size is not None:
            kp_size = np.ascontiguousarray(kp_size, np.float32).reshape(-1)
            if kp_size.shape[0] != n:
                raise ValueError("kp_size must hold one size per keypoint")
            self._check(self._L.rgbdfe_sift_node_features_min_depth(
                self._ctx, kp_xy.ctypes.data, kp_size.ctypes.data, n, desc.ctypes.data, depth.ctypes.data, depth.shape[0],
                depth.shape[1], fx, fy, cx, cy, depth_scaling, max_keypoints, int(bool(use_root_sift)),
                kept.ctypes.data, xyz1.ctypes.data, raw.ctypes.data, feat.ctypes.data, C.byref(n_out)))
        else:
            self._check(self._L.rgbdfe_sift_node_features(
                self._ctx, kp_xy.ctypes.data, n, desc.ctypes.data, depth.ctypes.data, depth.shape[0],
                depth.shape[1], fx, fy, cx, cy, depth_scaling, max_keypoints, int(bool(use_root_sift)),
                kept.ctypes.data, xyz1.ctypes.data, raw.ctypes.data, feat.ctypes.data, C.byref(n_out)))
        k = n_out.value
        return kept[:k].copy(), xyz1[:k].copy(), raw[:k].copy(), feat[:k].copy()

    # -- frame-level data either side of the pair path ------
    def depth_to_mono8(self, depth):
        """depthToCV8UC1 (misc.cpp:414-430).  float32 metres -> mono8; uint16 millimetres -> (mono8, metres)."""
        depth = np.ascontiguousarray(depth)
        rows, cols = depth.shape
        mono8 = np.empty((rows, cols), np.uint8)
        if depth.dtype == np.uint16:
            dm = np.empty((rows, cols), np.float32)
            self._check(self._L.rgbdfe_depth_to_mono8(self._ctx, depth.ctypes.data, 1, rows, cols,
                                                      mono8.ctypes.data, dm.ctypes.data))
            return mono8, dm
        depth = np.ascontiguousarray(depth, np.float32)
        self._check(self._L.rgbdfe_depth_to_mono8(self._ctx, depth.ctypes.data, 0, rows, cols,
                                                  mono8.ctypes.data, None))
        return mono8

    def upload_node_cloud(self, node_id, depth, fx, fy, cx, cy, rgb=None, encoding_bgr=False, depth_scaling=1.0,
                          min_depth=0.1, cloud_skip=2, return_cloud=False):
        """createXYZRGBPointCloud (misc.cpp:467-556): builds and keeps the node's structured cloud."""
        depth = np.ascontiguousarray(depth, np.float32)
        rows, cols = depth.shape
        ch = 0
        if rgb is not None:
            rgb = np.ascontiguousarray(rgb, np.uint8)
            ch = 1 if rgb.ndim == 2 else rgb.shape[2]
            if rgb.shape[:2] != (rows, cols):
                raise ValueError("rgb and depth must have the same size")
        out = np.empty((rows // cloud_skip, cols // cloud_skip, 4), np.float32) if return_cloud else None
        self._check(self._L.rgbdfe_upload_node_cloud(
            self._ctx, int(node_id), depth.ctypes.data, rows, cols, rgb.ctypes.data if rgb is not None else None,
            ch, int(bool(encoding_bgr)), fx, fy, cx, cy, depth_scaling, min_depth, int(cloud_skip),
            out.ctypes.data if out is not None else None))
        return out

    def release_node_cloud(self, node_id):
        self._check(self._L.rgbdfe_release_node_cloud(self._ctx, int(node_id)))

    def observation_likelihood(self, new_ids, old_ids, transforms, emm_skip_step=8):
        """observationLikelihood (misc.cpp:814-969) for a batch of directed edges; transforms: n x 4 x 4
        (row-major numpy matrices, new -> old).  Returns an n x 4 uint32 array (inliers, outliers, occluded, all)."""
        new_ids = np.ascontiguousarray(new_ids, np.int32)
        old_ids = np.ascontiguousarray(old_ids, np.int32)
        T = np.ascontiguousarray(np.asarray(transforms, np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))  # column-major
        n = len(new_ids)
        out = np.zeros((max(n, 1), 4), np.uint32)
        self._check(self._L.rgbdfe_observation_likelihood(self._ctx, n, new_ids.ctypes.data, old_ids.ctypes.data,
                                                          T.ctypes.data, int(emm_skip_step), out.ctypes.data))
        return out[:n]

    def pairwise_observation_likelihood(self, results, observability_threshold, emm_skip_step=8):
        """pairwiseObservationLikelihood (node.cpp:1520-1554) + observation_criterion_met (misc.cpp:1136-1148)
        for match results with an edge: returns (counts n x 4 summed over both directions, criterion_met)."""
        results = np.asarray(results)
        n = len(results)
        T = np.array([np.array(r["trafo"], np.float32).reshape(4, 4).T for r in results], np.float32).reshape(-1, 4, 4)
        Tinv = np.array([np.linalg.inv(t.astype(np.float64)).astype(np.float32) for t in T], np.float32).reshape(-1, 4, 4)
        newer, older = results["id2"].astype(np.int32), results["id1"].astype(np.int32)
        a = self.observation_likelihood(newer, older, T, emm_skip_step)
        b = self.observation_likelihood(older, newer, Tinv, emm_skip_step)
        c = a + b
        met = np.zeros(n, bool)
        q = C.c_double(0)
        for i in range(n):
            met[i] = bool(self._L.rgbdfe_observation_criterion_met(
                int(c[i, 0]), int(c[i, 1]), int(c[i, 2] + c[i, 0] + c[i, 1]), observability_threshold, C.byref(q)))
        return c, met

    def set_latency_mode(self, max_pairs=(1 << 31) - 1, chunk_iterations=0):
        """Batches of at most max_pairs pairs spread each pair's RANSAC iterations over several waves (record /
        replay, identical results; the default for every batch size); max_pairs = 0 forces one wave per pair,
        chunk_iterations = 0 is automatic."""
        self._check(self._L.rgbdfe_set_latency_mode(self._ctx, int(max_pairs), int(chunk_iterations)))

    # -- measurement ----------------------------------------------------------------------
    def set_profiling(self, enable: bool):
        self._check(self._L.rgbdfe_set_profiling(self._ctx, int(enable)))

    def reset_kernel_time(self):
        self._check(self._L.rgbdfe_reset_kernel_time(self._ctx))

    def set_graph_capture(self, enable: bool):
        """Cached hipGraphs for ORB pair batches (rgbdfe_set_graph_capture): off by default -- an open capture makes
        device-wide synchronisations on other threads of the process fail; for callers that own every HIP-calling thread."""
        self._check(self._L.rgbdfe_set_graph_capture(self._ctx, int(bool(enable))))

    def graph_stats(self):
        """The hipGraph cache of the ORB pair path (rgbdfe_graph_stats)."""
        v = (C.c_int64 * 8)()
        self._check(self._L.rgbdfe_graph_stats(self._ctx, v, 8))
        names = ("captures", "launches", "misses", "plain_batches", "capture_failures", "launch_failures", "cached", "enabled")
        return dict(zip(names, (int(x) for x in v)))

    def kernel_time(self, which: int):
        ms, n, p = C.c_double(0), C.c_int64(0), C.c_int64(0)
        self._check(self._L.rgbdfe_get_kernel_time(self._ctx, which, C.byref(ms), C.byref(n),
                                                   C.byref(p)))
        return ms.value, n.value, p.value


class Node:
    """The slice of the reference's Node (src/node.h) that the pair path touches."""

    def __init__(self, frontend: FrontEnd, node_id: int, feature_descriptors: np.ndarray,
                 feature_locations_3d: np.ndarray):
        self.frontend = frontend
        self.id_ = node_id
        self.feature_descriptors_ = np.ascontiguousarray(feature_descriptors, np.uint8)
        self.feature_locations_3d_ = np.ascontiguousarray(feature_locations_3d, np.float32)
        self.matchable_ = True
        frontend.upload_node(node_id, self.feature_descriptors_, self.feature_locations_3d_)

    def matchNodePair(self, older_node: "Node") -> MatchingResult:
        """src/node.cpp:1305-1429"""
        rec = self.frontend.match_node_pairs(self.id_, [older_node.id_])[0]
        return record_to_matching_result(rec)

    def featureMatching(self, other: "Node") -> List[DMatch]:
        """ORB branch of src/node.cpp:535-690 (matches sorted by (hd, queryIdx))."""
        rec = self.frontend.match_node_pairs(self.id_, [other.id_])[0]
        return record_to_matching_result(rec).all_matches

    def clearFeatureInformation(self):
        """src/node.cpp:1431-1443"""
        self.frontend.release_node(self.id_)
        self.matchable_ = False


class GraphManager:
    """Only the fan-out of GraphManager::nodeComparisons (src/graph_manager.cpp:531-583)."""

    def __init__(self, frontend: FrontEnd):
        self.frontend = frontend

    def nodeComparisons(self, new_node: Node, nodes_to_comp: Sequence[Node]) -> List[MatchingResult]:
        recs = self.frontend.match_node_pairs(new_node.id_, [n.id_ for n in nodes_to_comp])
        return [record_to_matching_result(r) for r in recs]
