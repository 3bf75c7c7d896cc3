"""Candidate selection for loop closure: the host-side mirror of
``GraphManager::getPotentialEdgeTargetsWithDijkstra`` (graph_manager.cpp:204-324) over the C ABI
(``rgbdfe_pose_graph_*`` / ``rgbdfe_potential_edge_targets`` in include/rgbdfe.h).  Its output is the list of
earlier nodes a new node is compared with, i.e. the ``tids`` of ``FrontEnd.match_node_pairs``."""
import ctypes as C

import numpy as np

from . import _lib


class PoseGraph:
    """What the reference's candidate selection reads from GraphManager: graph_ (id_, vertex_id_, matchable_),
    camera_vertices, the optimizer's edges and keyframe_ids_."""

    def __init__(self):
        self._L = _lib.load()
        self._g = self._L.rgbdfe_pose_graph_create()
        if not self._g:
            raise _lib.RgbdfeError("rgbdfe_pose_graph_create failed")

    def close(self):
        if self._g:
            self._L.rgbdfe_pose_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise _lib.RgbdfeError(self._L.rgbdfe_status_string(rc).decode())

    def add_node(self, node_id, vertex_id=None, matchable=True, keyframe=False):
        """A Node entering graph_ (GraphManager::addNode, graph_manager.cpp:681; firstNode :326)."""
        self._check(self._L.rgbdfe_pose_graph_add_node(self._g, int(node_id), int(node_id if vertex_id is None else vertex_id),
                                                       int(bool(matchable)), int(bool(keyframe))))

    def add_edge(self, id1, id2):
        """An edge entering the optimizer (GraphManager::addEdgeToG2O, graph_manager.cpp:811)."""
        self._check(self._L.rgbdfe_pose_graph_add_edge(self._g, int(id1), int(id2)))

    def set_matchable(self, node_id, matchable):
        self._check(self._L.rgbdfe_pose_graph_set_matchable(self._g, int(node_id), int(bool(matchable))))

    def potential_edge_targets(self, sequential_targets, geodesic_targets, sampled_targets, geodesic_depth=3,
                               predecessor_id=-1, include_predecessor=False, rand=None, seed=0):
        """ids to compare the next node with.  rand: a callable returning the next rand() value (e.g. libc's rand for
        the reference's stream); None = the library's counter-based generator seeded with `seed`."""
        cap = int(sequential_targets + geodesic_targets + sampled_targets + 1)
        out = np.zeros(max(cap, 1), np.int32)
        n = C.c_int32(0)
        cb = self._L.rgbdfe_rand_fn((lambda _state: int(rand())) if rand is not None else 0)
        self._check(self._L.rgbdfe_potential_edge_targets(
            self._g, int(sequential_targets), int(geodesic_targets), int(sampled_targets), int(geodesic_depth),
            int(predecessor_id), int(bool(include_predecessor)), cb, None, int(seed) & 0xFFFFFFFF,
            out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value].copy()
