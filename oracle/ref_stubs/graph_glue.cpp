// oracle/ref_stubs/graph_glue.cpp -- TEST INFRASTRUCTURE ONLY: C entry point over the reference's
// GraphManager::getPotentialEdgeTargetsWithDijkstra compiled from its source (see graph_prelude.h).  rand() is libc's:
// the caller seeds it with srand() and feeds the same stream to the product code.
extern "C" int ref_potential_edge_targets(int n_nodes, const int* node_ids, const int* vertex_ids, const int* matchable,
                                          int n_keyframes, const int* keyframes, int n_edges, const int* edge_a,
                                          const int* edge_b, int sequential_targets, int geodesic_targets,
                                          int sampled_targets, int geodesic_depth, int predecessor_id,
                                          int include_predecessor, unsigned srand_seed, int* ids_out, int capacity) {
  GraphManager gm;
  g2o::SparseOptimizer opt;
  gm.optimizer_ = &opt;
  std::vector<Node> nodes((size_t)n_nodes);
  std::vector<g2o::VertexSE3> verts((size_t)n_nodes);
  for (int i = 0; i < n_nodes; ++i) {
    nodes[i].id_ = node_ids[i];
    nodes[i].vertex_id_ = vertex_ids[i];
    nodes[i].matchable_ = matchable[i] != 0;
    gm.graph_[node_ids[i]] = &nodes[i];
    verts[i].id_ = vertex_ids[i];
    opt.v_[vertex_ids[i]] = &verts[i];
    gm.camera_vertices.insert(&verts[i]);
  }
  for (int e = 0; e < n_edges; ++e) {
    g2o::HyperGraph::Vertex* a = opt.vertex(gm.graph_[edge_a[e]]->vertex_id_);
    g2o::HyperGraph::Vertex* b = opt.vertex(gm.graph_[edge_b[e]]->vertex_id_);
    a->neighbours.insert(b);
    b->neighbours.insert(a);
  }
  for (int k = 0; k < n_keyframes; ++k) gm.keyframe_ids_.push_back(keyframes[k]);
  ParameterServer::instance()->geodesic_depth = geodesic_depth;
  srand(srand_seed);
  QList<int> r = gm.getPotentialEdgeTargetsWithDijkstra(nullptr, sequential_targets, geodesic_targets, sampled_targets,
                                                        predecessor_id, include_predecessor != 0);
  for (int i = 0; i < r.size() && i < capacity; ++i) ids_out[i] = r[i];
  return r.size();
}
