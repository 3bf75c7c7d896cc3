// Candidate selection for loop closure: which earlier nodes a new node is compared with
// (GraphManager::getPotentialEdgeTargetsWithDijkstra, graph_manager.cpp:204-324) -- the step in front of the pair
// path; its output is the `tids` list of rgbdfe_match_node_pairs.  Host code only (no device work): the pose-graph
// topology it reads is a few integers per node.
//
// What the reference reads, and what stands for it here:
//   graph_ (std::map<int, Node*>: id_, vertex_id_, matchable_)      PoseGraph::nodes
//   camera_vertices (g2o::HyperGraph::VertexSet)                   PoseGraph::camera_vertices (vertex ids)
//   optimizer_ edges between camera vertices                       PoseGraph::adjacency
//   keyframe_ids_ (QList<int>)                                     PoseGraph::keyframes
//   g2o::HyperDijkstra::shortestPaths(v, UniformCostFunction, geodesic_depth)   bounded_neighbourhood()
//   rand()                                                         the caller's generator (NULL: a counter-based one)
#include <cstdint>
#include <deque>
#include <limits>
#include <map>
#include <new>
#include <queue>
#include <set>
#include <utility>
#include <vector>

#include "rgbdfe.h"

struct rgbdfe_pose_graph {
  struct NodeInfo {
    int32_t vertex_id;
    bool matchable;
  };
  std::map<int32_t, NodeInfo> nodes;                 // node id -> node (graph_)
  std::set<int32_t> camera_vertices;                 // vertex ids
  std::map<int32_t, std::set<int32_t>> adjacency;    // vertex id -> vertex ids joined by an edge
  std::deque<int32_t> keyframes;
};

namespace {

// g2o::HyperDijkstra::shortestPaths with UniformCostFunction (every edge costs 1) and maxDistance: a vertex is
// relaxed only while its distance stays BELOW maxDistance (g2o/core/hyper_dijkstra.cpp: `zDistance < maxDistance`),
// visited() = the start vertex and every vertex that was relaxed.  (g2o is a third-party dependency that is not part
// of the reference tree: restated.)
std::set<int32_t> bounded_neighbourhood(const rgbdfe_pose_graph& g, int32_t start, double max_distance) {
  std::map<int32_t, double> dist;
  std::set<int32_t> visited;
  typedef std::pair<double, int32_t> Entry;
  std::priority_queue<Entry, std::vector<Entry>, std::greater<Entry>> frontier;
  dist[start] = 0.0;
  frontier.push(Entry(0.0, start));
  while (!frontier.empty()) {
    const Entry e = frontier.top();
    frontier.pop();
    const int32_t u = e.second;
    const double du = dist[u];
    visited.insert(u);
    const auto at = g.adjacency.find(u);
    if (at == g.adjacency.end()) continue;
    for (int32_t z : at->second) {
      const double dz = du + 1.0;
      const auto zt = dist.find(z);
      const double old = zt == dist.end() ? std::numeric_limits<double>::max() : zt->second;
      if (dz < old && dz < max_distance) {
        dist[z] = dz;
        frontier.push(Entry(dz, z));
      }
    }
  }
  return visited;
}

struct CounterRand {  // stand-in for rand() when the caller brings no generator: same integer mixer as the RANSAC sampler
  uint32_t seed, k = 0;
  static uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
  }
  int next() { return (int)(mix(mix(seed ^ 0x9E3779B9u) + (k++) * 0x85EBCA6Bu) >> 1); }
};

}  // namespace

extern "C" {

rgbdfe_pose_graph* rgbdfe_pose_graph_create(void) { return new (std::nothrow) rgbdfe_pose_graph(); }

void rgbdfe_pose_graph_destroy(rgbdfe_pose_graph* g) { delete g; }

int rgbdfe_pose_graph_add_node(rgbdfe_pose_graph* g, int32_t node_id, int32_t vertex_id, int32_t matchable,
                               int32_t keyframe) try {
  if (!g || node_id < 0 || vertex_id < 0) return RGBDFE_ERR_INVALID_ARG;
  g->nodes[node_id] = rgbdfe_pose_graph::NodeInfo{vertex_id, matchable != 0};
  g->camera_vertices.insert(vertex_id);
  g->adjacency[vertex_id];
  if (keyframe) g->keyframes.push_back(node_id);
  return RGBDFE_OK;
} catch (const std::bad_alloc&) {
  return RGBDFE_ERR_OUT_OF_MEMORY;
} catch (...) {
  return RGBDFE_ERR_INTERNAL;  // no exception crosses the C ABI
}

int rgbdfe_pose_graph_add_edge(rgbdfe_pose_graph* g, int32_t node_id1, int32_t node_id2) try {
  if (!g) return RGBDFE_ERR_INVALID_ARG;
  const auto a = g->nodes.find(node_id1), b = g->nodes.find(node_id2);
  if (a == g->nodes.end() || b == g->nodes.end() || node_id1 == node_id2) return RGBDFE_ERR_INVALID_ARG;
  g->adjacency[a->second.vertex_id].insert(b->second.vertex_id);
  g->adjacency[b->second.vertex_id].insert(a->second.vertex_id);
  return RGBDFE_OK;
} catch (const std::bad_alloc&) {
  return RGBDFE_ERR_OUT_OF_MEMORY;
} catch (...) {
  return RGBDFE_ERR_INTERNAL;  // no exception crosses the C ABI
}

int rgbdfe_pose_graph_set_matchable(rgbdfe_pose_graph* g, int32_t node_id, int32_t matchable) try {
  if (!g) return RGBDFE_ERR_INVALID_ARG;
  const auto a = g->nodes.find(node_id);
  if (a == g->nodes.end()) return RGBDFE_ERR_INVALID_ARG;
  a->second.matchable = matchable != 0;
  return RGBDFE_OK;
} catch (const std::bad_alloc&) {
  return RGBDFE_ERR_OUT_OF_MEMORY;
} catch (...) {
  return RGBDFE_ERR_INTERNAL;  // no exception crosses the C ABI
}

int rgbdfe_potential_edge_targets(const rgbdfe_pose_graph* g, int32_t sequential_targets, int32_t geodesic_targets,
                                  int32_t sampled_targets, int32_t geodesic_depth, int32_t predecessor_id,
                                  int32_t include_predecessor, rgbdfe_rand_fn rand_fn, void* rand_state, uint32_t seed,
                                  int32_t* ids_out, int32_t capacity, int32_t* n_out) try {
  if (!g || !ids_out || !n_out || capacity < 0) return RGBDFE_ERR_INVALID_ARG;
  CounterRand own{seed};
  auto draw = [&]() { return rand_fn ? rand_fn(rand_state) : own.next(); };
  const int graph_size = (int)g->nodes.size();
  std::deque<int32_t> ids;  // QList<int> ids_to_link_to: sampled ids go to the front, sequential ones to the back
  if (predecessor_id < 0) predecessor_id = graph_size - 1;  // :207

  // fewer previous nodes than targets requested: just use all of them (:212-219)
  if ((int)g->camera_vertices.size() <= sequential_targets + geodesic_targets + sampled_targets ||
      g->camera_vertices.size() <= 1) {
    sequential_targets = sequential_targets + geodesic_targets + sampled_targets;
    geodesic_targets = 0;
    sampled_targets = 0;
    predecessor_id = graph_size - 1;
  }

  if (sequential_targets > 0)  // :221-227
    for (int i = 1; i < sequential_targets + 1 && predecessor_id - i >= 0; i++) ids.push_back(predecessor_id - i);

  if (geodesic_targets > 0) {  // :229-295
    const auto pred = g->nodes.find(predecessor_id);
    if (pred == g->nodes.end()) return RGBDFE_ERR_INVALID_ARG;
    const std::set<int32_t> vs = bounded_neighbourhood(*g, pred->second.vertex_id, (double)geodesic_depth);
    std::map<int32_t, int32_t> vertex_id_to_node_id;
    for (const auto& kv : g->nodes) vertex_id_to_node_id[kv.second.vertex_id] = kv.first;
    // geodesic neighbours except the sequential ones, weighted by their distance in time (:246-270)
    std::map<int32_t, int32_t> neighbour_weights;
    int sum_of_weights = 0;
    for (int32_t vid : vs) {
      const auto it = vertex_id_to_node_id.find(vid);
      const int32_t id = it == vertex_id_to_node_id.end() ? 0 : it->second;  // the reference falls back to id 0 (:250-265)
      const auto nd = g->nodes.find(id);
      if (nd == g->nodes.end() || !nd->second.matchable) continue;
      if (id < predecessor_id - sequential_targets || (id > predecessor_id && id <= graph_size - 1)) {
        const int weight = id > predecessor_id ? id - predecessor_id : predecessor_id - id;
        neighbour_weights[id] = weight;
        sum_of_weights += weight;
      }
    }
    // weighted sampling without replacement (:273-294)
    while ((int)ids.size() < sequential_targets + geodesic_targets && !neighbour_weights.empty()) {
      if (sum_of_weights <= 0) return RGBDFE_ERR_INVALID_ARG;  // rand() % 0 in the reference
      const int random_pick = draw() % sum_of_weights;
      int weight_so_far = 0;
      for (auto mit = neighbour_weights.begin(); mit != neighbour_weights.end(); ++mit) {
        weight_so_far += mit->second;
        if (weight_so_far > random_pick) {
          ids.push_front(mit->first);
          sum_of_weights -= mit->second;
          neighbour_weights.erase(mit);
          break;
        }
      }
    }
  }

  if (sampled_targets > 0) {  // :297-317: uniform sampling among the keyframes not yet chosen
    std::vector<int32_t> pool;
    pool.reserve(g->nodes.size());
    for (int32_t kf : g->keyframes) {
      bool chosen = false;
      for (int32_t v : ids) chosen |= (v == kf);
      const auto nd = g->nodes.find(kf);
      if (!chosen && nd != g->nodes.end() && nd->second.matchable) pool.push_back(kf);
    }
    while ((int)ids.size() < geodesic_targets + sampled_targets + sequential_targets && !pool.empty()) {
      const size_t k = (size_t)draw() % pool.size();
      const int32_t sampled = pool[k];
      pool[k] = pool.back();
      pool.pop_back();
      ids.push_front(sampled);
    }
  }

  if (include_predecessor) ids.push_back(predecessor_id);  // :319-322
  *n_out = (int32_t)ids.size();
  if ((int32_t)ids.size() > capacity) return RGBDFE_ERR_CAPACITY;
  for (size_t i = 0; i < ids.size(); ++i) ids_out[i] = ids[i];
  return RGBDFE_OK;
} catch (const std::bad_alloc&) {
  return RGBDFE_ERR_OUT_OF_MEMORY;
} catch (...) {
  return RGBDFE_ERR_INTERNAL;  // no exception crosses the C ABI
}

}  // extern "C"
