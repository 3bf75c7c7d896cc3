// oracle/ref_stubs/graph_prelude.h -- TEST INFRASTRUCTURE ONLY.
// Stand-ins for Qt, g2o, ROS and the parameter server, just enough for the reference's candidate selection
//   GraphManager::getPotentialEdgeTargetsWithDijkstra     src/graph_manager.cpp:204-324
// to compile FROM WHERE IT LIES in /root/reference.  The first-party logic (the three target classes, the weights,
// the sampling without replacement, the order of the returned list) is then the reference's own code; the g2o
// shortest-path search behind HyperDijkstra is third-party and restated here (uniform edge cost, vertices relaxed
// while their distance stays below maxDistance) -- that part stays "parity unpinned".
#ifndef REF_STUB_GRAPH_PRELUDE_H
#define REF_STUB_GRAPH_PRELUDE_H
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <exception>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#define ROS_INFO(...)
#define ROS_WARN(...)
#define ROS_ERROR(...)
#define ROS_DEBUG(...)
#define ROS_ERROR_COND(c, ...)

template <class T>
class QList {  // the slice of QList<int> the function uses
 public:
  typedef typename std::deque<T>::iterator iterator;
  void push_back(const T& v) { d_.push_back(v); }
  void push_front(const T& v) { d_.push_front(v); }
  T& back() { return d_.back(); }
  T& front() { return d_.front(); }
  int size() const { return (int)d_.size(); }
  int contains(const T& v) const { int c = 0; for (const T& x : d_) c += (x == v); return c; }
  iterator begin() { return d_.begin(); }
  iterator end() { return d_.end(); }
  const T& operator[](int i) const { return d_[(size_t)i]; }
 private:
  std::deque<T> d_;
};

namespace g2o {
struct HyperGraph {
  struct Edge;
  struct Vertex {
    virtual ~Vertex() {}
    int id() const { return id_; }
    int id_ = 0;
    std::set<Vertex*> neighbours;  // stands for edges(): every edge joins two camera vertices
  };
  typedef std::set<Vertex*> VertexSet;
};
struct VertexSE3 : HyperGraph::Vertex {};
struct UniformCostFunction {};
struct SparseOptimizer {
  typedef std::map<int, HyperGraph::Vertex*> VertexIDMap;
  HyperGraph::Vertex* vertex(int id) { auto it = v_.find(id); return it == v_.end() ? nullptr : it->second; }
  VertexIDMap& vertices() { return v_; }
  VertexIDMap v_;
};
// g2o/core/hyper_dijkstra.cpp, shortestPaths(v, cost, maxDistance) with UniformCostFunction (restated, see above)
struct HyperDijkstra {
  explicit HyperDijkstra(SparseOptimizer*) {}
  void shortestPaths(HyperGraph::Vertex* v, UniformCostFunction*, double maxDistance) {
    std::map<HyperGraph::Vertex*, double> dist;
    typedef std::pair<double, HyperGraph::Vertex*> Entry;
    std::priority_queue<Entry, std::vector<Entry>, std::greater<Entry>> frontier;
    dist[v] = 0.0;
    frontier.push(Entry(0.0, v));
    while (!frontier.empty()) {
      HyperGraph::Vertex* u = frontier.top().second;
      frontier.pop();
      const double du = dist[u];
      visited_.insert(u);
      for (HyperGraph::Vertex* z : u->neighbours) {
        const double dz = du + 1.0;
        auto zt = dist.find(z);
        const double old = zt == dist.end() ? std::numeric_limits<double>::max() : zt->second;
        if (dz < old && dz < maxDistance) { dist[z] = dz; frontier.push(Entry(dz, z)); }
      }
    }
  }
  HyperGraph::VertexSet& visited() { return visited_; }
  HyperGraph::VertexSet visited_;
};
}  // namespace g2o

struct ParameterServer {
  static ParameterServer* instance() { static ParameterServer p; return &p; }
  template <class T> T get(const std::string&) { return (T)geodesic_depth; }
  int geodesic_depth = 3;
};

struct Node {
  int id_ = 0, vertex_id_ = 0;
  bool matchable_ = true;
};
typedef std::map<int, Node*>::iterator graph_it;

class GraphManager {
 public:
  QList<int> getPotentialEdgeTargetsWithDijkstra(const Node* new_node, int sequential_targets, int geodesic_targets,
                                                 int sampled_targets, int predecessor_id = -1,
                                                 bool include_predecessor = false);
  std::map<int, Node*> graph_;
  g2o::HyperGraph::VertexSet camera_vertices;
  g2o::SparseOptimizer* optimizer_ = nullptr;
  QList<int> keyframe_ids_;
};
#endif
