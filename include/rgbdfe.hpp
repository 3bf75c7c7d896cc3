// rgbdfe.hpp -- header-only C++ host side above the C ABI (include/rgbdfe.h), mirroring the slice of
// rgbdslam_v2's interface that the pair path exposes: same names, argument meaning and error behaviour
// as src/node.h, src/matching_result.h, src/edge.h and GraphManager::nodeComparisons' fan-out
// (src/graph_manager.cpp:541-548), without the OpenCV / Eigen / Qt / ROS types.
//
//   reference                                   here
//   cv::DMatch                                  rgbdslam::DMatch {queryIdx, trainIdx, distance}
//   Eigen::Matrix4f (column-major)              std::array<float,16> (column-major)
//   LoadedEdge3D / MatchingResult               rgbdslam::LoadedEdge3D / rgbdslam::MatchingResult
//   Node::matchNodePair(const Node*)            rgbdslam::Node::matchNodePair(const Node*)
//   QtConcurrent::blockingMapped(nodes, ...)    rgbdslam::GraphManager::nodeComparisons(new_node, nodes)
//   GraphManager::getPotentialEdgeTargetsWithDijkstra   rgbdslam::GraphManager::getPotentialEdgeTargetsWithDijkstra
// No exceptions are thrown on the pair path: like Node::matchNodePair (node.cpp:1424-1426) failures end
// in a MatchingResult whose edge ids are -1.
#ifndef RGBDFE_HPP
#define RGBDFE_HPP

#include <array>
#include <cmath>
#include <utility>
#include <cstdint>
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rgbdfe.h"

namespace rgbdslam {

struct DMatch {  // cv::DMatch
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0.f;
};

struct LoadedEdge3D {  // src/edge.h:24-32
  int id1 = -1, id2 = -1;
  std::array<double, 16> transform{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};  // Isometry3d, column-major
  double informationScale = 0.0;  // informationMatrix = Identity(6,6) * informationScale (node.cpp:1335)
};

struct MatchingResult {  // src/matching_result.h:24-46
  std::vector<DMatch> inlier_matches, all_matches;
  LoadedEdge3D edge;
  float rmse = 0.0f;
  std::array<float, 16> ransac_trafo{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
  std::array<float, 16> final_trafo{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
  unsigned int inlier_points = 0, outlier_points = 0, occluded_points = 0, all_points = 0;  // environment measurement model
};

// float_dist: the DMatch.distance values of a SIFT / FLANN pair (float L2, rgbdfe_match_sift_pair_list's out_dist), or null
inline MatchingResult toMatchingResult(const rgbdfe_match_result& r, const float* float_dist = nullptr) {
  MatchingResult mr;
  mr.all_matches.resize((size_t)r.n_all);
  for (int m = 0; m < r.n_all; ++m) {
    mr.all_matches[m].queryIdx = r.all_q[m];
    mr.all_matches[m].trainIdx = r.all_t[m];
    mr.all_matches[m].distance = float_dist ? float_dist[m] : r.all_hd[m] / 256.0f;  // node.cpp:573 without the random jitter
  }
  for (int m = 0; m < r.n_all; ++m)
    if ((r.inlier_mask[m >> 6] >> (m & 63)) & 1ull) mr.inlier_matches.push_back(mr.all_matches[m]);
  mr.rmse = r.rmse;
  for (int i = 0; i < 16; ++i) mr.ransac_trafo[i] = mr.final_trafo[i] = r.trafo[i];  // node.cpp:1334
  mr.edge.id1 = r.id1;
  mr.edge.id2 = r.id2;
  if (r.id1 >= 0) {
    for (int i = 0; i < 16; ++i) mr.edge.transform[i] = (double)r.trafo[i];  // node.cpp:1339
    mr.edge.informationScale = r.info_scale;
  }
  return mr;
}

// Owns the rgbdfe context (one GPU).  Construction throws (like the reference's fatal start-up errors);
// the per-pair calls never do.
class FrontEnd {
 public:
  explicit FrontEnd(const rgbdfe_config& cfg) {
    rgbdfe_ctx* c = nullptr;
    const int st = rgbdfe_create(&cfg, &c);
    if (st != RGBDFE_OK) throw std::runtime_error(std::string("rgbdfe_create: ") + rgbdfe_status_string(st));
    ctx_.reset(c, rgbdfe_destroy);
  }
  // Several GPUs behind one handle (rgbdfe_create_multi): nodes are replicated, nodeComparisons shards its candidates
  // (candidate k -> device k mod G) -- nothing else in this header changes.
  FrontEnd(const rgbdfe_config& cfg, const std::vector<int32_t>& device_ids) {
    rgbdfe_ctx* c = nullptr;
    const int st = rgbdfe_create_multi(&cfg, device_ids.data(), (int32_t)device_ids.size(), &c);
    if (st != RGBDFE_OK)
      throw std::runtime_error(std::string("rgbdfe_create_multi: ") + rgbdfe_status_string(st) + ": " + rgbdfe_last_error(nullptr));
    ctx_.reset(c, rgbdfe_destroy);
  }
  int deviceCount() const { return rgbdfe_device_count(ctx_.get()); }
  static rgbdfe_config defaultConfig() {
    rgbdfe_config c;
    rgbdfe_default_config(&c);
    return c;
  }
  rgbdfe_ctx* get() const { return ctx_.get(); }

 private:
  std::shared_ptr<rgbdfe_ctx> ctx_;
};

// src/sift_gpu_wrapper.h:49 -- SiftGPUWrapper::detect(image, keypoints, descriptors, mask): SiftGPU's extraction with the
// options the reference's constructor sets (sift_gpu_wrapper.cpp:29-88).  keypoints: pt, size = 12 * scale, angle in
// degrees; descriptors: n x 128 floats, unnormalised.  `mask` does not exist here: the reference ignores it.
class SiftGPUWrapper {
 public:
  explicit SiftGPUWrapper(const FrontEnd& fe, int max_keypoints) : fe_(fe), max_keypoints_(max_keypoints) {}
  void detect(const uint8_t* image, int rows, int cols, std::vector<rgbdfe_keypoint>& keypoints,
              std::vector<float>& descriptors) const {
    if (!keypoints.empty()) {   // "Use Keypoints if provided" (:132-142): descriptors for the caller's keypoints
      descriptors.assign(keypoints.size() * 128, 0.f);
      if (rgbdfe_sift_describe(fe_.get(), image, rows, cols, keypoints.data(), (int32_t)keypoints.size(), descriptors.data()) !=
          RGBDFE_OK) {
        keypoints.clear();
        descriptors.clear();
      }
      return;
    }
    int32_t cap = 2 * max_keypoints_ + 1024, n = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      keypoints.resize((size_t)cap);
      descriptors.resize((size_t)cap * 128);
      const int rc = rgbdfe_sift_detect(fe_.get(), image, nullptr, rows, cols, max_keypoints_, keypoints.data(),
                                        descriptors.data(), cap, &n);
      if (rc == RGBDFE_OK) break;
      if (rc != RGBDFE_ERR_CAPACITY) { n = 0; break; }   // "SIFTGPU->RunSIFT() failed!" (:158): no features
      cap = n;                                            // n = the number of features: once more with room for them
    }
    keypoints.resize((size_t)n);
    descriptors.resize((size_t)n * 128);
  }
 private:
  FrontEnd fe_;
  int max_keypoints_;
};

class Node {  // the slice of src/node.h the pair path touches
 public:
  // feature_descriptors: n x 32 bytes (cv::Mat CV_8U, continuous); feature_locations_3d: n x (x,y,z,1)
  Node(const FrontEnd& fe, int id, const uint8_t* feature_descriptors, const float* feature_locations_3d, int n)
      : id_(id), fe_(fe), n_(n) {
    matchable_ = rgbdfe_upload_node(fe_.get(), id_, feature_descriptors, feature_locations_3d, n) == RGBDFE_OK;
  }
  // The depth-image constructor (src/node.cpp:139-210): detect (grid of threshold-adaptive ORB detectors) ->
  // removeDepthless -> retainBest(max_keypoints) -> cv::ORB::compute -> projectTo3D, then the node's features go to the
  // device.  gray / mask: rows x cols uint8 (mask may be null), depth: rows x cols float metres.  The 2-D keypoints,
  // descriptors and points stay available as in the reference (feature_locations_2d_, feature_descriptors_,
  // feature_locations_3d_).  The detector's per-cell thresholds live in the FrontEnd and carry over from node to node.
  Node(const FrontEnd& fe, int id, const uint8_t* gray, const uint8_t* mask, const float* depth, int rows, int cols,
       double fx, double fy, double cx, double cy, double depth_scaling, int max_keypoints)
      : id_(id), fe_(fe), n_(0) {
    std::vector<rgbdfe_keypoint> kp((size_t)max_keypoints);
    feature_descriptors_.resize((size_t)max_keypoints * 32);
    feature_locations_3d_.resize((size_t)max_keypoints * 4);
    int32_t n = 0;
    if (rgbdfe_detect_describe(fe_.get(), gray, mask, depth, rows, cols, fx, fy, cx, cy, depth_scaling, kp.data(),
                               feature_descriptors_.data(), feature_locations_3d_.data(), &n) != RGBDFE_OK)
      n = 0;
    n_ = n;
    feature_descriptors_.resize((size_t)n * 32);
    feature_locations_3d_.resize((size_t)n * 4);
    feature_locations_2d_.assign(kp.begin(), kp.begin() + n);
    matchable_ = n > 0 && rgbdfe_upload_node(fe_.get(), id_, feature_descriptors_.data(), feature_locations_3d_.data(), n) == RGBDFE_OK;
  }
  // A node of 128-d float descriptors for matcher_type == "SIFTGPU" (Node::siftgpu_descriptors, node.h:172;
  // Node::featureMatching's SIFTGPU branch, node.cpp:553-557): desc128 = n x 128 floats
  struct SiftDescriptors {};
  Node(const FrontEnd& fe, int id, const float* desc128, const float* feature_locations_3d, int n, SiftDescriptors)
      : id_(id), fe_(fe), n_(n), sift_(true) {
    matchable_ = rgbdfe_upload_sift_node(fe_.get(), id_, desc128, feature_locations_3d, n) == RGBDFE_OK;
  }
  // The depth-image constructor with feature_detector_type == "SIFTGPU" (node.cpp:147-152, 195-200): SiftGPUWrapper::detect ->
  // projectTo3DSiftGPU (:695-769: truncating depth lookup, NaN depth drops the keypoint, max_keypoints cut, descriptors
  // re-packed) -> the node's siftgpu_descriptors go to the device for the SIFTGPU matcher branch.
  struct SiftGPU {};
  Node(const FrontEnd& fe, int id, const uint8_t* gray, const float* depth, int rows, int cols, double fx, double fy, double cx,
       double cy, double depth_scaling, int max_keypoints, SiftGPU)
      : id_(id), fe_(fe), n_(0), sift_(true) {
    std::vector<rgbdfe_keypoint> kp;
    std::vector<float> desc;
    SiftGPUWrapper(fe, max_keypoints).detect(gray, rows, cols, kp, desc);
    const int m = (int)kp.size();
    std::vector<float> xy((size_t)m * 2);
    for (int i = 0; i < m; ++i) { xy[(size_t)2 * i] = kp[(size_t)i].x; xy[(size_t)2 * i + 1] = kp[(size_t)i].y; }
    std::vector<int32_t> kept((size_t)(m > 0 ? m : 1));
    feature_locations_3d_.resize((size_t)(m > 0 ? m : 1) * 4);
    siftgpu_descriptors_.resize((size_t)(m > 0 ? m : 1) * 128);
    int32_t n = 0;
    if (m > 0 && rgbdfe_sift_node_features(fe_.get(), xy.data(), m, desc.data(), depth, rows, cols, fx, fy, cx, cy, depth_scaling,
                                           max_keypoints, 0, kept.data(), feature_locations_3d_.data(),
                                           siftgpu_descriptors_.data(), nullptr, &n) != RGBDFE_OK)
      n = 0;
    n_ = n;
    feature_locations_3d_.resize((size_t)n * 4);
    siftgpu_descriptors_.resize((size_t)n * 128);
    for (int i = 0; i < n; ++i) feature_locations_2d_.push_back(kp[(size_t)kept[(size_t)i]]);
    matchable_ = n > 0 && rgbdfe_upload_sift_node(fe_.get(), id_, siftgpu_descriptors_.data(), feature_locations_3d_.data(), n) == RGBDFE_OK;
  }
  // The point-cloud constructor (src/node.cpp:218-369): detect -> projectTo3D(cloud) (maximum_depth, truncating lookup,
  // the max_keypoints cut, :855-898) -> cv::ORB::compute.  cloud: rows x cols x (x, y, z, rgb) floats, organised.
  struct FromPointCloud {};
  Node(const FrontEnd& fe, int id, const uint8_t* gray, const uint8_t* mask, const float* cloud, int rows, int cols,
       double maximum_depth, int max_keypoints, FromPointCloud)
      : id_(id), fe_(fe), n_(0) {
    std::vector<rgbdfe_keypoint> kp((size_t)max_keypoints);
    feature_descriptors_.resize((size_t)max_keypoints * 32);
    feature_locations_3d_.resize((size_t)max_keypoints * 4);
    int32_t n = 0;
    if (rgbdfe_detect_describe_cloud(fe_.get(), gray, mask, cloud, rows, cols, maximum_depth, kp.data(),
                                     feature_descriptors_.data(), feature_locations_3d_.data(), &n) != RGBDFE_OK)
      n = 0;
    n_ = n;
    feature_descriptors_.resize((size_t)n * 32);
    feature_locations_3d_.resize((size_t)n * 4);
    feature_locations_2d_.assign(kp.begin(), kp.begin() + n);
    matchable_ = n > 0 && rgbdfe_upload_node(fe_.get(), id_, feature_descriptors_.data(), feature_locations_3d_.data(), n) == RGBDFE_OK;
  }
  ~Node() { clearFeatureInformation(); }
  Node(const Node&) = delete;
  Node& operator=(const Node&) = delete;

  // src/node.cpp:1305-1429.  Never throws; no edge <=> mr.edge.id1 == -1 && mr.edge.id2 == -1.
  MatchingResult matchNodePair(const Node* older_node) const {
    rgbdfe_match_result pod;
    if (!matchable_ || !older_node || !older_node->matchable_) return MatchingResult();
    if (sift_) {  // SiftGPUWrapper::match semantics; the distances are the float L2 of the raw descriptors
      std::vector<float> dist((size_t)RGBDFE_MAX_MATCHES);
      if (rgbdfe_match_sift_pair_list(fe_.get(), &id_, &older_node->id_, 1, &pod, dist.data()) != RGBDFE_OK)
        return MatchingResult();
      return toMatchingResult(pod, dist.data());
    }
    if (rgbdfe_match_node_pairs(fe_.get(), id_, &older_node->id_, 1, &pod) != RGBDFE_OK) return MatchingResult();
    return toMatchingResult(pod);
  }
  // src/node.cpp:535-690 (ORB branch): matches sorted by (hd, queryIdx), at most max_matches
  unsigned int featureMatching(const Node* other, std::vector<DMatch>* matches) const {
    *matches = matchNodePair(other).all_matches;
    return (unsigned int)matches->size();
  }
  // Node::pc_col (createXYZRGBPointCloud, src/misc.cpp:467-556): built on the device from the depth image and kept
  // resident for the environment measurement model; rgb may be null
  bool setPointCloud(const float* depth, int rows, int cols, const uint8_t* rgb, int rgb_channels, bool encoding_bgr,
                     double fx, double fy, double cx, double cy, double depth_scaling, double minimum_depth,
                     int cloud_creation_skip_step) {
    return rgbdfe_upload_node_cloud(fe_.get(), id_, depth, rows, cols, rgb, rgb_channels, encoding_bgr ? 1 : 0, fx, fy,
                                    cx, cy, depth_scaling, minimum_depth, cloud_creation_skip_step, nullptr) == RGBDFE_OK;
  }
  // feature_locations_2d_ (node.h:160): only the g2o pair refinement reads them (params.g2o_iterations > 0,
  // node.cpp:1222-1268); kp_xy = n x (u, v)
  bool setKeypoints(const float* kp_xy) { return rgbdfe_upload_node_keypoints(fe_.get(), id_, kp_xy, n_) == RGBDFE_OK; }
  void clearFeatureInformation() {  // src/node.cpp:1431-1443
    if (matchable_) rgbdfe_release_node(fe_.get(), id_);
    matchable_ = false;
  }
  int id_;
  bool matchable_ = false;
  // filled by the depth-image constructor only (node.h:160-170)
  std::vector<rgbdfe_keypoint> feature_locations_2d_;
  std::vector<uint8_t> feature_descriptors_;   // n x 32
  std::vector<float> feature_locations_3d_;    // n x (x, y, z, 1)
  std::vector<float> siftgpu_descriptors_;     // n x 128 (node.h:172; SiftGPU nodes built from an image)
  int featureCount() const { return n_; }

 private:
  const FrontEnd& fe_;
  int n_;
  bool sift_ = false;
  friend class GraphManager;
};

class GraphManager {  // candidate selection (graph_manager.cpp:204-324) + the fan-out of nodeComparisons (:531-583)
 public:
  explicit GraphManager(const FrontEnd& fe)
      : fe_(fe), topology_(rgbdfe_pose_graph_create(), &rgbdfe_pose_graph_destroy) {
    if (!topology_) throw std::runtime_error("rgbdfe_pose_graph_create failed");
  }
  // what addNode / addEdgeToG2O tell the pose graph (graph_manager.cpp:681, :811); candidate selection reads it
  void nodeAdded(const Node* n, int vertex_id, bool keyframe) {
    rgbdfe_pose_graph_add_node(topology_.get(), n->id_, vertex_id, n->matchable_ ? 1 : 0, keyframe ? 1 : 0);
  }
  void edgeAdded(int id1, int id2) { rgbdfe_pose_graph_add_edge(topology_.get(), id1, id2); }
  // QList<int> GraphManager::getPotentialEdgeTargetsWithDijkstra(new_node, sequential_targets, geodesic_targets,
  // sampled_targets, predecessor_id, include_predecessor); geodesic_depth is the parameter server's "geodesic_depth".
  // rand_fn == nullptr: reproducible draws from `seed` instead of rand().
  std::vector<int> getPotentialEdgeTargetsWithDijkstra(const Node* /*new_node*/, int sequential_targets,
                                                       int geodesic_targets, int sampled_targets,
                                                       int predecessor_id = -1, bool include_predecessor = false,
                                                       int geodesic_depth = 3, rgbdfe_rand_fn rand_fn = nullptr,
                                                       void* rand_state = nullptr, uint32_t seed = 0) const {
    std::vector<int32_t> ids((size_t)(sequential_targets + geodesic_targets + sampled_targets + 1));
    int32_t n = 0;
    if (rgbdfe_potential_edge_targets(topology_.get(), sequential_targets, geodesic_targets, sampled_targets,
                                      geodesic_depth, predecessor_id, include_predecessor ? 1 : 0, rand_fn, rand_state,
                                      seed, ids.data(), (int32_t)ids.size(), &n) != RGBDFE_OK)
      return {};
    return std::vector<int>(ids.begin(), ids.begin() + n);
  }
  // GraphManager::getNeighbours (loop_closing.cpp:190-277): candidates ranked by the votes of the new node's
  // descriptors (exact Hamming neighbours); returns at most max_out node ids, best first
  std::vector<int> getNeighbours(const Node* new_node, const std::vector<const Node*>& candidates, int neighbour_cnt,
                                 int max_out, int max_hd = 256) const {
    std::vector<int32_t> ids, out((size_t)std::max(max_out, 0));
    std::vector<float> score(out.size());
    for (const Node* n : candidates) ids.push_back(n->id_);
    int32_t n_out = 0;
    if (ids.empty() || out.empty() ||
        rgbdfe_place_recognition(fe_.get(), new_node->id_, ids.data(), (int32_t)ids.size(), neighbour_cnt, max_hd,
                                 (int32_t)out.size(), out.data(), score.data(), &n_out) != RGBDFE_OK)
      return {};
    return std::vector<int>(out.begin(), out.begin() + n_out);
  }
  // 1:1 replacement of QtConcurrent::blockingMapped(nodes_to_comp, bind(&Node::matchNodePair, new_node, _1))
  std::vector<MatchingResult> nodeComparisons(const Node* new_node, const std::vector<const Node*>& nodes_to_comp) const {
    std::vector<int32_t> ids;
    ids.reserve(nodes_to_comp.size());
    for (const Node* n : nodes_to_comp) ids.push_back(n->id_);
    std::vector<rgbdfe_match_result> pods(ids.size());
    std::vector<MatchingResult> out(ids.size());
    if (ids.empty()) return out;
    if (new_node->sift_) {  // matcher_type == "SIFTGPU": the same fan-out over the SIFT pair op
      std::vector<int32_t> q(ids.size(), new_node->id_);
      std::vector<float> dist(ids.size() * (size_t)RGBDFE_MAX_MATCHES);
      if (rgbdfe_match_sift_pair_list(fe_.get(), q.data(), ids.data(), (int32_t)ids.size(), pods.data(), dist.data()) != RGBDFE_OK)
        return out;
      for (size_t i = 0; i < pods.size(); ++i) out[i] = toMatchingResult(pods[i], dist.data() + i * (size_t)RGBDFE_MAX_MATCHES);
      return out;
    }
    if (rgbdfe_match_node_pairs(fe_.get(), new_node->id_, ids.data(), (int32_t)ids.size(), pods.data()) != RGBDFE_OK)
      return out;
    for (size_t i = 0; i < pods.size(); ++i) out[i] = toMatchingResult(pods[i]);
    return out;
  }

 private:
  const FrontEnd& fe_;
  std::unique_ptr<rgbdfe_pose_graph, void (*)(rgbdfe_pose_graph*)> topology_;
};

// 4x4 inverse of a column-major float matrix (the reference calls Eigen's Matrix4f::inverse(), node.cpp:1536):
// Gauss-Jordan with partial pivoting in double, rounded to float
inline std::array<float, 16> inverse4(const std::array<float, 16>& T) {
  double a[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) { a[r][c] = T[c * 4 + r]; a[r][c + 4] = r == c ? 1.0 : 0.0; }
  for (int i = 0; i < 4; ++i) {
    int p = i;
    for (int r = i + 1; r < 4; ++r) if (std::fabs(a[r][i]) > std::fabs(a[p][i])) p = r;
    for (int c = 0; c < 8; ++c) std::swap(a[i][c], a[p][c]);
    const double d = a[i][i];
    for (int c = 0; c < 8; ++c) a[i][c] /= d;
    for (int r = 0; r < 4; ++r)
      if (r != i) { const double f = a[r][i]; for (int c = 0; c < 8; ++c) a[r][c] -= f * a[i][c]; }
  }
  std::array<float, 16> out;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[c * 4 + r] = (float)a[r][c + 4];
  return out;
}

// pairwiseObservationLikelihood (src/node.cpp:1520-1554) + observation_criterion_met (src/misc.cpp:1136-1148) for a
// batch of results, as matchNodePair applies them when observability_threshold > 0 (node.cpp:1340-1343): an edge that
// fails the criterion is withdrawn (edge ids -1).  Both nodes of every edge need setPointCloud().
inline bool pairwiseObservationLikelihood(const FrontEnd& fe, std::vector<MatchingResult>& results, int emm_skip_step,
                                          double observability_threshold) {
  std::vector<int32_t> new_ids, old_ids;
  std::vector<float> T;
  std::vector<size_t> which;
  for (size_t i = 0; i < results.size(); ++i) {
    const MatchingResult& mr = results[i];
    if (mr.edge.id1 < 0) continue;
    const std::array<float, 16> inv = inverse4(mr.final_trafo);
    new_ids.push_back(mr.edge.id2); old_ids.push_back(mr.edge.id1); T.insert(T.end(), mr.final_trafo.begin(), mr.final_trafo.end());
    new_ids.push_back(mr.edge.id1); old_ids.push_back(mr.edge.id2); T.insert(T.end(), inv.begin(), inv.end());
    which.push_back(i);
  }
  if (which.empty()) return true;
  std::vector<rgbdfe_emm_counts> c(new_ids.size());
  if (rgbdfe_observation_likelihood(fe.get(), (int32_t)new_ids.size(), new_ids.data(), old_ids.data(), T.data(),
                                    emm_skip_step, c.data()) != RGBDFE_OK)
    return false;
  for (size_t k = 0; k < which.size(); ++k) {
    MatchingResult& mr = results[which[k]];
    const rgbdfe_emm_counts &a = c[2 * k], &b = c[2 * k + 1];
    mr.inlier_points = a.inliers + b.inliers;     // node.cpp:1548-1551
    mr.outlier_points = a.outliers + b.outliers;
    mr.occluded_points = a.occluded + b.occluded;
    mr.all_points = a.all + b.all;
    double quality = 0.0;
    if (!rgbdfe_observation_criterion_met(mr.inlier_points, mr.outlier_points,
                                          mr.occluded_points + mr.inlier_points + mr.outlier_points,
                                          observability_threshold, &quality))
      mr.edge.id1 = mr.edge.id2 = -1;  // node.cpp:1419-1422
  }
  return true;
}

}  // namespace rgbdslam
#endif
