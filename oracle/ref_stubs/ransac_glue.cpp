// oracle/ref_stubs/ransac_glue.cpp -- TEST INFRASTRUCTURE ONLY: C entry point around the reference's own
// Node::getRelativeTransformationTo (see ransac_prelude.h).  Appended by oracle/Makefile after the reference's
// line ranges in one translation unit.
RefParams g_ref;

// called by getRelativeTransformationTo once per RANSAC iteration: iteration k draws rand31(seed, uid, k, 0..)
std::vector<cv::DMatch> sample_matches_prefer_by_distance(unsigned int sample_size, std::vector<cv::DMatch>& m) {
  g_ref.iter = (uint32_t)g_ref.sampler_calls++;
  g_ref.k = 0;
  return ref_sampler_impl(sample_size, m);
}

extern "C" int ref_get_relative_transformation(
    const float* qxyz1, int nq, const float* txyz1, int nt, const int32_t* mq, const int32_t* mt,
    const float* mdist, int n_matches, int min_matches, int ransac_iterations, double max_dist_for_inliers,
    double depth_cov, uint32_t seed, uint32_t uid, float* T_colmajor, float* rmse_out, int32_t* inl_q,
    int32_t* inl_t, int* n_inl_out, int* real_iterations_out) {
  g_ref = RefParams();
  g_ref.min_matches = min_matches;
  g_ref.ransac_iterations = ransac_iterations;
  g_ref.max_dist_for_inliers = max_dist_for_inliers;
  g_ref.depth_cov = depth_cov;
  g_ref.seed = seed;
  g_ref.uid = uid;
  Node newer, older;
  newer.id_ = 1; older.id_ = 0;
  for (int i = 0; i < nq; ++i) newer.feature_locations_3d_.push_back(Eigen::Vector4f(qxyz1[4 * i], qxyz1[4 * i + 1], qxyz1[4 * i + 2], qxyz1[4 * i + 3]));
  for (int i = 0; i < nt; ++i) older.feature_locations_3d_.push_back(Eigen::Vector4f(txyz1[4 * i], txyz1[4 * i + 1], txyz1[4 * i + 2], txyz1[4 * i + 3]));
  std::vector<cv::DMatch> initial((size_t)n_matches), inliers;
  for (int i = 0; i < n_matches; ++i) { initial[i].queryIdx = mq[i]; initial[i].trainIdx = mt[i]; initial[i].imgIdx = 0; initial[i].distance = mdist[i]; }
  Eigen::Matrix4f T;
  float rmse = 0.f;  // MatchingResult() default (matching_result.h:27): untouched on the early return
  const bool found = newer.getRelativeTransformationTo(&older, &initial, T, rmse, inliers);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T_colmajor[c * 4 + r] = T.m[r][c];
  *rmse_out = rmse;
  *n_inl_out = (int)inliers.size();
  for (size_t i = 0; i < inliers.size(); ++i) { inl_q[i] = inliers[i].queryIdx; inl_t[i] = inliers[i].trainIdx; }
  *real_iterations_out = g_ref.sampler_calls;
  return found ? 1 : 0;
}

// The whole pair op: Node::matchNodePair = featureMatching (bruteForceSearchORB, keepStrongestMatches) ->
// getRelativeTransformationTo -> edge assembly (src/node.cpp:1305-1429).
extern "C" int ref_match_node_pair(
    const uint8_t* qdesc, const float* qxyz1, int nq, int qid, const uint8_t* tdesc, const float* txyz1, int nt, int tid,
    int max_matches, int min_matches, int ransac_iterations, double max_dist_for_inliers, double depth_cov,
    uint32_t seed, uint32_t uid, int32_t* all_q, int32_t* all_t, float* all_dist, int* n_all_out, int32_t* inl_q,
    int32_t* inl_t, int* n_inl_out, float* T_colmajor, float* rmse_out, int* id1_out, int* id2_out,
    double* info00_out, int* real_iterations_out) {
  g_ref = RefParams();
  g_ref.max_matches = max_matches;
  g_ref.min_matches = min_matches;
  g_ref.ransac_iterations = ransac_iterations;
  g_ref.max_dist_for_inliers = max_dist_for_inliers;
  g_ref.depth_cov = depth_cov;
  g_ref.seed = seed;
  g_ref.uid = uid;
  Node newer, older;
  newer.id_ = qid; older.id_ = tid;
  newer.feature_descriptors_.rows = nq; newer.feature_descriptors_.cols = 32;
  newer.feature_descriptors_.data = const_cast<uint8_t*>(qdesc);
  older.feature_descriptors_.rows = nt; older.feature_descriptors_.cols = 32;
  older.feature_descriptors_.data = const_cast<uint8_t*>(tdesc);
  newer.feature_locations_2d_.resize((size_t)nq);
  older.feature_locations_2d_.resize((size_t)nt);
  for (int i = 0; i < nq; ++i) newer.feature_locations_3d_.push_back(Eigen::Vector4f(qxyz1[4 * i], qxyz1[4 * i + 1], qxyz1[4 * i + 2], qxyz1[4 * i + 3]));
  for (int i = 0; i < nt; ++i) older.feature_locations_3d_.push_back(Eigen::Vector4f(txyz1[4 * i], txyz1[4 * i + 1], txyz1[4 * i + 2], txyz1[4 * i + 3]));
  MatchingResult mr = newer.matchNodePair(&older);
  *n_all_out = (int)mr.all_matches.size();
  for (size_t i = 0; i < mr.all_matches.size(); ++i) {
    all_q[i] = mr.all_matches[i].queryIdx; all_t[i] = mr.all_matches[i].trainIdx; all_dist[i] = mr.all_matches[i].distance;
  }
  *n_inl_out = (int)mr.inlier_matches.size();
  for (size_t i = 0; i < mr.inlier_matches.size(); ++i) { inl_q[i] = mr.inlier_matches[i].queryIdx; inl_t[i] = mr.inlier_matches[i].trainIdx; }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T_colmajor[c * 4 + r] = mr.final_trafo.m[r][c];
  *rmse_out = mr.rmse;
  *id1_out = mr.edge.id1;
  *id2_out = mr.edge.id2;
  *info00_out = mr.edge.id1 >= 0 ? mr.edge.informationMatrix.m[0][0] : 0.0;
  *real_iterations_out = g_ref.sampler_calls;
  return (int)newer.initial_node_matches_;
}
