// oracle/ref_stubs/frame_glue.cpp -- TEST INFRASTRUCTURE ONLY: C entry points around the reference's frame-level
// functions (see frame_prelude.h); appended after the reference's line ranges by oracle/Makefile.
FrameParams g_fp;

static sensor_msgs::CameraInfoConstPtr make_cam(double fx, double fy, double cx, double cy) {
  sensor_msgs::CameraInfo* c = new sensor_msgs::CameraInfo();
  for (double& k : c->K) k = 0.0;
  c->K[0] = fx; c->K[4] = fy; c->K[2] = cx; c->K[5] = cy; c->K[8] = 1.0;
  return sensor_msgs::CameraInfoConstPtr(c);
}
static std::vector<cv::KeyPoint> make_kps(const float* kp_xy, int n) {
  std::vector<cv::KeyPoint> k((size_t)n);
  for (int i = 0; i < n; ++i) { k[i].pt.x = kp_xy[2 * i]; k[i].pt.y = kp_xy[2 * i + 1]; k[i].size = 31.f; k[i].class_id = i; }
  return k;
}

// use_feature_min_depth (parameter_server.cpp:90): the same reference functions with the parameter switched on and the
// keypoints' real sizes
extern "C" float ref_min_depth_in_neighborhood(const float* depth, int rows, int cols, float x, float y, float diameter) {
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  cv::Point2f c; c.x = x; c.y = y;
  return getMinDepthInNeighborhood(d, c, diameter);
}
extern "C" int ref_remove_depthless_min_depth(const float* kp_xy, const float* kp_size, int n, const float* depth, int rows,
                                              int cols, int32_t* kept) {
  std::vector<cv::KeyPoint> k = make_kps(kp_xy, n);
  for (int i = 0; i < n; ++i) k[i].size = kp_size[i];
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  g_fp.use_feature_min_depth = true;
  removeDepthless(k, d);
  g_fp.use_feature_min_depth = false;
  for (size_t i = 0; i < k.size(); ++i) kept[i] = k[i].class_id;
  return (int)k.size();
}
extern "C" int ref_project_to_3d_min_depth(const float* kp_xy, const float* kp_size, int n, const float* depth, int rows,
                                           int cols, double fx, double fy, double cx, double cy, double depth_scaling,
                                           int max_keypoints, int32_t* kept, float* xyz1) {
  g_fp.depth_scaling_factor = depth_scaling; g_fp.max_keypoints = max_keypoints;
  std::vector<cv::KeyPoint> k = make_kps(kp_xy, n);
  for (int i = 0; i < n; ++i) k[i].size = kp_size[i];
  std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> > p3;
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  Node node;
  g_fp.use_feature_min_depth = true;
  node.projectTo3D(k, p3, d, make_cam(fx, fy, cx, cy));
  g_fp.use_feature_min_depth = false;
  for (size_t i = 0; i < p3.size(); ++i) { kept[i] = k[i].class_id; for (int c = 0; c < 4; ++c) xyz1[4 * i + c] = p3[i](c); }
  return (int)p3.size();
}
extern "C" int ref_remove_depthless(const float* kp_xy, int n, const float* depth, int rows, int cols, int32_t* kept) {
  std::vector<cv::KeyPoint> k = make_kps(kp_xy, n);
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  removeDepthless(k, d);
  for (size_t i = 0; i < k.size(); ++i) kept[i] = k[i].class_id;
  return (int)k.size();
}
extern "C" int ref_project_to_3d(const float* kp_xy, int n, const float* depth, int rows, int cols, double fx, double fy,
                                 double cx, double cy, double depth_scaling, int max_keypoints, int32_t* kept, float* xyz1) {
  g_fp.depth_scaling_factor = depth_scaling; g_fp.max_keypoints = max_keypoints;
  std::vector<cv::KeyPoint> k = make_kps(kp_xy, n);
  std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> > p3;
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  Node node;
  node.projectTo3D(k, p3, d, make_cam(fx, fy, cx, cy));
  for (size_t i = 0; i < p3.size(); ++i) { kept[i] = k[i].class_id; for (int c = 0; c < 4; ++c) xyz1[4 * i + c] = p3[i](c); }
  return (int)p3.size();
}
extern "C" int ref_project_to_3d_cloud(const float* kp_xy, int n, const float* cloud_xyzrgb, int rows, int cols,
                                       double maximum_depth, int max_keypoints, int32_t* kept, float* xyz1) {
  g_fp.maximum_depth = maximum_depth; g_fp.max_keypoints = max_keypoints;
  std::vector<cv::KeyPoint> k = make_kps(kp_xy, n);
  std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> > p3;
  pointcloud_type* pc = new pointcloud_type();
  pc->width = (uint32_t)cols; pc->height = (uint32_t)rows; pc->is_dense = false;
  pc->points.resize((size_t)rows * cols);
  for (size_t i = 0; i < pc->points.size(); ++i) {
    pc->points[i].x = cloud_xyzrgb[4 * i]; pc->points[i].y = cloud_xyzrgb[4 * i + 1]; pc->points[i].z = cloud_xyzrgb[4 * i + 2];
  }
  Node node;
  node.projectTo3D(k, p3, pointcloud_type::ConstPtr(pc));
  for (size_t i = 0; i < p3.size(); ++i) { kept[i] = k[i].class_id; for (int c = 0; c < 4; ++c) xyz1[4 * i + c] = p3[i](c); }
  return (int)p3.size();
}
extern "C" int ref_project_to_3d_sift(const float* kp_xy, int n, const float* desc_in, const float* depth, int rows, int cols,
                                      double fx, double fy, double cx, double cy, double depth_scaling, int max_keypoints,
                                      int32_t* kept, float* xyz1, float* desc_out, float* siftgpu_out) {
  g_fp.depth_scaling_factor = depth_scaling; g_fp.max_keypoints = max_keypoints;
  std::vector<cv::KeyPoint> k = make_kps(kp_xy, n);
  std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> > p3;
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  std::vector<float> din(desc_in, desc_in + (size_t)n * 128);
  cv::Mat dout;
  Node node;
  node.projectTo3DSiftGPU(k, p3, d, make_cam(fx, fy, cx, cy), din, dout);
  for (size_t i = 0; i < p3.size(); ++i) { kept[i] = k[i].class_id; for (int c = 0; c < 4; ++c) xyz1[4 * i + c] = p3[i](c); }
  if (!p3.empty()) {
    std::memcpy(desc_out, dout.data, p3.size() * 128 * 4);
    std::memcpy(siftgpu_out, node.siftgpu_descriptors.data(), p3.size() * 128 * 4);
  }
  return (int)p3.size();
}
extern "C" int ref_project_to_3d_sift_min_depth(const float* kp_xy, const float* kp_size, int n, const float* desc_in,
                                                const float* depth, int rows, int cols, double fx, double fy, double cx,
                                                double cy, double depth_scaling, int max_keypoints, int32_t* kept, float* xyz1) {
  g_fp.depth_scaling_factor = depth_scaling; g_fp.max_keypoints = max_keypoints;
  std::vector<cv::KeyPoint> k = make_kps(kp_xy, n);
  for (int i = 0; i < n; ++i) k[i].size = kp_size[i];
  std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> > p3;
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  std::vector<float> din(desc_in, desc_in + (size_t)n * 128);
  cv::Mat dout;
  Node node;
  g_fp.use_feature_min_depth = true;
  node.projectTo3DSiftGPU(k, p3, d, make_cam(fx, fy, cx, cy), din, dout);
  g_fp.use_feature_min_depth = false;
  for (size_t i = 0; i < p3.size(); ++i) { kept[i] = k[i].class_id; for (int c = 0; c < 4; ++c) xyz1[4 * i + c] = p3[i](c); }
  return (int)p3.size();
}
extern "C" void ref_root_sift(float* desc, int n_rows, int dim) {
  cv::Mat m(n_rows, dim, CV_32FC1, desc);
  squareroot_descriptor_space(m);  // `descriptors = cv::abs(descriptors)` re-seats the Mat: copy the result back
  std::memcpy(desc, m.data, (size_t)n_rows * dim * 4);
}
extern "C" void ref_create_point_cloud(const float* depth, int rows, int cols, const uint8_t* rgb, int channels, int encoding_bgr,
                                       double fx, double fy, double cx, double cy, double depth_scaling, double min_depth,
                                       int skip, float* cloud_out) {
  g_fp.depth_scaling_factor = depth_scaling; g_fp.minimum_depth = min_depth; g_fp.cloud_creation_skip_step = skip;
  g_fp.encoding_bgr = encoding_bgr != 0;
  cv::Mat d(rows, cols, CV_32FC1, (void*)depth);
  cv::Mat c(rows, cols, channels == 3 ? CV_8UC3 : CV_8UC1, (void*)rgb);
  pointcloud_type* pc = createXYZRGBPointCloud(d, c, make_cam(fx, fy, cx, cy));
  for (size_t i = 0; i < pc->points.size(); ++i) {
    cloud_out[4 * i] = pc->points[i].x; cloud_out[4 * i + 1] = pc->points[i].y; cloud_out[4 * i + 2] = pc->points[i].z;
    std::memcpy(cloud_out + 4 * i + 3, &pc->points[i].rgb, 4);
  }
  delete pc;
}
extern "C" void ref_observation_likelihood(const float* new_cloud, const float* old_cloud, int ch, int cw, const float* T_rowmajor,
                                           double fx, double fy, double cx, double cy, int cloud_skip, int skip_step,
                                           double depth_cov, uint32_t counts[4]) {
  g_fp.cloud_creation_skip_step = cloud_skip; g_fp.emm__skip_step = skip_step; g_fp.depth_cov = depth_cov;
  g_fp.observability_threshold = 0.6;
  auto make = [&](const float* c) {
    pointcloud_type::Ptr p(new pointcloud_type());
    p->width = (uint32_t)cw; p->height = (uint32_t)ch; p->is_dense = false;
    p->points.resize((size_t)ch * cw);
    for (size_t i = 0; i < p->points.size(); ++i) { p->points[i].x = c[4 * i]; p->points[i].y = c[4 * i + 1]; p->points[i].z = c[4 * i + 2]; }
    return p;
  };
  Eigen::Matrix4f T;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T.m[r][c] = T_rowmajor[r * 4 + c];
  double likelihood = 0, confidence = 0;
  unsigned int inl = 0, outl = 0, occ = 0, all = 0;
  observationLikelihood(T, make(new_cloud), make(old_cloud), *make_cam(fx, fy, cx, cy), likelihood, confidence, inl, outl, occ, all);
  counts[0] = inl; counts[1] = outl; counts[2] = occ; counts[3] = all;
}
extern "C" int ref_observation_criterion_met(unsigned int inliers, unsigned int outliers, unsigned int all, double thresh, double* quality) {
  g_fp.observability_threshold = thresh;
  return observation_criterion_met(inliers, outliers, all, *quality) ? 1 : 0;
}
