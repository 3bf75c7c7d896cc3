// oracle/ref_stubs/frame_prelude.h -- TEST INFRASTRUCTURE ONLY.
// Stand-ins for the OpenCV / PCL / ROS / parameter-server types used by the reference's frame-level functions,
// so that these compile FROM WHERE THEY LIE in /root/reference into oracle/_ref/libref_frame.so:
//   removeDepthless                 src/node.cpp:66-97
//   Node::projectTo3DSiftGPU        src/node.cpp:695-769
//   Node::projectTo3D               src/node.cpp:900-965 (depth image) and :855-898 (point cloud)
//   squareroot_descriptor_space     src/node.cpp:1557-1571
//   backProject                     src/misc2.h:49-65
//   getCameraIntrinsics*            src/misc.cpp:56-69
//   createXYZRGBPointCloud          src/misc.cpp:452-556
//   round / cdf / observationLikelihood  src/misc.cpp:800-969
//   observation_criterion_met       src/misc.cpp:1136-1148
// Third-party pieces behind the stand-ins (cv::abs / cv::reduce, pcl::transformPointCloud) carry the oracle's
// restated arithmetic; depth_covariance() is D3's explicit value.
#ifndef REF_STUB_FRAME_PRELUDE_H
#define REF_STUB_FRAME_PRELUDE_H
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <list>
#include <memory>
#include <string>
#include <vector>

#define ROS_INFO(...)
#define ROS_INFO_THROTTLE(...)
#define ROS_WARN(...)
#define ROS_ERROR(...)
#define ROS_DEBUG(...)
#define ROS_DEBUG_NAMED(...)
#define ROS_INFO_STREAM(x)
#define ROS_WARN_STREAM(x)
#define ROS_ERROR_STREAM(x)
#define ROS_WARN_COND(c, ...)
#define ROS_INFO_COND(c, ...)
struct ScopedTimer { explicit ScopedTimer(const char*, bool = false, bool = false) {} };

struct FrameParams {
  double depth_scaling_factor = 1.0, maximum_depth = -1.0, minimum_depth = 0.1, observability_threshold = 0.6;
  double depth_cov = 1e-4;
  int max_keypoints = 1000, cloud_creation_skip_step = 2, emm__skip_step = 8;
  bool encoding_bgr = false;
  bool use_feature_min_depth = false;
};
extern FrameParams g_fp;
struct ParameterServer {
  static ParameterServer* instance() { static ParameterServer p; return &p; }
  template <class T> T get(const std::string& n) {
    if (n == "depth_scaling_factor") return (T)g_fp.depth_scaling_factor;
    if (n == "maximum_depth") return (T)g_fp.maximum_depth;
    if (n == "minimum_depth") return (T)g_fp.minimum_depth;
    if (n == "observability_threshold") return (T)g_fp.observability_threshold;
    if (n == "max_keypoints") return (T)g_fp.max_keypoints;
    if (n == "cloud_creation_skip_step") return (T)g_fp.cloud_creation_skip_step;
    if (n == "emm__skip_step") return (T)g_fp.emm__skip_step;
    if (n == "encoding_bgr") return (T)g_fp.encoding_bgr;
    if (n == "use_feature_min_depth") return (T)g_fp.use_feature_min_depth;
    return T();  // depth_camera_fx.. = 0 (use CameraInfo), emm__mark_outliers = false, voxelfilter_size = 0
  }
};
template <> inline std::string ParameterServer::get<std::string>(const std::string&) { return std::string(); }  // topic_points: empty
inline double depth_covariance(double) { return g_fp.depth_cov; }  // D3

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5
#define CV_32F 5
#define CV_REDUCE_SUM 0
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint {
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};
// row-major view / owner of an 8-bit or float image
struct Range { int start, end; Range(int s, int e) : start(s), end(e) {} };
struct Mat {
  int rows = 0, cols = 0, type_ = CV_8UC1;
  int step_elems = 0;  // row stride in elements of a region-of-interest view (0: = cols)
  unsigned char* data = nullptr;
  std::shared_ptr<std::vector<unsigned char> > own;
  Mat() {}
  Mat(int r, int c, int t) : rows(r), cols(c), type_(t), own(new std::vector<unsigned char>((size_t)r * c * esz(t))) { data = own->data(); }
  Mat(int r, int c, int t, void* d) : rows(r), cols(c), type_(t), data((unsigned char*)d) {}
  // region of interest (a view): rows [rr.start, rr.end) x cols [cr.start, cr.end)
  Mat(const Mat& m, const Range& rr, const Range& cr)
      : rows(rr.end > rr.start ? rr.end - rr.start : 0), cols(cr.end > cr.start ? cr.end - cr.start : 0), type_(m.type_),
        step_elems(m.step_elems ? m.step_elems : m.cols),
        data(m.data + ((size_t)rr.start * (m.step_elems ? m.step_elems : m.cols) + cr.start) * esz(m.type_)), own(m.own) {}
  int stride() const { return step_elems ? step_elems : cols; }
  static size_t esz(int t) { return t == CV_32FC1 ? 4 : (t == CV_8UC3 ? 3 : 1); }
  int type() const { return type_; }
  size_t total() const { return (size_t)rows * cols; }
  template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(data)[(size_t)r * cols + c]; }
  template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[(size_t)r * cols + c]; }
  template <class T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
  template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }
};
// cv::minMaxLoc(src, &minVal) on a CV_32F matrix: OpenCV 3.3 minMaxIdx_32f -- `if (val < minVal)` starting from FLT_MAX
// (a NaN never compares less: skipped), and 0 when no element compared (minidx == 0).  Restated: OpenCV is absent.
inline void minMaxLoc(const Mat& m, double* minVal) {
  float mn = 3.402823466e+38f;
  bool found = false;
  for (int r = 0; r < m.rows; ++r)
    for (int c = 0; c < m.cols; ++c) {
      const float v = reinterpret_cast<const float*>(m.data)[(size_t)r * m.stride() + c];
      if (v < mn) { mn = v; found = true; }
    }
  *minVal = found ? (double)mn : 0.0;
}
// cv::abs on a CV_32F matrix
inline Mat abs(const Mat& m) {
  Mat r(m.rows, m.cols, CV_32FC1);
  for (size_t i = 0; i < m.total(); ++i) r.at<float>((int)i) = std::fabs(m.at<float>((int)i));
  return r;
}
// cv::reduce(src, dst, 1, CV_REDUCE_SUM, CV_32FC1): OpenCV 3.3 reduceC_<float, float, OpAdd>, restated as in
// the oracle (two float accumulators over even / odd columns, leftovers to the first, then a0 + a1)
inline void reduce(const Mat& src, Mat& dst, int dim, int op, int dtype) {
  assert(dim == 1 && op == CV_REDUCE_SUM && dtype == CV_32FC1);
  dst = Mat(src.rows, 1, CV_32FC1);
  for (int r = 0; r < src.rows; ++r) {
    const float* d = &src.at<float>(r, 0);
    float sum;
    if (src.cols == 1) {
      sum = d[0];
    } else {
      float a0 = d[0], a1 = d[1];
      int i = 2;
      for (; i <= src.cols - 4; i += 4) { a0 = a0 + d[i]; a1 = a1 + d[i + 1]; a0 = a0 + d[i + 2]; a1 = a1 + d[i + 3]; }
      for (; i < src.cols; ++i) a0 = a0 + d[i];
      sum = a0 + a1;
    }
    dst.at<float>(r) = sum;
  }
}
}  // namespace cv

namespace sensor_msgs {
struct CameraInfo { double K[9]; };
typedef std::shared_ptr<const CameraInfo> CameraInfoConstPtr;
}  // namespace sensor_msgs

namespace Eigen {
template <class T> using aligned_allocator = std::allocator<T>;
struct Vector4f {
  float v[4];
  Vector4f() : v{0, 0, 0, 0} {}
  Vector4f(float x, float y, float z, float w) : v{x, y, z, w} {}
  float operator()(int i) const { return v[i]; }
};
struct Matrix4f {
  float m[4][4];
  float operator()(int r, int c) const { return m[r][c]; }
};
}  // namespace Eigen

// pcl::PointXYZRGB / pcl::PointCloud as far as the reference touches them
struct point_type {
  float x = 0.f, y = 0.f, z = 0.f;
  float rgb = 0.f;
};
struct pointcloud_type {
  typedef std::shared_ptr<pointcloud_type> Ptr;
  typedef std::shared_ptr<const pointcloud_type> ConstPtr;
  typedef std::vector<point_type>::iterator iterator;
  std::vector<point_type> points;
  uint32_t width = 0, height = 0;
  bool is_dense = true;
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  point_type& at(int col, int row) { return points[(size_t)row * width + col]; }
  const point_type& at(int col, int row) const { return points[(size_t)row * width + col]; }
};
namespace pcl {
// pcl::transformPointCloud (PCL 1.7 common/impl/transforms.hpp), non-dense branch: points with a non-finite
// coordinate are copied, the others get x' = T00*x + T01*y + T02*z + T03 (float, left to right)
inline void transformPointCloud(const pointcloud_type& in, pointcloud_type& out, const Eigen::Matrix4f& T) {
  out = in;
  for (size_t i = 0; i < in.points.size(); ++i) {
    const point_type& p = in.points[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    out.points[i].x = T(0, 0) * p.x + T(0, 1) * p.y + T(0, 2) * p.z + T(0, 3);
    out.points[i].y = T(1, 0) * p.x + T(1, 1) * p.y + T(1, 2) * p.z + T(1, 3);
    out.points[i].z = T(2, 0) * p.x + T(2, 1) * p.y + T(2, 2) * p.z + T(2, 3);
  }
}
}  // namespace pcl

float getMinDepthInNeighborhood(const cv::Mat& depth, cv::Point2f center, float diameter);  // misc.cpp:774-793, compiled from the source

class Node {
 public:
  std::vector<float> siftgpu_descriptors;
  void projectTo3D(std::vector<cv::KeyPoint>& feature_locations_2d,
                   std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> >& feature_locations_3d,
                   const cv::Mat& depth, const sensor_msgs::CameraInfoConstPtr& cam_info);
  void projectTo3D(std::vector<cv::KeyPoint>& feature_locations_2d,
                   std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> >& feature_locations_3d,
                   pointcloud_type::ConstPtr point_cloud);
  void projectTo3DSiftGPU(std::vector<cv::KeyPoint>& feature_locations_2d,
                          std::vector<Eigen::Vector4f, Eigen::aligned_allocator<Eigen::Vector4f> >& feature_locations_3d,
                          const cv::Mat& depth, const sensor_msgs::CameraInfoConstPtr& cam_info,
                          std::vector<float>& descriptors_in, cv::Mat& descriptors_out);
};
#endif
