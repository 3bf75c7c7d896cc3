// oracle/ref_stubs/adjuster_glue.cpp -- TEST INFRASTRUCTURE ONLY.
// C entry points around the reference's own detector wiring (src/features.cpp:42-60, adjustedGridWrapper) and
// src/feature_adjuster.cpp (DetectorAdjuster / VideoDynamicAdaptedFeatureDetector /
// VideoGridAdaptedFeatureDetector), compiled from /root/reference by oracle/Makefile.  The innermost
// cv::ORB::detect is the oracle's restatement (orb_detect, oracle/orb_oracle.c).
#include <cstdint>
#include <vector>

#include "feature_adjuster.h"
#include "parameter_server.h"
#include "../orb_oracle.h"

StatefulFeatureDetector* adjustedGridWrapper(cv::Ptr<DetectorAdjuster> detadj);  // src/features.cpp:48

namespace cv { ref_stub_detect_fn ref_stub_detect = nullptr; }

static void oracle_orb_detect(const cv::Mat& image, const cv::Mat& mask, int fast_threshold,
                              std::vector<cv::KeyPoint>& keypoints) {
  std::vector<orb_keypoint> buf(60000);
  const int n = orb_detect(image.data, mask.empty() ? nullptr : mask.data, image.cols, image.rows, image.step,
                           mask.empty() ? 0 : mask.step, fast_threshold, buf.data(), (int)buf.size());
  keypoints.clear();
  for (int i = 0; i < n; ++i) {
    cv::KeyPoint k;
    k.pt.x = buf[i].x; k.pt.y = buf[i].y; k.size = buf[i].size; k.angle = buf[i].angle;
    k.response = buf[i].response; k.octave = buf[i].octave; k.class_id = -1;
    keypoints.push_back(k);
  }
}

extern "C" void* ref_grid_detector_create(int max_keypoints, int grid_res, int max_iters) {
  cv::ref_stub_detect = oracle_orb_detect;
  ParameterServer* ps = ParameterServer::instance();
  ps->max_keypoints = max_keypoints;
  ps->detector_grid_resolution = grid_res;
  ps->adjuster_max_iterations = max_iters;
  // createDetector("ORB") with grid and dynamic wrapping (src/features.cpp:92-104)
  return adjustedGridWrapper(new DetectorAdjuster("ORB", 20));
}
extern "C" void ref_grid_detector_destroy(void* h) { delete static_cast<StatefulFeatureDetector*>(h); }
extern "C" int ref_grid_detector_detect(void* h, const uint8_t* img, const uint8_t* mask, int cols, int rows,
                                        orb_keypoint* out, int cap) {
  std::vector<cv::KeyPoint> kps;
  cv::Mat image(rows, cols, img, cols), m;
  if (mask) m = cv::Mat(rows, cols, mask, cols);
  static_cast<StatefulFeatureDetector*>(h)->detect(cv::_InputArray(image), kps, cv::_InputArray(m));
  int n = 0;
  for (const cv::KeyPoint& k : kps) {
    if (n >= cap) break;
    out[n].x = k.pt.x; out[n].y = k.pt.y; out[n].size = k.size; out[n].angle = k.angle;
    out[n].response = k.response; out[n].octave = k.octave;
    ++n;
  }
  return n;
}
