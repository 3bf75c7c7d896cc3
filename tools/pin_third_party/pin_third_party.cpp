// pin_third_party.cpp -- runs the THIRD-PARTY arithmetic of the hot path (OpenCV 3.x cv::ORB / KeyPointsFilter, PCL's
// TransformationFromCorrespondences, Eigen's JacobiSVD and LLT) on the committed inputs and writes what it returns, so that
// oracle/orb_oracle.c and oracle/rgbd_oracle.c -- restatements of those libraries' published algorithms, "parity unpinned"
// on the build box where none of them exists -- can be pinned on the real thing in one command:
//
//     python tools/pin_third_party/export_inputs.py inputs.pin          (any machine with this repository)
//     cmake -S tools/pin_third_party -B build_pin && cmake --build build_pin
//     build_pin/pin_third_party inputs.pin pins.pin                     (a machine with OpenCV 3.x, Eigen 3.2+, PCL 1.7+)
//     python tools/pin_third_party/compare_pins.py inputs.pin pins.pin  (back here: bit-equality per field vs the oracle)
//
// Every call below is made the way the reference makes it; the call sites are cited.  It CANNOT be built on the round's
// build box (no OpenCV / Eigen / PCL there): it is syntax-checked against the declaration-only headers under stubs/
// (tests/test_pin_third_party.py), and the comparator is tested on pins written from the oracle itself.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Cholesky>
#include <Eigen/Geometry>
#include <Eigen/SVD>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <pcl/common/transformation_from_correspondences.h>

#include "pinfile.hpp"

static void put_keypoints(pin::Writer& w, const std::string& name, const std::vector<cv::KeyPoint>& kps) {
  // x, y, size, angle, response as f32 and octave as i32: the six KeyPoint fields SURVEY.md 8(a) a1 lists
  std::vector<float> f(kps.size() * 5);
  std::vector<int32_t> o(kps.size());
  for (size_t i = 0; i < kps.size(); ++i) {
    f[i * 5 + 0] = kps[i].pt.x; f[i * 5 + 1] = kps[i].pt.y; f[i * 5 + 2] = kps[i].size;
    f[i * 5 + 3] = kps[i].angle; f[i * 5 + 4] = kps[i].response;
    o[i] = kps[i].octave;
  }
  w.put(name + "_f", pin::F32, {(uint64_t)kps.size(), 5}, f.data());
  w.put(name + "_octave", pin::I32, {(uint64_t)kps.size()}, o.data());
}

static std::vector<cv::KeyPoint> get_keypoints(const pin::Array& f, const pin::Array& o) {
  std::vector<cv::KeyPoint> kps(o.count());
  for (size_t i = 0; i < kps.size(); ++i) {
    const float* p = f.as<float>() + i * 5;
    kps[i] = cv::KeyPoint(p[0], p[1], p[2], p[3], p[4], o.as<int32_t>()[i]);
  }
  return kps;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s inputs.pin pins.pin\n", argv[0]); return 2; }
  const auto in = pin::read(argv[1]);
  pin::Writer out(argv[2]);
  const int n_img = in.at("n_images").as<int32_t>()[0];
  for (int i = 0; i < n_img; ++i) {
    const std::string tag = "img" + std::to_string(i);
    const pin::Array& g = in.at(tag + "_gray");
    const pin::Array& m = in.at(tag + "_mask");
    const int rows = (int)g.dims[0], cols = (int)g.dims[1];
    const cv::Mat gray(rows, cols, CV_8UC1, const_cast<unsigned char*>(g.as<unsigned char>()));
    const cv::Mat mask(rows, cols, CV_8UC1, const_cast<unsigned char*>(m.as<unsigned char>()));
    const pin::Array& thr = in.at(tag + "_fast_thresholds");
    for (size_t t = 0; t < thr.count(); ++t) {
      // DetectorAdjuster::detect, feature_adjuster.cpp:94,121
      cv::Ptr<cv::Feature2D> detector = cv::ORB::create(10000, 1.2, 8, 15, 0, 2, 0, 31, thr.as<int32_t>()[t]);
      std::vector<cv::KeyPoint> kps;
      detector->detect(gray, kps, mask);
      put_keypoints(out, tag + "_detect" + std::to_string(t), kps);
      if (t == 0) {
        // Node::Node, node.cpp:187-191: retainBest + resize, then node.cpp:202 with createDescriptorExtractor("ORB") =
        // cv::ORB::create() (features.cpp:117-119)
        const int max_keyp = in.at(tag + "_max_keypoints").as<int32_t>()[0];
        std::vector<cv::KeyPoint> best = kps;
        if ((int)best.size() > max_keyp) {
          cv::KeyPointsFilter::retainBest(best, max_keyp);
          const int32_t before_resize = (int32_t)best.size();
          out.put(tag + "_retain_best_size", pin::I32, {1}, &before_resize);
          best.resize(max_keyp);
        }
        put_keypoints(out, tag + "_retained", best);
        cv::Ptr<cv::Feature2D> extractor = cv::ORB::create();
        cv::Mat desc;
        extractor->compute(gray, best, desc);
        put_keypoints(out, tag + "_described", best);  // compute() may drop keypoints near the border
        cv::Mat dc = desc.isContinuous() ? desc : desc.clone();
        out.put(tag + "_desc", pin::U8, {(uint64_t)dc.rows, 32}, dc.data);
      }
    }
    // cv::ORB::compute at GIVEN keypoints (the oracle's detections travel in inputs.pin: pins rBRIEF independently of detect)
    if (in.count(tag + "_given_f")) {
      std::vector<cv::KeyPoint> given = get_keypoints(in.at(tag + "_given_f"), in.at(tag + "_given_octave"));
      cv::Mat desc;
      cv::ORB::create()->compute(gray, given, desc);
      put_keypoints(out, tag + "_given_described", given);
      cv::Mat dc = desc.isContinuous() ? desc : desc.clone();
      out.put(tag + "_given_desc", pin::U8, {(uint64_t)dc.rows, 32}, dc.data);
    }
  }
  // KeyPointsFilter::retainBest on lists with ties at the cut (node.cpp:189; feature_adjuster.cpp:247-255 uses nth_element itself)
  const int n_rb = in.at("n_retain").as<int32_t>()[0];
  for (int i = 0; i < n_rb; ++i) {
    const std::string tag = "retain" + std::to_string(i);
    std::vector<cv::KeyPoint> kps = get_keypoints(in.at(tag + "_f"), in.at(tag + "_octave"));
    cv::KeyPointsFilter::retainBest(kps, in.at(tag + "_n").as<int32_t>()[0]);
    put_keypoints(out, tag + "_kept", kps);
  }
  // getTransformFromMatches, transformation_estimation_euclidean.cpp:13-60: tfc.add(from, to, 1/(from.z*to.z)) in list order
  const int n_fit = in.at("n_fits").as<int32_t>()[0];
  for (int i = 0; i < n_fit; ++i) {
    const std::string tag = "fit" + std::to_string(i);
    const pin::Array& from = in.at(tag + "_from");
    const pin::Array& to = in.at(tag + "_to");
    pcl::TransformationFromCorrespondences tfc;
    for (size_t k = 0; k < (size_t)from.dims[0]; ++k) {
      const Eigen::Vector3f f(from.as<float>()[k * 3], from.as<float>()[k * 3 + 1], from.as<float>()[k * 3 + 2]);
      const Eigen::Vector3f t(to.as<float>()[k * 3], to.as<float>()[k * 3 + 1], to.as<float>()[k * 3 + 2]);
      if (std::isnan(f(2)) || std::isnan(t(2))) continue;
      float weight = 1.0;
      weight = 1.0 / (f(2) * t(2));
      tfc.add(f, t, weight);
    }
    const Eigen::Matrix4f T = tfc.getTransformation().matrix();
    float row_major[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) row_major[r * 4 + c] = T(r, c);
    out.put(tag + "_T", pin::F32, {4, 4}, row_major);
  }
  // Eigen::JacobiSVD<Matrix3f> as pcl::TransformationFromCorrespondences::getTransformation uses it (ComputeFullU | ComputeFullV)
  {
    const pin::Array& A = in.at("svd_in");
    const size_t n = (size_t)A.dims[0];
    std::vector<float> U(n * 9), S(n * 3), V(n * 9);
    for (size_t k = 0; k < n; ++k) {
      Eigen::Matrix<float, 3, 3> a;
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a(r, c) = A.as<float>()[k * 9 + r * 3 + c];
      Eigen::JacobiSVD<Eigen::Matrix<float, 3, 3> > svd(a, Eigen::ComputeFullU | Eigen::ComputeFullV);
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { U[k * 9 + r * 3 + c] = svd.matrixU()(r, c); V[k * 9 + r * 3 + c] = svd.matrixV()(r, c); }
      for (int r = 0; r < 3; ++r) S[k * 3 + r] = svd.singularValues()(r);
    }
    out.put("svd_U", pin::F32, {(uint64_t)n, 3, 3}, U.data());
    out.put("svd_S", pin::F32, {(uint64_t)n, 3}, S.data());
    out.put("svd_V", pin::F32, {(uint64_t)n, 3, 3}, V.data());
  }
  // errorFunction2's solve, misc.cpp:763: d^T * S.llt().solve(d)
  {
    const pin::Array& Sm = in.at("llt_S");
    const pin::Array& d = in.at("llt_d");
    const size_t n = (size_t)Sm.dims[0];
    std::vector<double> q(n);
    for (size_t k = 0; k < n; ++k) {
      Eigen::Matrix3d S;
      Eigen::Vector3d v;
      for (int r = 0; r < 3; ++r) { v(r) = d.as<double>()[k * 3 + r]; for (int c = 0; c < 3; ++c) S(r, c) = Sm.as<double>()[k * 9 + r * 3 + c]; }
      q[k] = v.transpose() * S.llt().solve(v);
    }
    out.put("llt_q", pin::F64, {(uint64_t)n}, q.data());
  }
  printf("wrote %s\n", argv[2]);
  return 0;
}
