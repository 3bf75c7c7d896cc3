// oracle/ref_stubs -- TEST INFRASTRUCTURE ONLY.  A minimal stand-in for the handful of OpenCV types the
// reference's src/feature_adjuster.cpp and src/features.cpp:42-60 use, so that those files can be compiled
// FROM WHERE THEY LIE (/root/reference) without OpenCV, into oracle/_ref/libref_adjuster.so.  The reference's
// grid / threshold-adaptation control flow then runs unmodified around an injected detector (the oracle's own
// restatement of cv::ORB::detect), which pins orb_grid_detect's control logic on the reference's code.
// Nothing here is product code, and nothing here is copied from OpenCV.
#ifndef REF_STUB_OPENCV_FEATURES2D_HPP
#define REF_STUB_OPENCV_FEATURES2D_HPP
#include <algorithm>
#include <cmath>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#define CV_WRAP
#define CV_OUT

namespace cv {

typedef unsigned char uchar;

struct Point2f { float x, y; };
struct KeyPoint {
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};
struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
};
struct Range {
  int start, end;
  Range(int s = 0, int e = 0) : start(s), end(e) {}
};

// a non-owning 8-bit single-channel view
struct Mat {
  int rows = 0, cols = 0;
  const uchar* data = nullptr;
  int step = 0;
  Mat() {}
  Mat(int r, int c, const uchar* d, int s) : rows(r), cols(c), data(d), step(s) {}
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  Size size() const { return Size(cols, rows); }
  Mat operator()(const Range& r, const Range& c) const {
    return Mat(r.end - r.start, c.end - c.start, data + (size_t)r.start * step + c.start, step);
  }
  template <class T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + c); }
};

struct _InputArray {
  Mat m;
  _InputArray() {}
  _InputArray(const Mat& mm) : m(mm) {}
  Mat getMat() const { return m; }
};
typedef const _InputArray& InputArray;
inline _InputArray noArray() { return _InputArray(); }

// cv::Ptr: shared ownership, constructible and assignable from a raw pointer
template <class T>
struct Ptr : std::shared_ptr<T> {
  Ptr() {}
  Ptr(T* p) : std::shared_ptr<T>(p) {}
  template <class U> Ptr(const Ptr<U>& o) : std::shared_ptr<T>(o) {}
};

struct Feature2D {
  virtual ~Feature2D() {}
  virtual void detect(InputArray image, std::vector<KeyPoint>& keypoints, InputArray mask = noArray()) {
    (void)image; (void)keypoints; (void)mask;
  }
};
typedef Feature2D DescriptorExtractor;

// the injected detector: (sub-image view, mask view, FAST threshold) -> keypoints
typedef void (*ref_stub_detect_fn)(const Mat& image, const Mat& mask, int fast_threshold,
                                   std::vector<KeyPoint>& keypoints);
extern ref_stub_detect_fn ref_stub_detect;

struct ORB : Feature2D {
  int fast_threshold;
  explicit ORB(int t) : fast_threshold(t) {}
  static Ptr<Feature2D> create(int = 500, float = 1.2f, int = 8, int = 31, int = 0, int = 2, int = 0, int = 31,
                               int fastThreshold = 20) {
    return Ptr<Feature2D>(new ORB(fastThreshold));
  }
  void detect(InputArray image, std::vector<KeyPoint>& keypoints, InputArray mask = noArray()) override {
    ref_stub_detect(image.getMat(), mask.getMat(), fast_threshold, keypoints);
  }
};
struct FastFeatureDetector : Feature2D {
  static Ptr<Feature2D> create(int = 10) { return Ptr<Feature2D>(new FastFeatureDetector()); }
};

}  // namespace cv
#endif
