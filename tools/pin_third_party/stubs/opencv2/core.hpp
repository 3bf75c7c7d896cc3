// declaration-only stand-in (see ../README.md)
#pragma once
#include <vector>
#define CV_8UC1 0
namespace cv {
struct Point2f { float x, y; };
class KeyPoint {
 public:
  KeyPoint();
  KeyPoint(float x, float y, float size, float angle = -1, float response = 0, int octave = 0, int class_id = -1);
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};
class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type, void* data);
  bool isContinuous() const;
  Mat clone() const;
  int rows, cols;
  unsigned char* data;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
template <typename T> class Ptr {
 public:
  Ptr();
  template <typename U> Ptr(const Ptr<U>&);
  T* operator->() const;
};
}  // namespace cv
