// declaration-only stand-in (see ../README.md)
#pragma once
#include "core.hpp"
namespace cv {
class Feature2D {
 public:
  virtual ~Feature2D();
  virtual void detect(InputArray image, std::vector<KeyPoint>& keypoints, InputArray mask);
  virtual void compute(InputArray image, std::vector<KeyPoint>& keypoints, OutputArray descriptors);
};
class ORB : public Feature2D {
 public:
  static Ptr<ORB> create(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8, int edgeThreshold = 31, int firstLevel = 0,
                         int WTA_K = 2, int scoreType = 0, int patchSize = 31, int fastThreshold = 20);
};
class KeyPointsFilter {
 public:
  static void retainBest(std::vector<KeyPoint>& keypoints, int npoints);
};
}  // namespace cv
