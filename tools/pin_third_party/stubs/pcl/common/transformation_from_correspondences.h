// declaration-only stand-in (see ../../README.md)
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace pcl {
class TransformationFromCorrespondences {
 public:
  TransformationFromCorrespondences();
  void add(const Eigen::Vector3f& point, const Eigen::Vector3f& corresponding_point, float weight = 1.0);
  Eigen::Affine3f getTransformation();
};
}  // namespace pcl
