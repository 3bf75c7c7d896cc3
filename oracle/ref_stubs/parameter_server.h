// oracle/ref_stubs -- TEST INFRASTRUCTURE ONLY: the three parameters src/features.cpp:42-60 reads.
#ifndef REF_STUB_PARAMETER_SERVER_H
#define REF_STUB_PARAMETER_SERVER_H
#include <cstring>
#include <string>
struct ParameterServer {
  int max_keypoints = 1000, detector_grid_resolution = 3, adjuster_max_iterations = 5;
  static ParameterServer* instance() { static ParameterServer p; return &p; }
  template <class T> T get(const std::string& name) {
    if (name == "max_keypoints") return (T)max_keypoints;
    if (name == "detector_grid_resolution") return (T)detector_grid_resolution;
    if (name == "adjuster_max_iterations") return (T)adjuster_max_iterations;
    return T();
  }
  template <class T> void set(const std::string&, const T&) {}
};
#endif
