// oracle/ref_stubs -- TEST INFRASTRUCTURE ONLY: logging macros of ROS as no-ops.
#ifndef REF_STUB_ROS_H
#define REF_STUB_ROS_H
#define ROS_INFO(...)
#define ROS_WARN(...)
#define ROS_ERROR(...)
#define ROS_INFO_STREAM(x)
#endif
