// Oracle shim (test infrastructure): geometry_msgs as surfel_map.cpp reads and writes them.
#pragma once
#include "ros/ros.h"
namespace geometry_msgs {
struct Point { double x, y, z; Point() : x(0), y(0), z(0) {} };
struct Quaternion { double x, y, z, w; Quaternion() : x(0), y(0), z(0), w(0) {} };
struct Vector3 { double x, y, z; Vector3() : x(0), y(0), z(0) {} };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct PointStamped { std_msgs::Header header; Point point; };
struct PoseWithCovariance { Pose pose; double covariance[36]; PoseWithCovariance() { for (int i = 0; i < 36; i++) covariance[i] = 0; } };
struct Twist { Vector3 linear, angular; };
struct TwistWithCovariance { Twist twist; double covariance[36]; };
}  // namespace geometry_msgs
