#pragma once
// declaration-only stand-in (see README.md): the fields of geometry_msgs/PoseWithCovarianceStamped GroundGrid keeps
#include <geometry_msgs/TransformStamped.h>
namespace geometry_msgs {
struct Point { double x, y, z; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; double covariance[36]; };
struct PoseWithCovarianceStamped { std_msgs::Header header; PoseWithCovariance pose; };
}
