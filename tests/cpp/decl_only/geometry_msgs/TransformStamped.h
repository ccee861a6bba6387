#pragma once
// declaration-only stand-in (see README.md): the fields of geometry_msgs/TransformStamped the bindings read
#include <ros/ros.h>
#include <string>
namespace std_msgs {
struct Header { unsigned seq; ros::Time stamp; std::string frame_id; };
}
namespace geometry_msgs {
struct Vector3 { double x, y, z; };
struct Quaternion { double x, y, z, w; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::Header header; Transform transform; };
}
