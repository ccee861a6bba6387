#pragma once
// declaration-only stand-in (see README.md)
#include <geometry_msgs/PoseWithCovarianceStamped.h>
namespace geometry_msgs {
struct PointStamped { std_msgs::Header header; Point point; };
}
