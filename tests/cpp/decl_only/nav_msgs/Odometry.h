#pragma once
// declaration-only stand-in (see README.md): nav_msgs/Odometry as GroundGrid reads it (header, pose.pose.position)
#include <geometry_msgs/PoseWithCovarianceStamped.h>
#include <memory>
namespace nav_msgs {
struct Odometry { std_msgs::Header header; geometry_msgs::PoseWithCovariance pose; };
typedef std::shared_ptr<Odometry const> OdometryConstPtr; // boost::shared_ptr in ROS 1
}
