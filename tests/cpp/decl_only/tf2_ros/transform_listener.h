#pragma once
// declaration-only stand-in (see README.md): the tf2_ros names include/groundgrid/GroundGrid.h and the binding mention
#include <geometry_msgs/TransformStamped.h>
#include <string>
namespace tf2 {
class LookupException { public: const char *what() const; };
class ExtrapolationException { public: const char *what() const; };
}
namespace tf2_ros {
class Buffer {
  public:
    Buffer();
    geometry_msgs::TransformStamped lookupTransform(const std::string &target_frame, const std::string &source_frame, const ros::Time &time) const;
};
class TransformListener {
  public:
    explicit TransformListener(Buffer &buffer);
};
}
