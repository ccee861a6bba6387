#pragma once
// declaration-only stand-in (see README.md)
namespace ros {
class NodeHandle;
struct Time { unsigned sec, nsec; };
}
namespace decl_only { void log(const char *fmt, ...); }
#define ROS_ERROR(...) ::decl_only::log(__VA_ARGS__)
#define ROS_FATAL(...) ::decl_only::log(__VA_ARGS__)
#define ROS_WARN(...) ::decl_only::log(__VA_ARGS__)
#define ROS_WARN_ONCE(...) ::decl_only::log(__VA_ARGS__)
#define ROS_INFO(...) ::decl_only::log(__VA_ARGS__)
#define ROS_DEBUG(...) ::decl_only::log(__VA_ARGS__)
