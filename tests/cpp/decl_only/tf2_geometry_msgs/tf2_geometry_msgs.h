#pragma once
// declaration-only stand-in (see README.md): included by the reference's header, nothing of it is used by the binding
#include <geometry_msgs/PointStamped.h>
