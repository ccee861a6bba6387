#pragma once
// declaration-only stand-in (see README.md)
namespace sensor_msgs { struct PointCloud2; }
