#pragma once
// declaration-only stand-in (see README.md): the macros include/velodyne_pointcloud/point_types.h is written with
#include <cstdint>
#define PCL_ADD_POINT4D union { float data[4]; struct { float x; float y; float z; }; }
#define EIGEN_ALIGN16 __attribute__((aligned(16)))
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define POINT_CLOUD_REGISTER_POINT_STRUCT(...)
