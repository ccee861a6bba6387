#pragma once
// declaration-only stand-in (see README.md)
#include <memory>
#include <vector>
#include <ros/ros.h>
namespace pcl {
struct PCLHeader { unsigned seq; unsigned long stamp; };
template <class PointT> class PointCloud {
  public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr; // boost::shared_ptr in PCL 1.10
    PCLHeader header;
    std::vector<PointT> points;
};
}
