#pragma once
#include <memory>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace pcl {
namespace search {
template <typename P>
struct KdTree {
  using Ptr = std::shared_ptr<KdTree<P>>;
};
}  // namespace search
template <typename P, typename N, typename F>
struct FPFHEstimationOMP {  // only named by the reference's fpfh.h; the matcher never touches it
  using Ptr = std::shared_ptr<FPFHEstimationOMP<P, N, F>>;
};
}  // namespace pcl
