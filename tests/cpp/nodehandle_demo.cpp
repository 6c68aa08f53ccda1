// nodehandle_demo.cpp — compile-only check that PatchWork's NodeHandle-style constructor (reference
// include/patchwork.hpp:47-139: nh->param(name, var, default), nh->getParam(name, var)) accepts any object with
// ros::NodeHandle's two members, so `new PatchWork<PointType>(&nh)` keeps compiling where ROS exists.
#include <map>
#include <string>
#include <vector>

#include "patchwork.hpp"

struct FakeNodeHandle {
  std::map<std::string, double> d;
  std::map<std::string, std::vector<double>> vd;
  std::map<std::string, std::vector<int>> vi;
  template <typename T>
  bool param(const std::string& name, T& var, const T& def) const {
    auto it = d.find(name);
    var = it == d.end() ? def : static_cast<T>(it->second);
    return it != d.end();
  }
  bool getParam(const std::string& name, int& v) const {
    auto it = d.find(name);
    if (it != d.end()) v = static_cast<int>(it->second);
    return it != d.end();
  }
  bool getParam(const std::string& name, std::vector<double>& v) const {
    auto it = vd.find(name);
    if (it != vd.end()) v = it->second;
    return it != vd.end();
  }
  bool getParam(const std::string& name, std::vector<int>& v) const {
    auto it = vi.find(name);
    if (it != vi.end()) v = it->second;
    return it != vi.end();
  }
};

int main() {
  FakeNodeHandle nh;
  nh.d["/patchwork/sensor_height"] = 1.723;
  nh.d["/patchwork/czm/num_zones"] = 4;
  nh.vi["/patchwork/czm/num_sectors_each_zone"] = {16, 32, 54, 32};
  nh.vi["/patchwork/czm/num_rings_each_zone"] = {2, 4, 4, 4};
  nh.vd["/patchwork/czm/min_ranges_each_zone"] = {2.7, 12.3625, 22.025, 41.35};
  nh.vd["/patchwork/czm/elevation_thresholds"] = {-1.2, -0.9984, -0.851, -0.605};
  nh.vd["/patchwork/czm/flatness_thresholds"] = {0.0001, 0.000125, 0.000185, 0.000185};
  PatchWork<pcl::PointXYZ> pw(&nh);
  return pw.params().num_zones == 4 && pw.params().num_thr == 4 ? 0 : 1;
}
