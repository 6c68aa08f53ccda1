#pragma once
namespace pcl {
struct PointXYZ {
  float x, y, z, pad;
};
struct Normal {
  float normal_x, normal_y, normal_z, curvature;
};
struct FPFHSignature33 {
  float histogram[33];
  static int descriptorSize() { return 33; }
};
}  // namespace pcl
