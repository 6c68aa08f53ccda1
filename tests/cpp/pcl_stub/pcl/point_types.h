// Stand-in for <pcl/point_types.h>: the point records on the drop-in API, with PCL's sizes and member names
// (PointXYZ 16 B, PointXYZI 32 B, Normal 32 B, FPFHSignature33 132 B).
#pragma once
#include <Eigen/Core>
namespace pcl {
struct EIGEN_ALIGN16 PointXYZ {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  PointXYZ() : data{0.f, 0.f, 0.f, 1.f} {}
  PointXYZ(float x_, float y_, float z_) : data{x_, y_, z_, 1.f} {}
};
struct EIGEN_ALIGN16 PointXYZI {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union {
    struct {
      float intensity;
    };
    float data_c[4];
  };
  PointXYZI() : data{0.f, 0.f, 0.f, 1.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
};
struct EIGEN_ALIGN16 Normal {
  union {
    float data_n[4];
    float normal[3];
    struct {
      float normal_x, normal_y, normal_z;
    };
  };
  union {
    struct {
      float curvature;
    };
    float data_c[4];
  };
  Normal() : data_n{0.f, 0.f, 0.f, 0.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
};
struct FPFHSignature33 {
  float histogram[33] = {0.f};
  static int descriptorSize() { return 33; }
};
}  // namespace pcl
