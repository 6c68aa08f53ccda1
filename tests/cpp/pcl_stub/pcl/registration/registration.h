// Stand-in for <pcl/registration/registration.h> (PCL 1.10): the members pcl::Registration exposes to a derived class
// such as Quatro (reference include/quatro.hpp:71-100): virtual setInputSource / setInputTarget, the pure virtual
// computeTransformation(output, guess), Matrix4 = Eigen::Matrix<Scalar, 4, 4>, and the protected state.
#pragma once
#include <string>

#include <pcl/pcl_base.h>
namespace pcl {
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration : public PCLBase<PointSource> {
 public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using Ptr = boost::shared_ptr<Registration<PointSource, PointTarget, Scalar>>;
  using ConstPtr = boost::shared_ptr<const Registration<PointSource, PointTarget, Scalar>>;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;

  Registration() = default;
  ~Registration() override = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { PCLBase<PointSource>::setInputCloud(cloud); }
  const PointCloudSourceConstPtr getInputSource() { return input_; }
  virtual inline void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
  const PointCloudTargetConstPtr getInputTarget() { return target_; }
  inline void setMaximumIterations(int nr_iterations) { max_iterations_ = nr_iterations; }
  inline int getMaximumIterations() { return max_iterations_; }
  inline Matrix4 getFinalTransformation() { return final_transformation_; }
  inline bool hasConverged() const { return converged_; }
  inline const std::string& getClassName() const { return reg_name_; }

 protected:
  using PCLBase<PointSource>::input_;
  std::string reg_name_;
  int nr_iterations_ = 0;
  int max_iterations_ = 10;
  PointCloudTargetConstPtr target_;
  Matrix4 final_transformation_, transformation_, previous_transformation_;
  bool converged_ = false;
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
};
}  // namespace pcl
