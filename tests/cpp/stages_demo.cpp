// stages_demo.cpp — the reference's individually callable pieces written against THIS repository's headers:
// teaser::FPFHEstimation::computeFPFHFeatures + teaser::Matcher::calculateCorrespondences (the two calls inside
// FPFHManager::setFeaturePair, include/fpfh_manager.hpp:118-127), then the public stage methods of class Quatro
// (computeTIMs, solveForScale, solveForRotation, solveForTranslation; include/quatro.hpp:307-615) on the matched
// key points, and finally computeTransformation for comparison.
// usage: stages_demo src.bin tgt.bin seed
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>

#include "quatro.hpp"
#include "teaser_utils/feature_matcher.h"
#include "teaser_utils/fpfh.h"

static teaser::PointCloud load(const char* path) {
  teaser::PointCloud c;
  std::ifstream f(path, std::ios::binary);
  float rec[4];
  while (f.read(reinterpret_cast<char*>(rec), sizeof(rec))) c.push_back({rec[0], rec[1], rec[2]});
  return c;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  teaser::PointCloud src = load(argv[1]), tgt = load(argv[2]);
  teaser::FPFHEstimation fpfh;
  pcl::PointCloud<pcl::Normal> ns, nt;
  teaser::FPFHCloudPtr fs = fpfh.computeFPFHFeatures(src, ns, 0.5, 0.75);
  teaser::FPFHCloudPtr ft = fpfh.computeFPFHFeatures(tgt, nt, 0.5, 0.75);
  teaser::Matcher matcher;
  matcher.seed_ = std::strtoull(argv[3], nullptr, 10);
  auto corr = matcher.calculateCorrespondences(src, tgt, *fs, *ft, true, true, true, 0.95f);
  std::printf("n_src %zu n_tgt %zu L %zu\n", src.size(), tgt.size(), corr.size());
  std::printf("corr");
  for (size_t i = 0; i < corr.size() && i < 8; ++i) std::printf(" %d:%d", corr[i].first, corr[i].second);
  std::printf("\n");

  const int L = static_cast<int>(corr.size());
  using Q = Quatro<pcl::PointXYZ, pcl::PointXYZ>;
  Q quatro;
  Q::Params p;
  p.noise_bound = 0.3;
  p.cbar2 = 1.0;
  p.rotation_gnc_factor = 1.4;
  p.rotation_max_iterations = 50;
  p.rotation_cost_threshold = 1.1e-4;
  p.inlier_selection_mode = Q::INLIER_SELECTION_MODE::PMC_HEU;
  quatro.reset(p);
  // stage by stage on the first 200 correspondences (TIMs are quadratic)
  const int N = L < 200 ? L : 200;
  Eigen::Matrix<double, 3, Eigen::Dynamic> a(3, N), b(3, N);
  for (int c = 0; c < N; ++c) {
    a(0, c) = src[static_cast<size_t>(corr[c].first)].x;
    a(1, c) = src[static_cast<size_t>(corr[c].first)].y;
    a(2, c) = src[static_cast<size_t>(corr[c].first)].z;
    b(0, c) = tgt[static_cast<size_t>(corr[c].second)].x;
    b(1, c) = tgt[static_cast<size_t>(corr[c].second)].y;
    b(2, c) = tgt[static_cast<size_t>(corr[c].second)].z;
  }
  Eigen::Matrix<int, 2, Eigen::Dynamic> map;
  auto ta = quatro.computeTIMs(a, &map);
  auto tb = quatro.computeTIMs(b, &map);
  Eigen::Matrix<bool, 1, Eigen::Dynamic> mask;
  double scale = 0;
  quatro.solveForScale(ta, tb, &scale, &mask);
  long long edges = 0;
  for (int c = 0; c < mask.cols(); ++c) edges += mask(0, c) ? 1 : 0;
  std::printf("tims %d scale %.1f consistent_pairs %lld last_map %d %d\n", static_cast<int>(ta.cols()), scale, edges,
              map(0, static_cast<int>(map.cols()) - 1), map(1, static_cast<int>(map.cols()) - 1));
  const Eigen::Matrix3d R = quatro.solveForRotation(a, b);
  std::printf("R %.17g %.17g %.17g %.17g\n", R(0, 0), R(0, 1), R(1, 0), R(1, 1));
  Eigen::Matrix<double, 3, Eigen::Dynamic> ra(3, N);
  for (int c = 0; c < N; ++c)
    for (int r = 0; r < 3; ++r) ra(r, c) = (R(r, 0) * a(0, c) + R(r, 1) * a(1, c)) + R(r, 2) * a(2, c);
  const Eigen::Vector3d t = quatro.solveForTranslation(ra, b, true);
  std::printf("t %.17g %.17g %.17g\n", t(0, 0), t(1, 0), t(2, 0));
  // reg_name "TEASER": the 3-DoF branch the reference names but throws for (include/quatro.hpp:409-411)
  Q teaser;
  p.reg_name = "TEASER";
  teaser.reset(p);
  const Eigen::Matrix3d R3 = teaser.solveForRotation(a, b);
  std::printf("R3");
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) std::printf(" %.17g", R3(r, c));
  std::printf("\n");
  Eigen::Matrix4d T3 = Eigen::Matrix4d::Identity();
  pcl::PointCloud<pcl::PointXYZ>::Ptr srcM(new pcl::PointCloud<pcl::PointXYZ>());
  pcl::PointCloud<pcl::PointXYZ>::Ptr tgtM(new pcl::PointCloud<pcl::PointXYZ>());
  for (int c = 0; c < L; ++c) {
    srcM->push_back(pcl::PointXYZ(src[static_cast<size_t>(corr[c].first)].x, src[static_cast<size_t>(corr[c].first)].y,
                                  src[static_cast<size_t>(corr[c].first)].z));
    tgtM->push_back(pcl::PointXYZ(tgt[static_cast<size_t>(corr[c].second)].x, tgt[static_cast<size_t>(corr[c].second)].y,
                                  tgt[static_cast<size_t>(corr[c].second)].z));
  }
  teaser.reset(p);
  teaser.setInputSource(srcM);
  teaser.setInputTarget(tgtM);
  teaser.computeTransformation(T3);
  std::printf("T3");
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf(" %.17g", T3(r, c));
  std::printf("\n");
  return 0;
}
