// dropin_demo.cpp — the reference demo's call sequence (examples/run_global_registration.cpp:92-108,
// 206-221, 243-246) written against include/quatro.hpp + include/fpfh_manager.hpp of THIS repository.
// usage: dropin_demo src.bin tgt.bin [seed]     (.bin = float32 x,y,z,intensity records, as KITTI)
// Prints the 4x4 transform (17 significant digits), clique size and final inlier indices.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <vector>

#include "fpfh_manager.hpp"
#include "imageProjection.hpp"
#include "patchwork.hpp"
#include "quatro.hpp"

static pcl::PointCloud<PointType>::Ptr getCloud(const char* path) {  // reference :377-402
  pcl::PointCloud<PointType>::Ptr cloud(new pcl::PointCloud<PointType>());
  std::vector<float> buffer(1000000);  // the demo's cap: 250 000 points
  int n = 0;
  if (qtr_read_kitti_bin(path, buffer.data(), 250000, &n) != QTR_OK) throw std::runtime_error(std::string("error: failed to load ") + path);
  for (int i = 0; i < n; ++i) cloud->push_back(PointType(buffer[4 * i], buffer[4 * i + 1], buffer[4 * i + 2]));
  return cloud;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s src.bin tgt.bin [seed]\n", argv[0]);
    return 2;
  }
  auto srcRaw = getCloud(argv[1]);
  auto tgtRaw = getCloud(argv[2]);
  using QuatroT = Quatro<PointType, PointType>;
  QuatroT quatro;
  QuatroT::Params params;  // setParams(...) of the demo, config/params.yaml values
  params.noise_bound = 0.3;
  params.cbar2 = 1.0;
  params.rotation_gnc_factor = 1.4;
  params.rotation_max_iterations = 50;
  params.rotation_cost_threshold = 1.1e-4;
  params.estimate_scaling = false;
  params.inlier_selection_mode = QuatroT::INLIER_SELECTION_MODE::PMC_HEU;
  quatro.reset(params);

  // optional 4th argument "raw": the demo's STEP 2 first (reference :136-146) — Patchwork ground removal on raw scans —
  // then STEP 3 as under "segment"
  const bool raw = argc > 4 && std::string(argv[4]) == "raw";
  if (raw) {
    PatchWork<PointType> patchwork;
    pcl::PointCloud<PointType> srcGround, tgtGround;
    pcl::PointCloud<PointType>::Ptr srcNonground(new pcl::PointCloud<PointType>());
    pcl::PointCloud<PointType>::Ptr tgtNonground(new pcl::PointCloud<PointType>());
    double tSrc = 0, tTgt = 0;
    patchwork.estimate_ground(*srcRaw, srcGround, *srcNonground, tSrc);
    patchwork.estimate_ground(*tgtRaw, tgtGround, *tgtNonground, tTgt);
    std::printf("ground %zu %zu nonground %zu %zu\n", srcGround.size(), tgtGround.size(), srcNonground->size(),
                tgtNonground->size());
    srcRaw = srcNonground;
    tgtRaw = tgtNonground;
  }
  // optional 4th argument "segment": the demo's STEP 3 (reference :124-160) — range-image sub-cluster rejection
  // before voxelisation (the inputs then play the role of the non-ground clouds)
  if (raw || (argc > 4 && std::string(argv[4]) == "segment")) {
    ImageProjection IPSrc("Velodyne-64-HDE", "4CrossNeighbor", "Patchwork"), IPTgt("Velodyne-64-HDE", "4CrossNeighbor", "Patchwork");
    IPSrc.segmentCloud(srcRaw);
    IPTgt.segmentCloud(tgtRaw);
    pcl::PointCloud<PointType>::Ptr srcValid(new pcl::PointCloud<PointType>());
    pcl::PointCloud<PointType>::Ptr tgtValid(new pcl::PointCloud<PointType>());
    IPSrc.getValidSegments(*srcValid);
    IPTgt.getValidSegments(*tgtValid);
    pcl::PointCloud<PointType> so, to;
    IPSrc.getOutliers(so);
    IPTgt.getOutliers(to);
    std::printf("segments %d %d valid %zu %zu outliers %zu %zu\n", IPSrc.numSegments(), IPTgt.numSegments(), srcValid->size(),
                tgtValid->size(), so.size(), to.size());
    srcRaw = srcValid;
    tgtRaw = tgtValid;
  }
  pcl::PointCloud<PointType>::Ptr srcFeat(new pcl::PointCloud<PointType>());
  pcl::PointCloud<PointType>::Ptr tgtFeat(new pcl::PointCloud<PointType>());
  voxelize(srcRaw, srcFeat, 0.3);
  voxelize(tgtRaw, tgtFeat, 0.3);
  FPFHManager fpfhmanager(0.5, 0.75);
  fpfhmanager.seed_ = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 0;
  fpfhmanager.flushAllFeatures();
  fpfhmanager.setFeaturePair(srcFeat, tgtFeat);
  pcl::PointCloud<PointType>::Ptr srcMatched(new pcl::PointCloud<PointType>(fpfhmanager.getSrcKps()));
  pcl::PointCloud<PointType>::Ptr tgtMatched(new pcl::PointCloud<PointType>(fpfhmanager.getTgtKps()));

  quatro.setInputSource(srcMatched);
  quatro.setInputTarget(tgtMatched);
  Eigen::Matrix4d output = Eigen::Matrix4d::Identity();
  quatro.computeTransformation(output);

  std::printf("n_src %zu n_tgt %zu L %zu valid %d clique %d rot_inliers %d\n", srcFeat->size(), tgtFeat->size(),
              srcMatched->size(), quatro.solution_.valid ? 1 : 0, quatro.getNumMaxCliqueInliers(),
              quatro.getNumRotaionInliers());
  for (int r = 0; r < 4; ++r)
    std::printf("T %.17g %.17g %.17g %.17g\n", output(r, 0), output(r, 1), output(r, 2), output(r, 3));
  std::printf("final");
  for (int i : quatro.getFinalInliersIndices()) std::printf(" %d", i);
  std::printf("\n");
  return 0;
}
