// make_pcl_golden.cpp — runs the REFERENCE's own call sequences into PCL and PMC on fixed inputs and writes the outputs
// as golden vectors (see README.md in this directory).  Built and run by a maintainer on a machine that has PCL and PMC;
// never part of the product and not built in this repository's image (neither library is installed here).
//
//   make_pcl_golden <inputs dir> <output dir>
//
// Call sequences (url-kaist/Quatro):
//   voxel grid    voxelize<T>()                                  include/quatro.hpp:49-68
//   normals/FPFH  teaser::FPFHEstimation::computeFPFHFeatures    src/teaser_utils/fpfh.cc:44-75
//   clique        teaser::MaxCliqueSolver::findMaxClique         src/graph.cc:12-104 (PMC_HEU branch), one thread
#include <pcl/features/fpfh_omp.h>
#include <pcl/features/normal_3d.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/search/kdtree.h>

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "pmc/pmc.h"

namespace {
void write_bin(const std::string& path, int rows, int cols, int dtype, const void* data) {
  std::ofstream f(path, std::ios::binary);
  const int32_t hdr[3] = {rows, cols, dtype};
  f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
  f.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(rows) * cols * 4);
}

// getCloud of the reference demo (examples/run_global_registration.cpp:377-402): x, y, z, intensity float32 records
pcl::PointCloud<pcl::PointXYZ>::Ptr read_kitti_bin(const std::string& path) {
  pcl::PointCloud<pcl::PointXYZ>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZ>);
  std::ifstream f(path, std::ios::binary);
  float rec[4];
  while (f.read(reinterpret_cast<char*>(rec), sizeof(rec))) cloud->push_back(pcl::PointXYZ(rec[0], rec[1], rec[2]));
  return cloud;
}

void front_end(const std::string& in, const std::string& out, const std::string& tag) {
  pcl::PointCloud<pcl::PointXYZ>::Ptr raw = read_kitti_bin(in + "/" + tag + ".bin");
  // include/quatro.hpp:49-68
  pcl::PointCloud<pcl::PointXYZ>::Ptr vox(new pcl::PointCloud<pcl::PointXYZ>);
  pcl::VoxelGrid<pcl::PointXYZ> voxel_filter;
  voxel_filter.setInputCloud(raw);
  voxel_filter.setLeafSize(0.3f, 0.3f, 0.3f);
  voxel_filter.filter(*vox);
  std::vector<float> v(vox->size() * 3);
  for (size_t i = 0; i < vox->size(); ++i) {
    v[3 * i] = (*vox)[i].x;
    v[3 * i + 1] = (*vox)[i].y;
    v[3 * i + 2] = (*vox)[i].z;
  }
  write_bin(out + "/vox_" + tag + ".bin", static_cast<int>(vox->size()), 3, 0, v.data());
  // src/teaser_utils/fpfh.cc:44-75
  pcl::PointCloud<pcl::Normal>::Ptr normals(new pcl::PointCloud<pcl::Normal>);
  pcl::NormalEstimation<pcl::PointXYZ, pcl::Normal> normalEstimation;
  normalEstimation.setInputCloud(vox);
  normalEstimation.setRadiusSearch(0.5);
  pcl::search::KdTree<pcl::PointXYZ>::Ptr kdtree(new pcl::search::KdTree<pcl::PointXYZ>);
  normalEstimation.setSearchMethod(kdtree);
  normalEstimation.compute(*normals);
  pcl::FPFHEstimationOMP<pcl::PointXYZ, pcl::Normal, pcl::FPFHSignature33> fpfh;
  fpfh.setInputCloud(vox);
  fpfh.setInputNormals(normals);
  fpfh.setSearchMethod(kdtree);
  fpfh.setRadiusSearch(0.75);
  pcl::PointCloud<pcl::FPFHSignature33> desc;
  fpfh.compute(desc);
  std::vector<float> nrm(normals->size() * 4), d(desc.size() * 33);
  for (size_t i = 0; i < normals->size(); ++i) {
    nrm[4 * i] = (*normals)[i].normal_x;
    nrm[4 * i + 1] = (*normals)[i].normal_y;
    nrm[4 * i + 2] = (*normals)[i].normal_z;
    nrm[4 * i + 3] = (*normals)[i].curvature;
  }
  for (size_t i = 0; i < desc.size(); ++i)
    for (int k = 0; k < 33; ++k) d[33 * i + k] = desc[i].histogram[k];
  write_bin(out + "/normals_" + tag + ".bin", static_cast<int>(normals->size()), 4, 0, nrm.data());
  write_bin(out + "/fpfh_" + tag + ".bin", static_cast<int>(desc.size()), 33, 0, d.data());
  std::printf("%s: %zu raw -> %zu voxels\n", tag.c_str(), raw->size(), vox->size());
}

// src/graph.cc:12-104, PMC_HEU branch, from the CSR arrays findMaxClique builds out of a teaser::Graph
void clique(const std::string& in, const std::string& out, int g) {
  const std::string base = in + "/graph" + std::to_string(g);
  std::ifstream f(base + ".csr", std::ios::binary);
  int32_t n = 0, m = 0;
  f.read(reinterpret_cast<char*>(&n), 4);
  f.read(reinterpret_cast<char*>(&m), 4);
  std::vector<int64_t> off64(static_cast<size_t>(n) + 1);
  std::vector<int32_t> adj(static_cast<size_t>(m));
  f.read(reinterpret_cast<char*>(off64.data()), static_cast<std::streamsize>(off64.size()) * 8);
  f.read(reinterpret_cast<char*>(adj.data()), static_cast<std::streamsize>(adj.size()) * 4);
  std::vector<long long> vertices(off64.begin(), off64.end());
  std::vector<int> edges(adj.begin(), adj.end());
  pmc::pmc_graph G(vertices, edges);
  pmc::input inp;
  inp.algorithm = 0;
  inp.threads = 1;  // the reference asks for 12 (src/graph.cc:39): its result then depends on thread timing (divergence D7)
  inp.experiment = 0;
  inp.lb = 0;
  inp.ub = 0;
  inp.param_ub = 0;
  inp.adj_limit = 20000;
  inp.time_limit = 3600;
  inp.remove_time = 4;
  inp.graph_stats = false;
  inp.verbose = false;
  inp.help = false;
  inp.MCE = false;
  inp.decreasing_order = false;
  inp.heu_strat = "kcore";
  inp.vertex_search_order = "deg";
  std::vector<int> C;
  G.compute_cores();
  const int max_core = G.get_max_core();
  std::vector<int>* kc = G.get_kcores();  // PMC: V + 1 entries, shifted by one
  std::vector<int32_t> cores(kc->begin(), kc->end());
  write_bin(out + "/cores_g" + std::to_string(g) + ".bin", static_cast<int>(cores.size()), 1, 1, cores.data());
  inp.ub = max_core + 1;
  pmc::pmc_heu maxclique(G, inp);
  inp.lb = maxclique.search(G, C);
  std::vector<int32_t> c32(C.begin(), C.end());
  if (c32.empty()) c32.push_back(-1);
  write_bin(out + "/clique_g" + std::to_string(g) + ".bin", static_cast<int>(C.size()), 1, 1, c32.data());
  std::printf("graph %d: %d vertices, max core %d, heuristic clique %zu\n", g, n, max_core, C.size());
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <inputs dir> <output dir>\n", argv[0]);
    return 2;
  }
  const std::string in = argv[1], out = argv[2];
  front_end(in, out, "src");
  front_end(in, out, "tgt");
  for (int g = 0; g < 3; ++g) clique(in, out, g);
  return 0;
}
