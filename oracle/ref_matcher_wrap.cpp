// C entry point around the REFERENCE's teaser::Matcher (include/teaser_utils/feature_matcher.h +
// src/teaser_utils/feature_matcher.cc of /root/reference, compiled in place by oracle/Makefile against the stand-in
// headers of oracle/ref_shim/).  Test infrastructure: used by tests/ to validate oracle/quatro_oracle.cpp's restatement
// of calculateCorrespondences / advancedMatching, and by tests/golden/make_golden.py to produce matcher_ref.npz.
#include <cstdint>
#include <cstring>

#include "../include/qtr_math.h"
#include "teaser_utils/feature_matcher.h"

static uint64_t g_seed = 0, g_counter = 0;
extern "C" unsigned int qref_rand_u32(void) { return qm_rand_u32(g_seed, g_counter++); }

// xyz: n x 3 floats; desc: n x 33 floats.  corr: capacity cap pairs (src, tgt).  Returns the number of pairs the
// reference produced (which may exceed cap; only cap are written).
extern "C" int ref_calculate_correspondences(const float* xyz_s, int ns, const float* desc_s, const float* xyz_t, int nt,
                                             const float* desc_t, int use_absolute_scale, int use_crosscheck,
                                             int use_tuple_test, float tuple_scale, unsigned long long seed, int* corr,
                                             int cap) {
  teaser::PointCloud src, tgt;
  teaser::FPFHCloud fs, ft;
  for (int i = 0; i < ns; ++i) {
    src.push_back({xyz_s[3 * i], xyz_s[3 * i + 1], xyz_s[3 * i + 2]});
    pcl::FPFHSignature33 f;
    std::memcpy(f.histogram, desc_s + (size_t)33 * i, sizeof(f.histogram));
    fs.push_back(f);
  }
  for (int i = 0; i < nt; ++i) {
    tgt.push_back({xyz_t[3 * i], xyz_t[3 * i + 1], xyz_t[3 * i + 2]});
    pcl::FPFHSignature33 f;
    std::memcpy(f.histogram, desc_t + (size_t)33 * i, sizeof(f.histogram));
    ft.push_back(f);
  }
  g_seed = seed;
  g_counter = 0;
  teaser::Matcher matcher;
  const auto c = matcher.calculateCorrespondences(src, tgt, fs, ft, use_absolute_scale != 0, use_crosscheck != 0,
                                                  use_tuple_test != 0, tuple_scale);
  for (size_t i = 0; i < c.size() && (int)i < cap; ++i) {
    corr[2 * i] = c[i].first;
    corr[2 * i + 1] = c[i].second;
  }
  return (int)c.size();
}
