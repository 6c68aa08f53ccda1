// teaser_utils/feature_matcher.h — drop-in for url-kaist/Quatro's include/teaser_utils/feature_matcher.h: class
// teaser::Matcher (:16-110) whose calculateCorrespondences (:42-74 + src/teaser_utils/feature_matcher.cc:18-265:
// FLANN kd-trees, mutual 1-NN, tuple test, sort + unique) becomes one qtr_match call (MFMA distance contraction
// with certified arg-min, cross-check, counter-RNG tuple test, compaction in source order).
//
// Differences a caller can observe: the tuple test draws from a counter-based generator seeded by `seed_` instead
// of srand(time(NULL)) (feature_matcher.cc:189), so equal inputs give equal outputs; use_absolute_scale = false
// (a global rescale of both clouds, :44-76) is accepted and has no effect — mutual nearest neighbours do not
// depend on it and the tuple test compares length ratios.
#pragma once
#include <utility>
#include <vector>

#include "fpfh.h"

namespace teaser {

class Matcher {
 public:
  unsigned long long seed_ = 0;

  Matcher() = default;

  std::vector<std::pair<int, int>> calculateCorrespondences(teaser::PointCloud& source_points,
                                                            teaser::PointCloud& target_points,
                                                            FPFHCloud& source_features, FPFHCloud& target_features,
                                                            bool use_absolute_scale = true, bool use_crosscheck = true,
                                                            bool use_tuple_test = true, float tuple_scale = 0) {
    (void)use_absolute_scale;
    corres_.clear();
    const int ns = static_cast<int>(source_points.size()), nt = static_cast<int>(target_points.size());
    if (ns == 0 || nt == 0) return corres_;
    std::vector<float> xs(static_cast<size_t>(4) * ns, 0.f), xt(static_cast<size_t>(4) * nt, 0.f);
    std::vector<float> ds(static_cast<size_t>(33) * ns), dt(static_cast<size_t>(33) * nt);
    pack(source_points, source_features, xs, ds);
    pack(target_points, target_features, xt, dt);
    qtr_frontend_params fp;
    qtr_default_frontend_params(&fp);
    fp.use_crosscheck = use_crosscheck ? 1 : 0;
    fp.use_tuple_test = use_tuple_test ? 1 : 0;
    fp.tuple_scale = tuple_scale;
    fp.seed = seed_;
    const int cap = ns < nt ? nt : ns;
    std::vector<int> corr2(static_cast<size_t>(2) * cap);
    int L = 0;
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_match(h, slot_lease.slot, xs.data(), ns, ds.data(), xt.data(), nt, dt.data(), &fp, corr2.data(), cap, &L,
                                   QTR_MEM_HOST));
    corres_.reserve(static_cast<size_t>(L));
    for (int c = 0; c < L; ++c) corres_.emplace_back(corr2[2 * static_cast<size_t>(c)], corr2[2 * static_cast<size_t>(c) + 1]);
    return corres_;
  }

 private:
  static void pack(const teaser::PointCloud& pts, const FPFHCloud& feat, std::vector<float>& xyz4, std::vector<float>& d33) {
    for (size_t i = 0; i < pts.size(); ++i) {
      xyz4[4 * i] = pts[i].x;
      xyz4[4 * i + 1] = pts[i].y;
      xyz4[4 * i + 2] = pts[i].z;
      for (int k = 0; k < 33; ++k) d33[33 * i + static_cast<size_t>(k)] = feat.points[i].histogram[k];
    }
  }
  std::vector<std::pair<int, int>> corres_;
};

}  // namespace teaser
