// cache_demo.cpp — the matched-pair cache of the reference's FPFHManager (include/fpfh_manager.hpp:179-232) through
// include/fpfh_manager.hpp of THIS repository.  usage: cache_demo dir src_idx tgt_idx out_src_idx out_tgt_idx
#include <cstdio>
#include <cstdlib>

#include "fpfh_manager.hpp"

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  FPFHManager fm(0.5, 0.75);
  try {
    fm.loadFeaturePair(0, 1);
    return 3;  // must have thrown: no load dir
  } catch (const std::invalid_argument&) {
  }
  fm.setLoadDir(argv[1]);
  fm.setSaveDir(argv[1]);
  fm.loadFeaturePair(std::atoi(argv[2]), std::atoi(argv[3]));
  std::printf("%zu %zu\n", fm.getSrcKps().size(), fm.getTgtKps().size());
  fm.saveFeaturePair(std::atoi(argv[4]), std::atoi(argv[5]));
  try {
    fm.loadFeaturePair(999, 999);
    return 4;
  } catch (const std::invalid_argument&) {
  }
  return 0;
}
