// Force-included in front of the reference's feature_matcher.cc (oracle/Makefile): the reference seeds the tuple test
// with srand(time(NULL)) and draws rand() % ncorr (src/teaser_utils/feature_matcher.cc:189-198).  To compare its output
// with the oracle's, the draws are redirected — by macro, the reference text itself is compiled unmodified — to the
// counter generator the oracle and the kernels use (include/qtr_math.h, declared divergence D1).
#pragma once
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <algorithm>
#include <vector>
extern "C" unsigned int qref_rand_u32(void);
#define rand() qref_rand_u32()
#define srand(x) ((void)(x))
