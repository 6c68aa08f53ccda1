// unity.hip — single translation unit of libquatro_hip.so (all kernels + the C ABI).
#include "solver.hip"
#include "stages.hip"
#include "frontend.hip"
#include "match.hip"
#include "segment.hip"
#include "patchwork.hip"
#include "capi.hip"
#include "formats.hip"
